#!/usr/bin/env python3
"""Time the pre-loop units of one Goal-Force video on one MI355X (the 'next' rows of SURVEY §8f that sit before the
denoising loop): tiled VAE encode of the control video and of the image conditioning (GF:791-805, 887-917; tile
(30,52)/(15,26) as the inference scripts use), the umT5-XXL text encoder at 512 tokens (positive + negative prompt,
wan_video_text_encoder.py:209-255) and the force-map render (DS:775-940).  Random-init weights of the real shapes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from goal_force_amd.text_encoder import WanTextEncoder  # noqa: E402
from goal_force_amd.vae import WanVideoVAE  # noqa: E402

BF = torch.bfloat16


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    dev = torch.device("cuda", 0)
    torch.set_grad_enabled(False)
    torch.manual_seed(7)
    out = {}
    vae = WanVideoVAE().to(BF).to(dev)
    video = (torch.rand((3, 81, 480, 832), device=dev) * 2 - 1).to(BF)
    kw = dict(device=dev, tiled=True, tile_size=(30, 52), tile_stride=(15, 26))
    out["vae_tiled_encode_81f_s"] = timed(lambda: vae.encode([video], **kw))
    z = vae.encode([video], **kw)
    assert tuple(z.shape) == (1, 16, 21, 60, 104)
    out["vae_tiled_decode_81f_s"] = timed(lambda: vae.decode(z, tiled=True, tile_size=(30, 52), tile_stride=(15, 26)))
    del vae
    # umT5-XXL encoder, 512 tokens, 40 real (the prompter zeroes the rest)
    with torch.device("meta"):
        te = WanTextEncoder()
    te = te.to_empty(device=dev).to(BF)
    g = torch.Generator(device=dev).manual_seed(1)
    for n, p in te.named_parameters():
        if n.endswith("norm.weight") or "norm1" in n or "norm2" in n:
            p.data.fill_(1.0)
        else:
            p.data.copy_(torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * 0.02)
    ids = torch.randint(0, 256384, (1, 512), device=dev)
    mask = torch.zeros((1, 512), dtype=torch.long, device=dev)
    mask[:, :40] = 1
    out["umt5_xxl_512_tokens_one_prompt_s"] = timed(lambda: te(ids, mask))
    out["umt5_xxl_params"] = sum(p.numel() for p in te.parameters())
    del te
    from goal_force_amd.force_map import plan_control_video, render_control_video
    # BASELINE config 4: projectile force 200 at 45 degrees, projectile mass 2.0, no goal force (ch0 + ch2 active)
    plan = plan_control_video(200.0, 45.0, 0.3, 0.4, -1, 0.0, 0.0, 0.0, 81, 480, 832,
                              {"projectile": 2.0, "target": -1, "distractors": []},
                              {"projectile": [250, 190], "target": [0, 0], "distractors": []}, 30.0, 400.0, 30.0, 400.0, 1.0, 4.0)
    out["force_map_render_81f_s"] = timed(lambda: render_control_video(plan, dev), reps=3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
