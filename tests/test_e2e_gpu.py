"""End-to-end run of the inference driver at the real resolution (832x480x81f) with random weights and a reduced
depth/step count: CSV row -> force map (HIP) -> VAE encode of the force map and of the first frame (HIP) ->
denoise loop with ControlNet + CFG + expert switch (HIP) -> tiled VAE decode (HIP) -> 81 frames."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

# the schema / numbers of the reference's example rows (README.md:92-107; goal-force mode: projectile force -1)
CSV = ("image,projectile_force_angle,projectile_force_magnitude,projectile_coordx,projectile_coordy,projectile_mass,"
       "target_indirect_force_angle,target_indirect_force_magnitude,target_coordx,target_coordy,target_mass,width,height,caption\n"
       "scene.png,-1.0,-1.0,368,108,-1,0.0,350.0,545,114,2.0,832,480,\"The pendulum swings, striking and toppling the red block.\"\n")


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "inference_goal_force.py")] + [str(x) for x in args],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r


def _check_frames(folder, n):
    from PIL import Image
    frames = sorted(os.listdir(folder))
    assert len(frames) == n
    im = Image.open(os.path.join(folder, frames[n // 2]))
    assert im.size == (832, 480)
    a = np.asarray(im).astype(np.float32)
    assert np.isfinite(a).all() and a.std() > 1.0   # not a constant / NaN frame


def test_inference_driver_end_to_end(tmp_path):
    """The reference launcher's own arguments (INF:38-56) — --device_id --world_size --seed --control_signal_type goal_force
    --example_paths — plus --synthetic in place of --model_ckpt_path (no checkpoint exists here); file names as INF:178-186."""
    from PIL import Image
    ex = tmp_path / "example"
    (ex / "images").mkdir(parents=True)
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (480, 832, 3), dtype=np.uint8)).save(ex / "images" / "scene.png")
    (ex / "row.csv").write_text(CSV)
    out = tmp_path / "out"
    _run(["--device_id", 0, "--world_size", 1, "--seed", 0, "--control_signal_type", "goal_force", "--example_paths", ex / "row.csv",
          "--synthetic", "--layers", 2, "--num_inference_steps", 3, "--output_dir", out])
    root = ("step-synthetic_scene__prj_coords_0.44_0.23__tgt_coords_0.66_0.24__prj_mass_-1.0__tgt_mass_2.0__prj_force_-1.0__prj_angle_-1.0"
            "__tgt_indirect_force_350.0__tgt_indirect_angle_0.0__seed_0")
    assert sorted(os.listdir(out)) == [root, root + "-image_condition.png", root + "-text.json"], os.listdir(out)
    _check_frames(out / root, 81)


def test_inference_driver_canny_edge_control(tmp_path):
    """`--control_signal_type canny_edge` (scripts/inference/inference_canny_edge_control.py:118-186): CSV rows (image, control_video,
    caption), the pre-computed edge clip under canny-videos/ through the dataset's video operator (centre crop + resize to 480 x 832),
    x / 127.5 - 1 in bf16, 49 frames -> 13 latent frames; two layers, three steps, random weights."""
    from PIL import Image
    ex = tmp_path / "canny"
    (ex / "images").mkdir(parents=True)
    (ex / "canny-videos").mkdir()
    rng = np.random.default_rng(1)
    Image.fromarray(rng.integers(0, 255, (480, 832, 3), dtype=np.uint8)).save(ex / "images" / "street.png")
    clip = (rng.random((49, 96, 170, 1)) > 0.9).astype(np.uint8).repeat(3, axis=-1) * 255        # sparse white edges, another aspect ratio
    np.save(ex / "canny-videos" / "street_canny.npy", clip)
    (ex / "rows.csv").write_text('image,control_video,caption\nstreet.png,street_canny.npy,"A street at dusk."\n')
    out = tmp_path / "out"
    _run(["--seed", 5, "--control_signal_type", "canny_edge", "--example_paths", ex / "rows.csv", "--synthetic", "--layers", 2,
          "--num_inference_steps", 3, "--output_dir", out])
    assert sorted(os.listdir(out)) == ["street_canny.npy-canny-output", "street_canny.npy-image-condition.png"], os.listdir(out)
    _check_frames(out / "street_canny.npy-canny-output", 49)


def test_training_launcher_end_to_end(tmp_path):
    """scripts/train.py with the reference's training command line (scripts/train/train_goal_force.sh) on synthetic tiny checkpoints
    and a three-set data tree (clips as frame directories — no video decoder in this image): from_pretrained, freeze all but the
    ControlNet, dataset mux in training mode with channel masking, forward_preprocess, training_loss + backward on the HIP kernels,
    AdamW / ConstantLR / clipping, checkpoints every 2 steps with `pipe.controlnet.*` keys, then a resumed run."""
    import json
    from PIL import Image
    from safetensors.torch import load_file
    from test_from_pretrained import _write_checkpoints
    root = tmp_path / "models" / "Wan-AI" / "Wan2.1-T2V-1.3B"
    root.parent.mkdir(parents=True)
    paths, _, _ = _write_checkpoints(root)
    rng = np.random.default_rng(5)

    def clip(folder, name, n):
        d = tmp_path / folder / name
        d.mkdir(parents=True)
        for i in range(n):
            Image.fromarray(np.kron(rng.integers(0, 255, (30, 52, 3), dtype=np.uint8), np.ones((16, 16, 1), dtype=np.uint8))).save(d / f"{i:03d}.png")
    clip("balls", "b0.mp4", 9)
    clip("dominos", "d0.mp4", 19)
    clip("plants", "fern0.mp4", 5)
    for folder, a, b in (("balls", "b0.mp4", "b1.mp4"), ("dominos", "d0.mp4", "d1.mp4"), ("plants", "fern0.mp4", "fern1.mp4")):
        os.symlink(tmp_path / folder / a, tmp_path / folder / b)          # a second listed clip per set: the force / mass ranges need two values
    hdr = ("video,projectile_force_angle,projectile_force_magnitude,projectile_coordx,projectile_coordy,projectile_mass,"
           "target_indirect_force_angle,target_indirect_force_magnitude,target_coordx,target_coordy,target_mass,width,height,caption\n")
    (tmp_path / "balls.csv").write_text(hdr + 'b0.mp4,10,120.0,100,200,1.5,30,80.5,400,210,2.0,832,480,"the ball rolls"\n'
                                        'b1.mp4,200,250.0,300,100,3.0,120,160.0,500,90,1.0,832,480,"a ball rolls"\n'
                                        'b9.mp4,20,300.0,100,200,3.0,30,200.0,400,210,3.0,832,480,"listed, but the clip is missing"\n')
    (tmp_path / "dominos.csv").write_text(hdr + 'd0.mp4,0,55.0,10,20,2.0,0,66.0,30,40,2.5,832,480,"the red block"\n'
                                          'd1.mp4,90,75.0,110,210,3.0,180,44.0,310,410,3.0,832,480,"a red block"\n'
                                          'd9.mp4,0,75.0,10,20,3.0,0,96.0,30,40,3.5,832,480,"missing too"\n')
    (tmp_path / "plants.csv").write_text('video,force,angle,coordx,coordy,width,height,caption\nfern0.mp4,12.5,30,100,100,832,480,"a ball"\n'
                                         'fern1.mp4,40.0,200,400,240,832,480,"the ball"\n'
                                         'fern9.mp4,48.0,30,100,100,832,480,"missing"\n')
    model_paths = json.dumps([paths["high_noise_model"][0], str(root / "models_t5_umt5-xxl-enc-bf16.pth"), str(root / "Wan2.1_VAE.pth")])
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "train.py"), "--dataset_base_path", "balls", "dominos", "plants",
           "--dataset_metadata_path", "balls.csv", "dominos.csv", "plants.csv", "--control_signal_type", "direct_force_and_goal_force_and_mass",
           "--controlnet_num_layers", "1", "--height", "480", "--width", "832", "--num_frames", "5", "--dataset_repeat", "1",
           "--model_paths", model_paths, "--learning_rate", "1e-4", "--num_epochs", "1", "--save_steps", "2", "--remove_prefix_in_ckpt", "pipe.dit.",
           "--trainable_models", "controlnet", "--output_path", "out", "--extra_inputs", "input_image", "--max_timestep_boundary", "0.358",
           "--min_timestep_boundary", "0", "--max_grad_norm", "1", "--p_mask_out_masses", "0.5", "--p_mask_out_direct_force", "0.5",
           "--p_mask_out_indirect_force", "0.5", "--wandb_logging"]
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "Dataset size:  6" in r.stdout and "Starting training from scratch" in r.stdout
    (run,) = os.listdir(tmp_path / "out")
    files = sorted(os.listdir(tmp_path / "out" / run))
    assert files == ["step-2.safetensors", "step-4.safetensors", "step-6.safetensors"], files
    sd = load_file(str(tmp_path / "out" / run / "step-6.safetensors"))
    assert all(k.startswith("pipe.controlnet.") for k in sd) and any("controlnet_zero_convs_after.0.weight" in k for k in sd)
    zc = sd["pipe.controlnet.controlnet_zero_convs_after.0.weight"].float()
    assert bool(torch.isfinite(zc).all()) and float(zc.abs().max()) > 0, "training from scratch moves the zero-convolution first"
    # resuming: the output goes next to the checkpoint and the step count continues at N + 1 (utils.py:773-785)
    r = subprocess.run(cmd + ["--controlnet_checkpoint", str(tmp_path / "out" / run / "step-4.safetensors")], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ControlNet checkpoint loaded" in r.stdout
    # from step-4: the count continues at 5 (sic), six items -> 6 .. 11; saves at the multiples of 2 and once more at the end
    assert sorted(os.listdir(tmp_path / "out" / run), key=lambda f: int(f.split("-")[1].split(".")[0])) == [
        "step-2.safetensors", "step-4.safetensors", "step-6.safetensors", "step-8.safetensors", "step-10.safetensors", "step-11.safetensors"]
