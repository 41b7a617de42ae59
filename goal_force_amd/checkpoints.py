"""Checkpoint files -> the modules of the Goal-Force pipeline (host-side I/O around the hot path).

What `WanVideoPipeline.from_pretrained` (GF:483-595) gets through the reference's ModelManager, restated for local files
only (there is no network): every `ModelConfig` names one model — a list of .safetensors shards (the two A14B experts,
INF:86-99), a `.pth` (umT5 encoder INF:100, VAE INF:101) — and the model's kind is recognised from its state-dict keys,
as the reference does (it hashes the key set; here the distinguishing keys are named).  Architecture sizes are read
from the tensor shapes, so the A14B files and small synthetic checkpoints go through the same code.

  kind            recognised by                                   converter
  wan_video_dit   blocks.0.self_attn.q.weight + patch_embedding   none (the civitai layout IS the module layout, DIT:423-466)
  text encoder    token_embedding.weight + blocks.0.attn.q.weight none (wan_video_text_encoder.py:261-269)
  wan_video_vae   (model_state.)encoder.conv1.weight + decoder.*  keys gain the 'model.' prefix (VAE:1251-1262)
"""
from __future__ import annotations

import contextlib
import threading
import glob
import os
from typing import Optional

import torch

from ._lib import GoalForceError


class ModelConfig:
    """diffsynth/utils ModelConfig (UTIL:158-214): where a model's files live.  `path` (a file, a list of shard files or a
    folder) is used as given; with only `model_id` + `origin_file_pattern` the files must already be under
    `<local_model_path or ./models>/<model_id>/` — what the reference's download step would have produced."""

    def __init__(self, path=None, model_id=None, origin_file_pattern=None, download_resource="ModelScope",
                 offload_device=None, offload_dtype=None, local_model_path=None, skip_download=False):
        self.path, self.model_id, self.origin_file_pattern = path, model_id, origin_file_pattern
        self.download_resource = download_resource
        self.offload_device, self.offload_dtype = offload_device, offload_dtype
        self.local_model_path, self.skip_download = local_model_path, skip_download

    def download_if_necessary(self, use_usp=False):
        """UTIL:168-214 without the download: resolves `path` from the local model folder or fails loudly."""
        if self.path is not None:
            return
        if self.model_id is None:
            raise ValueError('No valid model files. Please use `ModelConfig(path="xxx")` or '
                             '`ModelConfig(model_id="xxx/yyy", origin_file_pattern="zzz")`.')
        root = os.path.join(self.local_model_path or "./models", self.model_id)
        pat = self.origin_file_pattern or ""
        if pat == "" or (isinstance(pat, str) and pat.endswith("/")):
            found = os.path.join(root, pat)
            ok = os.path.isdir(found)
        else:
            pats = [pat] if isinstance(pat, str) else list(pat)
            found = sorted(f for p_ in pats for f in glob.glob(os.path.join(root, p_)))
            ok = bool(found)
            if len(found) == 1:
                found = found[0]
        if not ok:
            raise GoalForceError(f"ModelConfig({self.model_id!r}, {self.origin_file_pattern!r}): nothing under {root} and "
                                 "no network to download from — place the files there or pass path=")
        self.path = found


def load_state_dict(path, torch_dtype=None, device="cpu"):
    """safetensors / torch checkpoint reader over one file or a list of shards (diffsynth/models/utils.py load_state_dict)."""
    paths = path if isinstance(path, (list, tuple)) else [path]
    sd = {}
    for p in paths:
        if str(p).endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(p, framework="pt", device=str(device)) as f:
                for k in f.keys():
                    t = f.get_tensor(k)
                    sd[k] = t.to(torch_dtype) if torch_dtype is not None else t
        else:
            part = torch.load(p, map_location=device, weights_only=True)
            for k, t in part.items():
                sd[k] = t.to(torch_dtype) if (torch_dtype is not None and torch.is_tensor(t) and t.is_floating_point()) else t
    return sd


@contextlib.contextmanager
def params_on_meta():
    """Modules built inside allocate no parameter storage and run no random initialisation: every parameter is moved to the
    meta device the moment it is registered (what the reference's `init_weights_on_device(torch.device("meta"))` does,
    diffsynth/models/model_manager.py:71-75).  Buffers and plain tensor attributes (RoPE tables, VAE statistics) stay real.
    The checkpoint's tensors are then ADOPTED with `load_state_dict(assign=True)` instead of being copied into a second,
    randomly initialised fp32 copy of the model — for an A14B expert that copy would be 57 GB of host memory and minutes of
    CPU random-number generation."""
    # The patch replaces a class attribute of torch.nn.Module, i.e. it is process-wide while installed.  It is therefore
    # (1) installed under a module-level lock by the OUTERMOST context only (nested contexts just count), restored by that same
    # context — never in the wrong order —, and (2) inert for every other thread: a module another thread builds meanwhile
    # registers real parameters as usual.  Two loaders in two threads serialise on the lock.
    me = threading.get_ident()
    with _META_LOCK:
        if _META_STATE["depth"] and _META_STATE["owner"] == me:        # nested use by the owner: nothing to install
            _META_STATE["depth"] += 1
            try:
                yield
            finally:
                _META_STATE["depth"] -= 1
            return
        old = torch.nn.Module.register_parameter

        def register(module, name, param):
            old(module, name, param)
            if param is not None and threading.get_ident() == me:
                p = module._parameters[name]
                module._parameters[name] = torch.nn.Parameter(p.detach().to("meta"), requires_grad=p.requires_grad)

        _META_STATE.update(depth=1, owner=me)
        torch.nn.Module.register_parameter = register
        try:
            yield
        finally:
            torch.nn.Module.register_parameter = old
            _META_STATE.update(depth=0, owner=None)


_META_LOCK = threading.RLock()
_META_STATE = {"depth": 0, "owner": None}


def _adopt(module, sd, what):
    """strict load that takes over the checkpoint's tensors; a parameter the file does not provide would stay on the meta
    device — strict=True names it first."""
    module.load_state_dict(sd, strict=True, assign=True)
    left = [n for n, p in module.named_parameters() if p.is_meta]
    if left:
        raise GoalForceError(f"{what}: parameters without checkpoint data: {left[:4]}")
    return module


def model_kind(sd) -> Optional[str]:
    if "model_state" in sd and isinstance(sd["model_state"], dict):
        sd = sd["model_state"]
    if "blocks.0.self_attn.q.weight" in sd and "patch_embedding.weight" in sd:
        return "wan_video_dit"
    if "token_embedding.weight" in sd and "blocks.0.attn.q.weight" in sd:
        return "wan_video_text_encoder"
    if any(k.endswith("encoder.conv1.weight") for k in sd) and any(k.endswith("decoder.conv1.weight") for k in sd):
        return "wan_video_vae"
    return None


def _count(sd, fmt):
    n = 0
    while fmt.format(n) in sd:
        n += 1
    return n


def build_dit(sd, head_dim=128):
    """WanModel sized from the checkpoint: Wan2.2-I2V-A14B gives dim 5120 / in 36 / ffn 13824 / 40 layers / 40 heads
    (DIT:703-718; the Wan family keeps 128-wide heads)."""
    from .dit import WanModel
    pw = sd["patch_embedding.weight"]
    if any(k.startswith(("img_emb.", "ref_conv.", "control_adapter.")) or ".k_img." in k for k in sd):
        raise NotImplementedError("this DiT checkpoint has CLIP-image / reference / camera branches: not a Goal-Force expert")
    dim, in_dim = int(pw.shape[0]), int(pw.shape[1])
    cfg = dict(has_image_input=False, patch_size=tuple(int(v) for v in pw.shape[2:]), in_dim=in_dim, dim=dim,
               ffn_dim=int(sd["blocks.0.ffn.0.weight"].shape[0]), freq_dim=int(sd["time_embedding.0.weight"].shape[1]),
               text_dim=int(sd["text_embedding.0.weight"].shape[1]),
               out_dim=int(sd["head.head.weight"].shape[0]) // (int(pw.shape[2]) * int(pw.shape[3]) * int(pw.shape[4])),
               num_heads=dim // head_dim, num_layers=_count(sd, "blocks.{}.self_attn.q.weight"), eps=1e-6,
               require_clip_embedding=False)
    with params_on_meta():
        m = WanModel(**cfg)
    return _adopt(m, sd, "wan_video_dit")


def build_text_encoder(sd, head_dim=64):
    """WanTextEncoder sized from the checkpoint (umT5-XXL: vocab 256384, dim 4096, 64 heads x 64, FFN 10240, 24 layers)."""
    from .text_encoder import WanTextEncoder
    vocab, dim = (int(v) for v in sd["token_embedding.weight"].shape)
    dim_attn = int(sd["blocks.0.attn.q.weight"].shape[0])
    shared = "pos_embedding.embedding.weight" in sd
    pe = sd["pos_embedding.embedding.weight" if shared else "blocks.0.pos_embedding.embedding.weight"]
    with params_on_meta():
        m = WanTextEncoder(vocab=vocab, dim=dim, dim_attn=dim_attn, dim_ffn=int(sd["blocks.0.ffn.fc1.weight"].shape[0]),
                           num_heads=int(pe.shape[1]), num_layers=_count(sd, "blocks.{}.attn.q.weight"),
                           num_buckets=int(pe.shape[0]), shared_pos=shared)
    return _adopt(m, sd, "wan_video_text_encoder")


def build_vae(sd):
    """WanVideoVAE from `Wan2.1_VAE.pth`: the file holds the inner model's keys (optionally under 'model_state'); the
    module keeps them under 'model.' (WanVideoVAEStateDictConverter.from_civitai, VAE:1256-1262)."""
    from .vae import WanVideoVAE
    if "model_state" in sd:
        sd = sd["model_state"]
    with params_on_meta():
        m = WanVideoVAE()
    return _adopt(m, {("model." + k): v for k, v in sd.items()}, "wan_video_vae")


BUILDERS = {"wan_video_dit": build_dit, "wan_video_text_encoder": build_text_encoder, "wan_video_vae": build_vae}


def load_model(model_config: ModelConfig, torch_dtype=torch.bfloat16, device="cuda"):
    """One ModelConfig -> (kind, module on `device` in `torch_dtype`).  ModelManager.load_model (GF:520-526) for the three
    kinds Goal Force loads; anything else is refused by name."""
    model_config.download_if_necessary()
    path = model_config.path
    if isinstance(path, str) and os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "*.safetensors"))) or sorted(glob.glob(os.path.join(path, "*.pth")))
    # every tensor is cast to torch_dtype as it is read and lands on `device` directly (safetensors reads straight into device
    # memory): the host never holds more than one tensor of the model.  ModelConfig.offload_device / offload_dtype are accepted
    # and unused — nothing is offloaded on a 288 GB part.
    sd = load_state_dict(path, torch_dtype=torch_dtype, device=device)
    kind = model_kind(sd)
    if kind is None:
        some = ", ".join(list(sd)[:4])
        raise NotImplementedError(f"checkpoint {model_config.path}: not a Wan DiT expert, umT5 encoder or Wan VAE "
                                  f"(first keys: {some}); the other DiffSynth model families are out of scope")
    module = BUILDERS[kind](sd)
    del sd
    return kind, module.to(device=device, dtype=torch_dtype)     # parameters are there already; this moves the (few) buffers
