"""WanVideoPipeline — drop-in for the Goal-Force fork of the DiffSynth pipeline
(src/goal_force/wan_video_new.py:120-737; `pipe(...)` kwargs GF:599-661 kept verbatim).

MI355X design choices (SURVEY.md §7 'Memory plan'): both 14B experts, both ControlNets and the VAE stay
resident in the 288 GB of HBM — there is no offload state machine, `enable_vram_management()` is an
accepted no-op (the fp32-LayerNorm numerics it switches on in the reference are what the HIP
LayerNorm kernel always computes).  The denoising loop is `denoise()`: two model_fn calls per step
(cond / uncond), CFG + Euler update fused in one kernel, expert switch by pointer swap.

Pre-loop conditioning (umT5 text encoder, VAE encoder — the SURVEY.md §8f rows, also on the HIP kernels) runs when
the models are attached; it can also be supplied pre-computed: `context_posi`, `context_nega`, `y`,
`control_signal_video_latents`.
"""
from __future__ import annotations

import copy
from typing import Optional

import numpy as np
import torch

from . import ops
from ._lib import GoalForceError
from .checkpoints import ModelConfig, load_model, load_state_dict  # noqa: F401  (re-exported under the reference's names)
from .controlnet import ControlNet
from .dit import A14B_CONFIG, WanModel
from .model_fn import ContextCache, model_fn_wan_video
from .scheduler import FlowMatchScheduler


class WanVideoPipeline(torch.nn.Module):
    """`WanVideoPipeline(BasePipeline)` (GF:120-737 on UTIL:13-157).  Like the reference's it IS a torch.nn.Module whose children are
    the models (`named_children`, `parameters`, `to`, `freeze_except` work as there: the training module wraps it, utils.py:428-447)."""

    def __init__(self, device="cuda", torch_dtype=torch.bfloat16, tokenizer_path=None, controlnet=False,
                 controlnet_num_layers=0, controlnet_stride=None):
        super().__init__()
        if torch_dtype != torch.bfloat16:
            raise GoalForceError("the HIP kernels compute in bf16 storage / fp32 accumulate only")
        self.device, self.torch_dtype = device, torch_dtype
        self.height_division_factor, self.width_division_factor = 16, 16
        self.time_division_factor, self.time_division_remainder = 4, 1
        self.scheduler = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)  # GF:127
        from .text_encoder import WanPrompter
        self.prompter = WanPrompter(tokenizer_path=tokenizer_path)   # GF:128
        self.text_encoder = None
        self.image_encoder = None
        self.dit: Optional[WanModel] = None
        self.dit2: Optional[WanModel] = None
        self.vae = None
        self.motion_controller = None
        self.vace = None
        self.controlnet: Optional[ControlNet] = None
        self.controlnet2: Optional[ControlNet] = None
        self.model_fn = model_fn_wan_video  # GF:161 — the reference's own swap point
        self.share_cfg_prefix = True        # denoise(): block 0's self-attention half once per CFG step (bit-identical)
        # denoise(): the uncond forward of a CFG step on a second HIP stream, concurrent with the cond forward (same kernels, same
        # bits).  Measured (3 % slower at full size) and left off; a plain attribute, set by whoever wants the A/B.
        self.cfg_streams = False
        self._cfg_side_stream = None
        self.vram_management_enabled = False
        self.elide_zero_controlnet = True
        self.num_layers = controlnet_num_layers
        self._want_controlnet = controlnet
        self._controlnet_stride = controlnet_stride
        self.last_step_ms = []

    # ------------------------------------------------------------------ construction
    @staticmethod
    def from_pretrained(torch_dtype=torch.bfloat16, device="cuda", model_configs=(),
                        tokenizer_config="default", audio_processor_config=None, redirect_common_files=True, use_usp=False,
                        controlnet=False, controlnet_num_layers=0, controlnet_stride=None, apply_strided_controlnet=False):
        """GF:483-595 for local files (no network): every ModelConfig is loaded and recognised by its keys
        (checkpoints.load_model) — Wan DiT experts in the order given (high-noise first, then low-noise: GF:529-533), the umT5
        text encoder, the Wan VAE; `tokenizer_config.path` is handed to the prompter (GF:584-586); ControlNet blocks
        are initialised as copies of DiT blocks 0..N-1 of the matching expert, controlnet2 is a copy made before any
        ControlNet weights are loaded (GF:559-568).  The reference's call INF:81-106 runs as written.
        `redirect_common_files` (GF:498-510): a config given as (model_id, origin_file_pattern) whose pattern is one of the files
        every Wan model shares — the umT5 encoder, the VAE, the CLIP encoder — is looked up under the repository the reference
        redirects it to (`Wan-AI/Wan2.1-T2V-1.3B`, resp. `Wan-AI/Wan2.1-I2V-14B-480P`), which is where its download step puts them:
        the training script names them under `Wan-AI/Wan2.2-I2V-A14B` (scripts/train/train_goal_force.sh) and relies on this."""
        if redirect_common_files:
            redirect = {"models_t5_umt5-xxl-enc-bf16.pth": "Wan-AI/Wan2.1-T2V-1.3B", "Wan2.1_VAE.pth": "Wan-AI/Wan2.1-T2V-1.3B",
                        "models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth": "Wan-AI/Wan2.1-I2V-14B-480P"}
            for mc in model_configs:
                if mc.origin_file_pattern is None or mc.model_id is None:
                    continue
                if isinstance(mc.origin_file_pattern, str) and mc.origin_file_pattern in redirect and mc.model_id != redirect[mc.origin_file_pattern]:
                    print(f"To avoid repeatedly downloading model files, ({mc.model_id}, {mc.origin_file_pattern}) is redirected to "
                          f"({redirect[mc.origin_file_pattern]}, {mc.origin_file_pattern}). You can use `redirect_common_files=False` to disable file redirection.")
                    mc.model_id = redirect[mc.origin_file_pattern]
        if apply_strided_controlnet:
            raise NotImplementedError("strided ControlNet is not part of the Goal-Force sampling path")
        if audio_processor_config is not None:
            raise NotImplementedError("audio processor (S2V) is a pipeline branch Goal Force never takes")
        pipe = WanVideoPipeline(device=device, torch_dtype=torch_dtype, controlnet=controlnet,
                                controlnet_num_layers=controlnet_num_layers, controlnet_stride=controlnet_stride)
        dits = []
        for mc in model_configs:
            kind, module = load_model(mc, torch_dtype=mc.offload_dtype or torch_dtype, device=device)
            if kind == "wan_video_dit":
                dits.append(module)
            elif kind == "wan_video_text_encoder":
                pipe.text_encoder = module
            elif kind == "wan_video_vae":
                pipe.vae = module
        if len(dits) > 2:
            raise GoalForceError(f"{len(dits)} DiT checkpoints given: the pipeline holds a high-noise and a low-noise expert")
        # GF:516/594 `use_usp`: head-parallel attention over the whole process group (sequence_parallel.py); the
        # pipeline's __call__ / denoise then pass use_unified_sequence_parallel to model_fn (GF:1107-1109)
        pipe.use_unified_sequence_parallel = bool(use_usp)
        if dits:
            pipe.dit = dits[0]
            pipe.dit2 = dits[1] if len(dits) > 1 else None
        if controlnet:
            if pipe.dit is None:
                raise GoalForceError("controlnet=True needs a DiT expert among model_configs (its blocks seed the ControlNet)")
            pipe.init_controlnets()
        pipe._after_models_attached()
        if isinstance(tokenizer_config, str) and tokenizer_config == "default":
            # the reference's default argument (GF:486): the umT5 tokenizer of Wan2.1-T2V-1.3B under ./models — what the training script
            # relies on (train.py:52-55 passes no tokenizer_config).  The reference would download it when absent; here an absent folder
            # leaves the prompter without a tokenizer, and the first prompt that needs one says so.
            tokenizer_config = ModelConfig(model_id="Wan-AI/Wan2.1-T2V-1.3B", origin_file_pattern="google/*")
            try:
                tokenizer_config.download_if_necessary(use_usp=use_usp)
            except GoalForceError:
                tokenizer_config = None
        if tokenizer_config is not None:
            tokenizer_config.download_if_necessary(use_usp=use_usp)
            pipe.prompter.fetch_tokenizer(tokenizer_config.path)                     # GF:586
        return pipe

    @staticmethod
    def from_modules(dit, dit2=None, controlnet=None, controlnet2=None, vae=None, device="cuda"):
        """Assemble a pipeline from already-built modules (synthetic weights: tests, bench)."""
        pipe = WanVideoPipeline(device=device, controlnet=controlnet is not None,
                                controlnet_num_layers=0 if controlnet is None else controlnet.num_layers)
        pipe.dit, pipe.dit2, pipe.controlnet, pipe.controlnet2, pipe.vae = dit, dit2, controlnet, controlnet2, vae
        pipe._after_models_attached()
        return pipe

    def _after_models_attached(self):
        """GF:580-582 size division factors follow the VAE; GF:585-586 the prompter learns its text encoder."""
        if self.vae is not None:
            self.height_division_factor = self.width_division_factor = self.vae.upsampling_factor * 2
        if self.text_encoder is not None:
            self.prompter.fetch_models(self.text_encoder)

    def init_controlnets(self):
        d = self.dit
        kw = dict(dim=d.dim, num_heads=d.num_heads, ffn_dim=d.blocks[0].ffn_dim, eps=d.eps)
        self.controlnet = ControlNet(self.num_layers, stride=None, **kw).to(dtype=self.torch_dtype, device=self.device)
        for i in range(self.num_layers):
            self.controlnet.controlnet_dit.blocks[i].load_state_dict(d.blocks[i].state_dict())
        if self.dit2 is not None:
            self.controlnet2 = copy.deepcopy(self.controlnet)
            for i in range(self.num_layers):
                self.controlnet2.controlnet_dit.blocks[i].load_state_dict(self.dit2.blocks[i].state_dict())

    def load_controlnet_weights(self, module, path, torch_dtype=torch.bfloat16):
        """GF:176-178 — strips the 'pipe.controlnet.' prefix, strict load."""
        sd = load_state_dict(path, torch_dtype=torch_dtype)
        module.load_state_dict({k.replace("pipe.controlnet.", "", 1): v for k, v in sd.items()}, strict=True)

    def training_loss(self, **inputs):
        """GF:180-193 — same keyword inputs as the reference (`input_latents`, `noise`, `context`, `y`,
        `control_signal_video_latents`, optional `max_timestep_boundary` / `min_timestep_boundary`); returns the loss with
        the HIP backward behind it (goal_force_amd/training.py).  The scheduler must be in training mode
        (`set_timesteps(1000, training=True)`, utils.py:560)."""
        from . import training
        keys = ("input_latents", "noise", "context", "y", "control_signal_video_latents", "max_timestep_boundary",
                "min_timestep_boundary", "timestep_id")
        return training.training_loss(self, **{k: inputs[k] for k in keys if k in inputs})

    def enable_vram_management(self, num_persistent_param_in_dit=None, vram_limit=None, vram_buffer=0.5):
        """GF:196-452 — accepted for interface parity; everything is already resident in HBM."""
        self.vram_management_enabled = True

    def load_models_to_device(self, model_names=()):
        return None  # no offload state machine on 288 GB parts

    def enable_cpu_offload(self):
        """UTIL:125-127 (deprecated there in favour of enable_vram_management): accepted, nothing is offloaded."""
        self.vram_management_enabled = True

    def get_vram(self):
        """UTIL:130-131 — total memory of the pipeline's device in GiB."""
        return torch.cuda.mem_get_info(self.device)[1] / (1024 ** 3)

    def to(self, *args, **kwargs):
        """UTIL:33-40 — the pipeline's own `device` / `torch_dtype` follow, then every child module is moved."""
        device, dtype, _, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype is not None and dtype != torch.bfloat16:
            raise GoalForceError("the HIP kernels compute in bf16 storage / fp32 accumulate only")
        if device is not None:
            self.device = device
        super().to(*args, **kwargs)
        return self

    def freeze_except(self, model_names):
        """UTIL:134-141 — what `switch_pipe_to_training_mode` calls with --trainable_models (utils.py:563)."""
        for name, model in self.named_children():
            if name in model_names:
                model.train()
                model.requires_grad_(True)
            else:
                model.eval()
                model.requires_grad_(False)

    def blend_with_mask(self, base, addition, mask):
        """UTIL:144-145."""
        return base * (1 - mask) + addition * mask

    def step(self, scheduler, latents, progress_id, noise_pred, input_latents=None, inpaint_mask=None, **kwargs):
        """UTIL:148-154."""
        timestep = scheduler.timesteps[progress_id]
        if inpaint_mask is not None:
            noise_pred_expected = scheduler.return_to_timestep(timestep, latents, input_latents)
            noise_pred = self.blend_with_mask(noise_pred_expected, noise_pred, inpaint_mask)
        return scheduler.step(noise_pred, timestep, latents)

    # ------------------------------------------------------------------ host helpers (UTIL)
    def check_resize_height_width(self, height, width, num_frames=None):
        """UTIL:41-57 (round up to the division factors)."""
        if height % self.height_division_factor != 0:
            height = (height + self.height_division_factor - 1) // self.height_division_factor * self.height_division_factor
        if width % self.width_division_factor != 0:
            width = (width + self.width_division_factor - 1) // self.width_division_factor * self.width_division_factor
        if num_frames is None:
            return height, width
        if num_frames % self.time_division_factor != self.time_division_remainder:
            num_frames = (num_frames + self.time_division_factor - 1) // self.time_division_factor \
                * self.time_division_factor + self.time_division_remainder
        return height, width, num_frames

    def generate_noise(self, shape, seed=None, rand_device="cpu", rand_torch_dtype=torch.float32, device=None,
                       torch_dtype=None):
        """UTIL:117-122 — CPU generator, fp32 randn, then cast/move."""
        generator = None if seed is None else torch.Generator(rand_device).manual_seed(seed)
        noise = torch.randn(shape, generator=generator, device=rand_device, dtype=rand_torch_dtype)
        return noise.to(dtype=torch_dtype or self.torch_dtype, device=device or self.device)

    def frames_uint8(self, vae_output, min_value=-1, max_value=1):
        """[1,3,T,H,W] in [-1,1] -> [T,H,W,3] uint8 on the same device.  UTIL:76-91 computes `(x - min) * (255 / (max -
        min))`, the clip and the truncating uint8 cast in the VAE output's own dtype (bf16 here, so 200.7 becomes 201
        before the cast): same arithmetic, same bytes (tests/golden/g10 `frames_u8`)."""
        v = vae_output.mean(dim=0).permute(1, 2, 3, 0)
        return ((v - min_value) * (255 / (max_value - min_value))).clip(0, 255).to(torch.uint8)

    def vae_output_to_video(self, vae_output, pattern="B C T H W", min_value=-1, max_value=1):
        """UTIL:85-91 — list of PIL frames."""
        from PIL import Image
        if pattern != "B C T H W":
            raise NotImplementedError(f"vae_output_to_video(pattern={pattern!r}): the pipeline only produces 'B C T H W'")
        return [Image.fromarray(f.numpy()) for f in self.frames_uint8(vae_output, min_value, max_value).cpu()]

    def vae_output_to_image(self, vae_output, pattern="B C H W", min_value=-1, max_value=1):
        """UTIL:76-83 — one PIL image from [B,C,H,W] (mean over B), the scale / clip / truncating cast in the tensor's own dtype."""
        from PIL import Image
        if pattern == "B C H W":
            vae_output = vae_output.mean(dim=0).permute(1, 2, 0)
        elif pattern != "H W C":
            raise NotImplementedError(f"vae_output_to_image(pattern={pattern!r})")
        img = ((vae_output - min_value) * (255 / (max_value - min_value))).clip(0, 255)
        return Image.fromarray(img.to(device="cpu", dtype=torch.uint8).numpy())

    # ------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def denoise(self, latents, context_posi, context_nega, y, control_signal_video_latents, num_inference_steps=50,
                cfg_scale=5.0, switch_DiT_boundary=0.875, sigma_shift=5.0, denoising_strength=1.0, controlnet=True,
                progress_bar_cmd=None, record_step_times=False, step_ids=None, cfg_parallel=None,
                sequence_parallel=None):
        """GF:663 + GF:697-723.  Returns the final latents [1,16,f,H/8,W/8] (a new tensor).
        `step_ids` (optional) restricts the loop to a sub-range of the schedule (benchmarks).
        `cfg_parallel` (distributed.CfgPairParallel): this rank computes only its branch of the CFG pair and
        exchanges the noise prediction with its partner once per step (RCCL all-gather, 4.2 MB).
        `sequence_parallel` (sequence_parallel.SequenceParallel): every forward runs on this rank's token chunk with
        head-parallel attention; the noise prediction comes back whole on every rank of the group."""
        self.scheduler.set_timesteps(num_inference_steps, denoising_strength=denoising_strength, shift=sigma_shift)
        latents = latents.clone()
        models = {"dit": self.dit, "controlnet": self.controlnet if controlnet else None}
        caches = {}
        self.last_step_ms = []
        ids = range(len(self.scheduler.timesteps)) if step_ids is None else step_ids
        it = ids if progress_bar_cmd is None else progress_bar_cmd(ids)
        for progress_id in it:
            timestep = self.scheduler.timesteps[progress_id]
            # expert switch (GF:699-704): both experts are resident, so this is a pointer swap
            if timestep.item() < switch_DiT_boundary * self.scheduler.num_train_timesteps and self.dit2 is not None \
                    and models["dit"] is not self.dit2:
                models["dit"] = self.dit2
                if controlnet:
                    models["controlnet"] = self.controlnet2
            ts = timestep.unsqueeze(0).to(dtype=self.torch_dtype, device=self.device)  # bf16-rounded (GF:707)
            key = id(models["dit"])
            if key not in caches:
                caches[key] = (ContextCache(), ContextCache())
            if record_step_times:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            shared = dict(latents=latents, timestep=ts, y=y, control_signal_video_latents=control_signal_video_latents,
                          elide_zero_controlnet=self.elide_zero_controlnet)
            if sequence_parallel is not None:
                shared["sequence_parallel"] = sequence_parallel
            elif getattr(self, "use_unified_sequence_parallel", False):
                shared["use_unified_sequence_parallel"] = True          # GF:1107-1109
            if cfg_parallel is not None and cfg_scale != 1.0:
                b = cfg_parallel.branch
                mine = self.model_fn(**models, **shared, context=context_nega if b else context_posi,
                                     context_cache=caches[key][b])
                posi, nega = cfg_parallel.exchange(mine)
            else:
                # both branches on this GPU: the context-independent half of block 0 is computed once (model_fn: cfg_shared)
                pair = {} if (cfg_scale != 1.0 and self.share_cfg_prefix and self.model_fn is model_fn_wan_video) else None
                extra = {} if pair is None else {"cfg_shared": pair}
                two = self.cfg_streams and cfg_scale != 1.0 and sequence_parallel is None
                if two:
                    main = torch.cuda.current_stream(self.device)
                    if self._cfg_side_stream is None:
                        self._cfg_side_stream = torch.cuda.Stream(self.device)
                    inputs_ready = torch.cuda.Event()
                    inputs_ready.record(main)
                posi = self.model_fn(**models, **shared, **extra, context=context_posi, context_cache=caches[key][0])
                nega = None
                if two:
                    # enqueued after the whole cond forward (so the events of its block-0 halves exist), running beside it
                    side = self._cfg_side_stream
                    side.wait_event(inputs_ready)
                    with torch.cuda.stream(side):
                        nega = self.model_fn(**models, **shared, **extra, context=context_nega, context_cache=caches[key][1])
                    main.wait_stream(side)
                    nega.record_stream(main)
                elif cfg_scale != 1.0:
                    nega = self.model_fn(**models, **shared, **extra, context=context_nega, context_cache=caches[key][1])
            sigma, sigma_ = self.scheduler.sigma_pair(self.scheduler.timesteps[progress_id])
            # noise_pred = nega + cfg*(posi - nega); latents += noise_pred*(sigma_next - sigma)  (GF:716, FM:81)
            ops.cfg_euler_step(latents, posi.contiguous(), None if nega is None else nega.contiguous(), cfg_scale,
                               float(sigma_ - sigma))
            if record_step_times:
                ev1.record()
                self.last_step_ms.append((ev0, ev1, models["dit"] is self.dit2))
        if record_step_times:
            torch.cuda.synchronize()
            self.last_step_ms = [(a.elapsed_time(b), low) for a, b, low in self.last_step_ms]
        return latents

    @torch.no_grad()
    def __call__(self, prompt: str = "", negative_prompt: Optional[str] = "", input_image=None, end_image=None,
                 input_video=None, denoising_strength: Optional[float] = 1.0, input_audio=None, audio_embeds=None,
                 audio_sample_rate=16000, s2v_pose_video=None, s2v_pose_latents=None, motion_video=None,
                 control_video=None, reference_image=None, camera_control_direction=None, camera_control_speed=1 / 54,
                 camera_control_origin=None, vace_video=None, vace_video_mask=None, vace_reference_image=None,
                 vace_scale=1.0, seed: Optional[int] = None, rand_device: Optional[str] = "cpu",
                 height: Optional[int] = 480, width: Optional[int] = 832, num_frames=81,
                 cfg_scale: Optional[float] = 5.0, cfg_merge: Optional[bool] = False,
                 switch_DiT_boundary: Optional[float] = 0.875, num_inference_steps: Optional[int] = 50,
                 sigma_shift: Optional[float] = 5.0, motion_bucket_id=None, tiled: Optional[bool] = True,
                 tile_size=(30, 52), tile_stride=(15, 26), sliding_window_size=None, sliding_window_stride=None,
                 tea_cache_l1_thresh=None, tea_cache_model_id="", progress_bar_cmd=None, controlnet=False,
                 control_signal_video=None,
                 # pre-computed conditioning (outputs of the pre-loop units GF:791-917 / GF:808)
                 context_posi=None, context_nega=None, y=None, control_signal_video_latents=None,
                 output_type="pil", cfg_parallel=None, sequence_parallel=None, gather_frames=False):
        for name, val in (("end_image", end_image), ("input_video", input_video), ("input_audio", input_audio),
                          ("audio_embeds", audio_embeds), ("s2v_pose_video", s2v_pose_video),
                          ("motion_video", motion_video), ("control_video", control_video),
                          ("reference_image", reference_image), ("camera_control_direction", camera_control_direction),
                          ("vace_video", vace_video), ("vace_reference_image", vace_reference_image),
                          ("motion_bucket_id", motion_bucket_id), ("sliding_window_size", sliding_window_size),
                          ("tea_cache_l1_thresh", tea_cache_l1_thresh)):
            if val is not None:
                raise NotImplementedError(f"`{name}` belongs to a pipeline branch Goal Force never takes (SURVEY §2 #2)")
        if cfg_merge:
            raise NotImplementedError("cfg_merge=True: the reference default (two sequential forwards) is what is built")
        height, width, num_frames = self.check_resize_height_width(height, width, num_frames)
        length = (num_frames - 1) // 4 + 1
        noise = self.generate_noise((1, 16, length, height // 8, width // 8), seed=seed, rand_device=rand_device)
        if context_posi is None or (cfg_scale != 1.0 and context_nega is None):
            from .text_encoder import WanPrompter
            if isinstance(self.prompter, WanPrompter):
                if self.text_encoder is None or self.prompter.tokenizer is None:
                    raise GoalForceError("no text encoder / tokenizer loaded: attach pipe.text_encoder (WanTextEncoder) and "
                                         "pipe.prompter.fetch_tokenizer(path), or pass context_posi/context_nega")
                self.prompter.fetch_models(self.text_encoder)
            # WanVideoUnit_PromptEmbedder (GF:808-822) through the unit runner: any object with the reference's
            # `encode_prompt(prompt, positive=, device=)` works as pipe.prompter.  `positive` arrives as None for BOTH prompts in the
            # reference — the unit maps it from inputs_posi / inputs_nega, which never hold it (GF:812-813; pinned by g13) — and only
            # feeds prompt refiners, of which Goal Force loads none.
            context_posi = self.prompter.encode_prompt(prompt, positive=None, device=self.device)
            if cfg_scale != 1.0:                          # UTIL:262-271: the negative side is only computed under CFG
                context_nega = self.prompter.encode_prompt(negative_prompt, positive=None, device=self.device)
        if y is None and input_image is not None:
            y = self.embed_image(input_image, num_frames, height, width, tiled, tile_size, tile_stride)
        if y is None and self.dit is not None and self.dit.require_vae_embedding and self.dit.in_dim > noise.shape[1]:
            # without an image GF:896 returns {} and the reference's patch embedding then fails on the channel count;
            # the GEMM here K-pads its operand, so the missing channels must be refused, not read as zeros
            raise GoalForceError(f"this expert's patch embedding takes {self.dit.in_dim} channels: pass input_image "
                                 "(with pipe.vae attached) or a pre-computed y")
        if controlnet and control_signal_video_latents is None:
            if control_signal_video is None:
                raise GoalForceError("controlnet=True needs control_signal_video or control_signal_video_latents")
            control_signal_video_latents = self.embed_control_video(control_signal_video, tiled, tile_size, tile_stride)
        latents = self.denoise(noise, context_posi, context_nega, y, control_signal_video_latents,
                               num_inference_steps=num_inference_steps, cfg_scale=cfg_scale,
                               switch_DiT_boundary=switch_DiT_boundary, sigma_shift=sigma_shift,
                               denoising_strength=denoising_strength, controlnet=controlnet,
                               progress_bar_cmd=progress_bar_cmd, cfg_parallel=cfg_parallel,
                               sequence_parallel=sequence_parallel)
        if output_type == "latent":
            return latents
        if self.vae is None:
            raise GoalForceError("no VAE loaded: call with output_type='latent' or attach pipe.vae")
        # multi-GPU (distributed.CfgPairParallel): both ranks of the CFG pair hold these latents -> they share the tiles
        video = self.vae.decode(latents, device=self.device, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride,
                                tile_group=None if cfg_parallel is None else cfg_parallel.pair_group)
        if gather_frames:
            # end-of-run all-gather of every sample's decoded frames among the samples' lead ranks (RCCL / xGMI), then one broadcast inside each sample's group,
            # 81x480x832x3 uint8 = 97 MB per sample -> list (one [T,H,W,3] uint8 tensor per sample) on every rank
            if cfg_parallel is None:
                return [self.frames_uint8(video)]
            u8 = self.frames_uint8(video)
            return cfg_parallel.gather_frames(u8, tuple(u8.shape), torch.uint8, u8.device, everywhere=True)
        return video if output_type == "pt" else self.vae_output_to_video(video)

    # ---------------------------------------------------------------- pre-loop units that use the VAE encoder
    def preprocess_image(self, image, torch_dtype=None, device=None, pattern="B C H W", min_value=-1, max_value=1):
        """UTIL:60-66 — PIL -> [1,3,H,W] in [-1,1], the arithmetic in the target dtype (bf16) like the reference."""
        if pattern != "B C H W":
            raise NotImplementedError(f"preprocess_image(pattern={pattern!r}): only the default layout is used on this path")
        t = torch.Tensor(np.array(image, dtype=np.float32)).to(dtype=torch_dtype or self.torch_dtype, device=device or self.device)
        t = t * ((max_value - min_value) / 255) + min_value
        return t.permute(2, 0, 1).unsqueeze(0)

    def preprocess_video(self, video, torch_dtype=None, device=None, pattern="B C T H W", min_value=-1, max_value=1):
        """UTIL:69-73 — list of PIL frames -> [1,3,T,H,W] in [-1,1] (each frame through preprocess_image, stacked along T)."""
        if pattern != "B C T H W":
            raise NotImplementedError(f"preprocess_video(pattern={pattern!r})")
        return torch.stack([self.preprocess_image(f, torch_dtype, device, min_value=min_value, max_value=max_value) for f in video], dim=2)

    def embed_input_video(self, input_video, tiled, tile_size, tile_stride):
        """WanVideoUnit_InputVideoEmbedder (GF:767-789), the `input_video is not None` branch up to its VAE encode: the latents of
        the clip.  In training mode (scheduler.training) the unit returns them as `input_latents` next to `latents = noise`; in
        sampling mode it noises them to the first timestep (video-to-video, a branch Goal Force never takes)."""
        if self.vae is None:
            raise GoalForceError("embed_input_video needs pipe.vae")
        v = self.preprocess_video(input_video)[0]
        return self.vae.encode([v], device=self.device, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride)

    def embed_image(self, input_image, num_frames, height, width, tiled, tile_size, tile_stride):
        """WanVideoUnit_ImageEmbedderVAE, `end_image is None` branch (GF:887-917): y = [mask(4) ; vae.encode(video)(16)]
        where video = the image followed by num_frames-1 zero frames.  The reference builds the mask by repeating pixel
        frame 0 four times and folding every 4 pixel frames into the 4 channels of a latent frame (GF:899-909); with only
        frame 0 set that is: all 4 channels one on latent frame 0, zero elsewhere (pinned by tests/golden/g10)."""
        if self.vae is None:
            raise GoalForceError("embed_image needs pipe.vae")
        video = torch.zeros((3, num_frames, height, width), device=self.device, dtype=self.torch_dtype)
        video[:, 0] = self.preprocess_image(input_image.resize((width, height)))[0]
        lat = self.vae.encode([video], device=self.device, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride)[0]
        y = torch.zeros((1, 4 + lat.shape[0], *lat.shape[1:]), device=self.device, dtype=self.torch_dtype)
        y[0, :4, 0] = 1
        y[0, 4:] = lat
        return y

    def embed_control_video(self, control_signal_video, tiled, tile_size, tile_stride):
        """WanVideoUnit_ControlVideoEmbedder (GF:791-805): 'f h w c -> 1 c f h w' then vae.encode (un-rescaled [0,1])."""
        if self.vae is None:
            raise GoalForceError("embed_control_video needs pipe.vae")
        v = control_signal_video.to(dtype=self.torch_dtype, device=self.device).permute(3, 0, 1, 2)   # view, no copy
        return self.vae.encode([v], device=self.device, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride)


def build_random_expert(cfg=None, seed=0, device="cuda", std=0.02):
    """Random-init expert for benchmarks/tests: weights ~ N(0, std^2) generated ON the device (no
    checkpoints or network here).  norm weights 1, modulation as the reference initialises it."""
    cfg = dict(A14B_CONFIG if cfg is None else cfg)
    with torch.device("meta"):
        m = WanModel(**cfg)
    m = m.to_empty(device=device).to(torch.bfloat16)
    m.freqs = None
    g = torch.Generator(device=device).manual_seed(seed)
    for name, p in m.named_parameters():
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm3.weight"):
            p.data.fill_(1.0)
        elif name.endswith("norm3.bias"):
            p.data.zero_()
        elif name.endswith("modulation"):
            p.data.copy_(torch.randn(p.shape, generator=g, device=device, dtype=torch.float32) / cfg["dim"] ** 0.5)
        elif name.endswith(".bias"):
            p.data.copy_(torch.randn(p.shape, generator=g, device=device, dtype=torch.float32) * 0.01)
        else:
            p.data.copy_(torch.randn(p.shape, generator=g, device=device, dtype=torch.float32) * std)
    from .dit import precompute_freqs_cis_3d
    m.freqs = precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"])
    return m


def build_random_controlnet(num_layers=10, cfg=None, seed=1, device="cuda", std=0.02, zero_convs_zero=False):
    """Random-init ControlNet on the device (benchmarks/tests).  zero_convs_zero=True reproduces the
    reference's never-trained low-noise ControlNet2 (zero-convs exactly zero, GF:565)."""
    cfg = dict(A14B_CONFIG if cfg is None else cfg)
    with torch.device("meta"):
        cn = ControlNet(num_layers, dim=cfg["dim"], num_heads=cfg["num_heads"], ffn_dim=cfg["ffn_dim"], eps=cfg["eps"])
    cn = cn.to_empty(device=device).to(torch.bfloat16)
    g = torch.Generator(device=device).manual_seed(seed)
    for name, p in cn.named_parameters():
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm3.weight"):
            p.data.fill_(1.0)
        elif name.endswith("norm3.bias"):
            p.data.zero_()
        elif name.startswith("controlnet_zero_convs_after") and zero_convs_zero:
            p.data.zero_()
        elif name.endswith("modulation"):
            p.data.copy_(torch.randn(p.shape, generator=g, device=device, dtype=torch.float32) / cfg["dim"] ** 0.5)
        elif name.endswith(".bias"):
            p.data.copy_(torch.randn(p.shape, generator=g, device=device, dtype=torch.float32) * 0.01)
        else:
            p.data.copy_(torch.randn(p.shape, generator=g, device=device, dtype=torch.float32) * std)
    cn._is_zero = None
    return cn
