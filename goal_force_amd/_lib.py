"""ctypes binding of libgoalforce_hip.so — the C ABI declared in include/goalforce.h.

The product path has NO fallback: if the shared library is missing or an entry point
fails, a GoalForceError is raised (never a silent torch/CPU substitute).
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgoalforce_hip.so")
# GOALFORCE_HIP_LIB: A/B a differently built libgoalforce_hip.so (kernel tuning); it is still the HIP library, never a fallback
LIB_PATH = os.environ.get("GOALFORCE_HIP_LIB", LIB_PATH)

# every symbol include/goalforce.h declares (tests check the .so exports exactly these)
SYMBOLS = (
    "gf_version", "gf_last_error", "gf_abi_version", "gf_set_option", "gf_get_option", "gf_reset_options",
    "gf_modulation", "gf_layernorm_modulate", "gf_rmsnorm_rope", "gf_gemm_bf16", "gf_flash_attn_fwd",
    "gf_patchify_im2col", "gf_unpatchify", "gf_cfg_euler_step", "gf_act", "gf_add_bf16",
    "gf_force_map",
    "gf_vae_prep_latent", "gf_vae_im2col", "gf_vae_finish_latent", "gf_vae_rmsnorm_silu", "gf_softmax_rows", "gf_rowmax_neg_bf16", "gf_transpose_pad",
    "gf_vae_tile_blend", "gf_vae_tile_finalize",
    "gf_quant_fp8_rowscale", "gf_cast_fp8", "gf_gemm_fp8", "gf_layernorm_modulate_fp8", "gf_modulate", "gf_gate_residual", "gf_rope_apply", "gf_linear_vt32_fp8",
    "gf_flash_attn_fwd_lse", "gf_flash_attn_bwd", "gf_flash_attn_bwd_workspace_bytes",
    "gf_layernorm_bwd", "gf_rmsnorm_rope_bwd", "gf_colsum", "gf_act_bwd", "gf_mse_loss", "gf_adamw_step", "gf_f32_to_bf16", "gf_sumsq",
    "gf_transpose_v", "gf_flash_attn_fwd_vt", "gf_transpose_v32", "gf_flash_attn_fwd_vt32", "gf_linear_vt32",
    "gf_conv3d_bf16", "gf_conv3d_padded_bf16", "gf_vae_rmsnorm_silu_padded", "gf_vae_upsample2x_padded", "gf_gemm_bf16_batched", "gf_transpose_pad_batched",
    "gf_resize_lanczos4_u8", "gf_resize_area_u8", "gf_canny_u8", "gf_flash_attn_fwd_lastmult",
)

# the C ABI revision these bindings were written against (csrc/gf_abi.hip: GF_ABI_VERSION).  A stale or foreign .so whose entry
# points take differently sized buffers (gf_flash_attn_bwd's workspace grew 3x between revisions 7 and 10 under an unchanged
# signature) is refused at load time instead of overrunning memory.
ABI_VERSION = 18

EPI_BIAS, EPI_BIAS_GELU_TANH, EPI_BIAS_GATE_RESID, EPI_BIAS_RESID, EPI_BIAS_SILU, EPI_BIAS_MUL = range(6)


class GoalForceError(RuntimeError):
    """Raised when the HIP library is missing or a gf_* call returns an error."""


_lib = None
_lock = threading.Lock()

_vp, _i64, _f32, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int


def _declare(lib):
    lib.gf_version.restype = ctypes.c_char_p
    lib.gf_version.argtypes = []
    lib.gf_last_error.restype = ctypes.c_char_p
    lib.gf_last_error.argtypes = []
    lib.gf_abi_version.restype = _int
    lib.gf_abi_version.argtypes = []
    lib.gf_set_option.restype = _int
    lib.gf_set_option.argtypes = [ctypes.c_char_p, _int]
    lib.gf_get_option.restype = _int
    lib.gf_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(_int)]
    lib.gf_reset_options.restype = None
    lib.gf_reset_options.argtypes = []
    sigs = {
        "gf_layernorm_modulate": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp],
        "gf_modulation": [_vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_uint32, _vp],
        "gf_rmsnorm_rope": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp],
        "gf_gemm_bf16": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp, _vp],
        "gf_flash_attn_fwd": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp],
        "gf_flash_attn_fwd_lastmult": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _vp],
        "gf_flash_attn_fwd_lse": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp],
        "gf_flash_attn_bwd": [_vp] * 10 + [_i64] * 12 + [_f32, _vp],
        "gf_flash_attn_bwd_workspace_bytes": [_i64] * 3,
        "gf_layernorm_bwd": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _f32, _vp],
        "gf_rmsnorm_rope_bwd": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _f32, _vp],
        "gf_colsum": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _vp],
        "gf_act_bwd": [_vp, _vp, _vp, _i64, _int, _vp],
        "gf_mse_loss": [_vp, _vp, _vp, _vp, _i64, _f32, _vp],
        "gf_adamw_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _vp],
        "gf_f32_to_bf16": [_vp, _vp, _i64, _vp],
        "gf_sumsq": [_vp, _i64, _vp, _vp],
        "gf_transpose_v": [_vp, _i64, _vp, _i64, _i64, _i64, _vp],
        "gf_flash_attn_fwd_vt": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp],
        "gf_transpose_v32": [_vp, _i64, _vp, _i64, _i64, _i64, _vp],
        "gf_linear_vt32": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
        "gf_flash_attn_fwd_vt32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp],
        "gf_patchify_im2col": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp],
        "gf_unpatchify": [_vp, _vp, _i64, _i64, _i64, _i64, _vp],
        "gf_cfg_euler_step": [_vp, _vp, _vp, _f32, _f32, _i64, _vp],
        "gf_act": [_vp, _vp, _i64, _int, _vp],
        "gf_add_bf16": [_vp, _vp, _vp, _i64, _vp],
        "gf_vae_prep_latent": [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
        "gf_vae_im2col": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _i64, _i64, _i64, _vp],
        "gf_conv3d_bf16": [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int,
                           _i64, _i64, _int, _vp, _i64, _vp],
        "gf_vae_finish_latent": [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp],
        "gf_vae_rmsnorm_silu": [_vp, _vp, _vp, _i64, _i64, _int, _vp],
        "gf_vae_rmsnorm_silu_padded": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp],
        "gf_conv3d_padded_bf16": [_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp],
        "gf_vae_upsample2x_padded": [_vp, _vp, _i64, _i64, _i64, _i64, _vp],
        "gf_softmax_rows": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _f32, _vp],
        "gf_rowmax_neg_bf16": [_vp, _i64, _vp, _i64, _i64, _i64, _vp],
        "gf_transpose_pad": [_vp, _i64, _vp, _i64, _i64, _i64, _vp],
        "gf_transpose_pad_batched": [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
        "gf_gemm_bf16_batched": [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp],
        "gf_vae_tile_blend": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int,
                              _i64, _i64, _vp],
        "gf_vae_tile_finalize": [_vp, _vp, _i64, _i64, _int, _vp],
        "gf_quant_fp8_rowscale": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
        "gf_modulate": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
        "gf_gate_residual": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
        "gf_linear_vt32_fp8": [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
        "gf_rope_apply": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
        "gf_layernorm_modulate_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp],
        "gf_cast_fp8": [_vp, _vp, _i64, _vp],
        "gf_gemm_fp8": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp, _vp],
        "gf_force_map": [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _int, _vp],
        "gf_resize_lanczos4_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
        "gf_resize_area_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp],
        "gf_canny_u8": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _int, _int, _vp],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.restype = _i64 if name.endswith("_bytes") else _int
        fn.argtypes = argtypes


def load():
    """Load (once) and return the ctypes handle; raises GoalForceError if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            # torch ships its own libamdhip64.so.7; import it FIRST so this library binds to the same HIP
            # runtime instance (two runtimes in one process cannot both see the device).
            import torch  # noqa: F401
            if not os.path.exists(LIB_PATH):
                raise GoalForceError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C goal_force_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
            try:
                lib = ctypes.CDLL(LIB_PATH)
            except OSError as e:  # pragma: no cover
                raise GoalForceError(f"cannot load {LIB_PATH}: {e}") from e
            # the revision is compared BEFORE the symbols are bound: a stale build usually lacks a symbol, and the rebuild hint
            # must reach the user instead of a bare AttributeError out of _declare
            rebuild = "rebuild the library (`make -C goal_force_amd/csrc`)"
            try:
                lib.gf_abi_version.restype = _int
                have = int(lib.gf_abi_version())
            except AttributeError as e:
                raise GoalForceError(f"{LIB_PATH} predates the C ABI revision export: {rebuild}") from e
            if have != ABI_VERSION:
                raise GoalForceError(f"{LIB_PATH} speaks C ABI revision {have}, these bindings revision {ABI_VERSION}: {rebuild} "
                                     "— mixing revisions can overrun caller-owned workspaces")
            try:
                _declare(lib)
            except AttributeError as e:
                raise GoalForceError(f"{LIB_PATH} lacks a symbol these bindings declare ({e}): {rebuild}") from e
            _lib = lib
    return _lib


def check(status: int, what: str):
    if status != 0:
        msg = load().gf_last_error().decode("utf-8", "replace")
        raise GoalForceError(f"{what} failed with status {status}: {msg}")


def version() -> str:
    return load().gf_version().decode()
