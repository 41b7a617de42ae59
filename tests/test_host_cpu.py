"""CPU-side tests: C-ABI library loads and exports every symbol the header declares, host mirrors keep
the reference's state_dict names, scheduler drop-in is bit-exact with the goldens, and the product
path refuses to run without a GPU (no silent fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN, ROOT

BF = torch.bfloat16


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "goalforce.h")).read()
    return sorted(set(re.findall(r"GF_API\s+[\w\s\*]+?\b(gf_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from goal_force_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 14 and sorted(_lib.SYMBOLS) == syms
    assert os.path.exists(_lib.LIB_PATH), "libgoalforce_hip.so missing — run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), s
    assert _lib.version().startswith("goalforce-hip") and _lib.load().gf_abi_version() >= 1


def test_state_dict_names_match_reference_layout():
    from goal_force_amd.controlnet import ControlNet
    from goal_force_amd.dit import WanModel
    cfg = gi.TINY
    m = WanModel(has_image_input=False, require_clip_embedding=False, **cfg)
    ref_sd = gi.dit_sd(cfg, seed=41)  # these keys were loaded strict=True into the reference WanModel
    assert sorted(m.state_dict().keys()) == sorted(ref_sd.keys())
    m.load_state_dict(ref_sd, strict=True)
    cn = ControlNet(2, dim=cfg["dim"], num_heads=cfg["num_heads"], ffn_dim=cfg["ffn_dim"])
    ref_cn = gi.controlnet_sd(cfg, 2, seed=42)
    assert sorted(cn.state_dict().keys()) == sorted(ref_cn.keys())
    cn.load_state_dict(ref_cn, strict=True)
    assert not cn.all_zero()
    cn.load_state_dict(gi.controlnet_sd(cfg, 2, seed=42, zero_convs_zero=True), strict=True)
    assert cn.all_zero()
    # freshly constructed ControlNet has zero zero-convs (GF:113-116)
    assert ControlNet(1, dim=256, num_heads=2, ffn_dim=512).all_zero()


def test_a14b_config_and_param_count():
    from goal_force_amd.dit import A14B_CONFIG, DiTBlock
    assert (A14B_CONFIG["dim"], A14B_CONFIG["num_heads"], A14B_CONFIG["ffn_dim"], A14B_CONFIG["num_layers"]) == \
        (5120, 40, 13824, 40)
    with torch.device("meta"):
        blk = DiTBlock(False, 5120, 40, 13824, 1e-6)
    assert sum(p.numel() for p in blk.parameters()) == 351_394_304


def test_scheduler_dropin_bit_exact():
    from goal_force_amd.scheduler import FlowMatchScheduler
    g = np.load(os.path.join(GOLDEN, "g1_scheduler.npz"))
    s = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    for n in (50, 4, 3):
        s.set_timesteps(n, denoising_strength=1.0, shift=5.0)
        assert np.array_equal(s.sigmas.numpy(), g[f"sigmas_{n}"]) and np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])
    s.set_timesteps(50, denoising_strength=1.0, shift=5.0)
    sample, mo = torch.from_numpy(g["sample"]), torch.from_numpy(g["model_output"])
    for i in (0, 10, 49):
        assert np.array_equal(s.step(mo, s.timesteps[i], sample).numpy(), g[f"step_f32_{i}"])
        got = s.step(mo.to(BF), s.timesteps[i], sample.to(BF))  # host bf16 path
        assert torch.equal(got, gi.from_u16(g[f"step_bf16_{i}"]))
    assert int((s.timesteps >= 875).sum()) == 21


def test_product_path_refuses_cpu_tensors():
    from goal_force_amd import GoalForceError, ops
    a = torch.zeros((8, 64), dtype=BF)
    with pytest.raises(GoalForceError, match="GPU"):
        ops.gemm(a, a)
    with pytest.raises(GoalForceError, match="GPU"):
        ops.layernorm_modulate(a)
    with pytest.raises(GoalForceError, match="GPU"):
        ops.flash_attn(a, a, a, 1)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "goal_force_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


def test_pipeline_call_signature_matches_reference_kwargs():
    import inspect
    from goal_force_amd.pipeline import WanVideoPipeline
    params = inspect.signature(WanVideoPipeline.__call__).parameters
    for name, default in (("cfg_scale", 5.0), ("switch_DiT_boundary", 0.875), ("num_inference_steps", 50),
                          ("sigma_shift", 5.0), ("tile_size", (30, 52)), ("tile_stride", (15, 26)),
                          ("rand_device", "cpu"), ("height", 480), ("width", 832), ("num_frames", 81),
                          ("tiled", True), ("controlnet", False), ("control_signal_video", None)):
        assert name in params and params[name].default == default, name


def test_param_key_survives_inference_tensors_and_sees_edits():
    """dit.param_key keys every derived weight copy: it must not raise on inference tensors (no version counter), must change
    on in-place edits of ordinary tensors and on invalidate_caches()."""
    import torch
    from goal_force_amd import dit
    w = torch.zeros(8)
    k0 = dit.param_key(w)
    w.add_(1)
    assert dit.param_key(w) != k0
    with torch.inference_mode():
        wi = torch.ones(8)
    ki = dit.param_key(wi)                      # used to raise: "Inference tensors do not track version counter"
    assert ki == dit.param_key(wi)
    dit.invalidate_caches()
    assert dit.param_key(wi) != ki


def test_a4_accumulator_handoff_is_clean_in_the_built_kernel():
    """tools/check_a4_agpr.py on the shipped source: between the asm K loop and the last v_accvgpr_read of every
    gemm_a4_kernel instantiation nothing but those reads touches an AGPR (the compiler sees them as free there)."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_a4_agpr.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok  ") >= 13


def test_bf16_quotient_equals_product_with_the_rounded_reciprocal():
    """What gf_vae.hip::rmsnorm_silu_kernel relies on when it replaces x / norm (one IEEE division per value) by x * (1 / norm)
    (one division per row): for bf16 x and norm, bf16(fp32(x / n)) == bf16(fp32(x * fp32(1 / n))).  The quotient of two 8-bit
    significands is either exactly a bf16 number or at least 2^-17 away from every bf16 rounding boundary, so the 2^-23 error of
    the product cannot flip the rounding.  Checked here for EVERY pair of significands over a range of exponents."""
    import torch
    BF = torch.bfloat16
    mant = torch.arange(128, 256, dtype=torch.float32) / 128.0                      # all normalised 8-bit significands in [1, 2)
    exps = 2.0 ** torch.arange(-6, 7, dtype=torch.float32)
    x = (mant[:, None] * exps[None, :]).reshape(-1)
    x = torch.cat([x, -x, torch.zeros(1)])
    n = (mant[:, None] * exps[None, :]).reshape(-1)
    assert torch.equal(x.to(BF).float(), x) and torch.equal(n.to(BF).float(), n)   # all exactly bf16
    q = (x[:, None] / n[None, :]).to(BF)
    p = (x[:, None] * (1.0 / n)[None, :]).to(BF)
    assert torch.equal(q.view(torch.int16), p.view(torch.int16))


def test_attention_backward_workspace_size_is_host_arithmetic():
    """gf_flash_attn_bwd_workspace_bytes touches no device: rowsum(dO.O) [q, heads] fp32 + one (-lse | -delta) record of 64 floats per
    32-query granule and head, each part rounded up to 256 bytes; zero for empty problems.  (A -DGF_BWD_QSCALE=1 build adds the
    pre-scaled copy of Q, [q, heads * 128] bf16: measured in round 4, not shipped.)"""
    from goal_force_amd import _lib
    lib = _lib.load()
    f = lib.gf_flash_attn_bwd_workspace_bytes
    assert f(0, 5, 1) == 0 and f(5, 0, 1) == 0 and f(5, 5, 0) == 0
    for q, kv, h in ((1, 1, 1), (32, 7, 2), (33, 4000, 8), (32760, 32760, 40)):
        n = f(q, kv, h)
        pad64 = -(-q // 64) * 64
        want = -(-(q * h * 4) // 256) * 256 + -(-(h * pad64 * 2 * 4) // 256) * 256
        assert n == want and n % 256 == 0, (q, kv, h, n, want)


def test_scheduler_remaining_methods_match_the_reference():
    """FlowMatchScheduler.return_to_timestep (FM:85-91) and calculate_shift (FM:114-125) against the reference's own (g1)."""
    from goal_force_amd.scheduler import FlowMatchScheduler
    g = np.load(os.path.join(GOLDEN, "g1_scheduler.npz"))
    sch = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(50, denoising_strength=1.0, shift=5.0)
    sample, stab = torch.from_numpy(g["sample"]), torch.from_numpy(g["stabilised"])
    for i in (0, 10, 49):
        assert torch.equal(sch.return_to_timestep(sch.timesteps[i], sample, stab), torch.from_numpy(g[f"return_f32_{i}"]))
    assert [sch.calculate_shift(n) for n in (256, 1560, 8192, 32760)] == g["calculate_shift"].tolist()


def test_training_keep_level_is_validated():
    import goal_force_amd.training as tr
    from goal_force_amd._lib import GoalForceError
    assert tr.KEEP_AUTO and tr.KEEP_ATTENTION and not tr.KEEP_WIDE      # default: wide only where the device has room for it
    assert not tr._wide_fits(torch.zeros(4, 8))                         # host tensor: no device to ask -> the 17 GB setting
    with pytest.raises(GoalForceError):
        tr.set_keep_level("everything")
    try:
        assert tr.set_keep_level("attn") == "auto" and tr.KEEP_ATTENTION and not tr.KEEP_WIDE and not tr.KEEP_AUTO
        assert tr.set_keep_level("wide") == "attn" and tr.KEEP_WIDE and tr._wide_fits(torch.zeros(4, 8))
        assert tr.set_keep_level("none") == "wide" and not tr.KEEP_ATTENTION
    finally:
        tr.set_keep_level("auto")


# environment variables the package may read: where the library is, and the launcher's rendezvous (torch.distributed.run's own names)
ENV_WHITELIST = {"GOALFORCE_HIP_LIB", "GF_DIST_BACKEND", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                 "HSA_ENABLE_IPC_MODE_LEGACY"}


def test_package_reads_no_environment_variable_outside_the_whitelist():
    """no Python-level env selector of a code path.  Every `os.environ` / `getenv` use in goal_force_amd/*.py
    names a whitelisted variable (library location, the launcher's rendezvous, the transport) — found on the AST, not by grep."""
    import ast
    pkg = os.path.join(ROOT, "goal_force_amd")
    seen = set()
    for f in sorted(os.listdir(pkg)):
        if not f.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, f)).read(), f)
        parents = {c: n for n in ast.walk(tree) for c in ast.iter_child_nodes(n)}
        for node in ast.walk(tree):
            is_env = isinstance(node, ast.Attribute) and node.attr in ("environ", "getenv", "putenv") and \
                isinstance(node.value, ast.Name) and node.value.id in ("os", "_os")
            if not is_env:
                continue
            # the use must be os.environ.get("X") / .setdefault("X", ..) / os.environ["X"] / os.getenv("X") with a literal name
            cur, name = node, None
            for _ in range(3):
                cur = parents.get(cur)
                if isinstance(cur, ast.Call) and cur.args and isinstance(cur.args[0], ast.Constant):
                    name = cur.args[0].value
                    break
                if isinstance(cur, ast.Subscript) and isinstance(cur.slice, ast.Constant):
                    name = cur.slice.value
                    break
            assert name in ENV_WHITELIST, f"{f}:{node.lineno}: environment use outside the whitelist ({name!r})"
            seen.add(name)
    assert "GOALFORCE_HIP_LIB" in seen and "WORLD_SIZE" in seen, seen


def test_bindings_refuse_a_library_of_another_abi_revision(monkeypatch):
    """_lib.load() compares gf_abi_version() with the revision the bindings were written for (a stale .so would be
    overrun through gf_flash_attn_bwd's caller-owned workspace, whose size changed under an unchanged signature)."""
    from goal_force_amd import _lib
    assert _lib.load().gf_abi_version() == _lib.ABI_VERSION
    src = open(os.path.join(ROOT, "goal_force_amd", "csrc", "gf_abi.hip")).read()
    assert int(re.search(r"#define GF_ABI_VERSION (\d+)", src).group(1)) == _lib.ABI_VERSION
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.GoalForceError, match="ABI revision"):
        _lib.load()


def test_prompt_cleaner_uses_ftfy_when_importable(monkeypatch, capsys):
    """wan_prompter.py:11-14: ftfy.fix_text, html-unescape twice, strip, collapse whitespace.  With ftfy importable it is
    called first; without it the first call says so once on stderr."""
    import sys
    import types
    from goal_force_amd.text_encoder import WanPrompter
    fake = types.ModuleType("ftfy")
    fake.fix_text = lambda t: t.replace("\u201c", '"').replace("\u201d", '"')
    monkeypatch.setitem(sys.modules, "ftfy", fake)
    assert WanPrompter.clean("  a \u201cball\u201d  &amp;amp;  a\n cube ") == 'a "ball" & a cube'
    monkeypatch.setitem(sys.modules, "ftfy", None)            # import ftfy -> ImportError
    monkeypatch.setattr(WanPrompter, "ftfy_missing_warned", False)
    assert WanPrompter.clean("  a  &lt;ball&gt; ") == "a <ball>"
    WanPrompter.clean("again")
    assert capsys.readouterr().err.count("ftfy") >= 1 and WanPrompter.ftfy_missing_warned


@pytest.mark.skipif(torch.cuda.is_available(), reason="with a GPU the two ranks would really run (tests/test_bench_gpu.py covers that)")
def test_bench_self_launch_reports_a_failed_rank():
    """`python bench.py --gpus 2` with no launcher starts its own ranks; here (no GPU) both fail at torch.cuda.set_device and the
    parent must exit non-zero promptly, print no JSON line and leave no rank waiting in a rendezvous."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "1", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "stopping the other ranks" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_pad_run_finds_the_run_of_identical_rows_that_ends_a_context():
    """dit.pad_run (the detection behind the cross-attention's pad-key folding): index of the first row of the trailing run of
    identical rows; L - 1 when the last two rows differ (a run of one: nothing to fold), 0 when all rows are equal."""
    from goal_force_amd.dit import pad_run
    g = torch.Generator().manual_seed(0)
    x = torch.randn(12, 8, generator=g).to(BF)
    assert pad_run(x) == 11
    y = x.clone()
    y[5:] = y[5]
    assert pad_run(y) == 5
    y[7, 3] += 1                                   # a different row inside the tail: the run starts after it
    assert pad_run(y) == 8
    assert pad_run(x[:1]) == 0 and pad_run(x[3:4].expand(6, -1)) == 0
    z = torch.zeros(512, 16, dtype=BF)
    z[:40] = torch.randn(40, 16, generator=g).to(BF)
    assert pad_run(z) == 40                        # the prompter's layout: 40 real rows, 472 zero rows


def test_generated_k_loops_are_what_their_generators_emit(tmp_path):
    """The three asm K loops under goal_force_amd/csrc/*.inc are committed generator output (the build does not run the generators):
    regenerating them must reproduce the committed files byte for byte — a hand edit or a stale file fails here, and each generator's
    own schedule replay (register lifetimes, SCC chains, counted waits) runs as part of it."""
    import subprocess
    import sys
    for gen, env_key, inc in (("gen_gemm_a4.py", "A4_OUT", "gf_gemm_a4_loop.inc"), ("gen_gemm_a4f8.py", "A4F8_OUT", "gf_gemm_a4f8_loop.inc"),
                              ("gen_conv_a4.py", "CONV_A4_OUT", "gf_conv_a4_loop.inc")):
        out = tmp_path / inc
        env = {k: v for k, v in os.environ.items() if not k.startswith(("A4", "CONV_A4"))}
        env[env_key] = str(out)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert out.read_bytes() == open(os.path.join(ROOT, "goal_force_amd", "csrc", inc), "rb").read(), f"{inc} is not what tools/{gen} emits"


def test_experiment_patches_still_apply_to_the_product_sources():
    """tools/patches/*_experiments.patch hold the kernels and diagnostic builds that left the product sources (attention kernel 1, the
    sl GEMMs, the first backward kernels, the env-variable selectors).  They are diffs against the CURRENT product files: an edit to
    one of those files without `tools/patches/make_patches.sh` would silently orphan the experimental tree — so it is re-created here."""
    import shutil
    import subprocess
    if shutil.which("patch") is None:
        pytest.skip("no `patch` tool on this host")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "experimental_tree.sh")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FAILED" not in r.stdout + r.stderr, (r.stdout + r.stderr)[-2000:]
    x = os.path.join(ROOT, "build", "experimental", "csrc")
    assert "void flash_attn_fwd_kernel(const AttnArgs p)" in open(os.path.join(x, "gf_attention.hip")).read()
    assert "gemm_sl_kernel" in open(os.path.join(x, "gf_gemm.hip")).read() and "getenv" in open(os.path.join(x, "gf_abi.hip")).read()
    assert "getenv" not in open(os.path.join(ROOT, "goal_force_amd", "csrc", "gf_abi.hip")).read(), "the product library reads no environment variable"
    for f in os.listdir(os.path.join(ROOT, "goal_force_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "goal_force_amd", "csrc", f)).read(), f


def test_padded_conv_path_is_refused_for_shapes_beyond_its_32_bit_offsets():
    """vae.py sends a convolution to the padded-layout kernel only when its zero-bordered input stays below 4 GiB: a production tile
    does at every level; the untiled 480 x 832 decode does not at the 384 -> 192 resample convolution and stays on the implicit GEMM."""
    from goal_force_amd import ops
    assert ops.padded_conv_fits(81, 120, 208, 192) and ops.padded_conv_fits(41, 60, 104, 384) and ops.padded_conv_fits(21, 30, 52, 384)
    assert ops.padded_conv_fits(81, 120, 208, 384, history=False)            # the tile's resample convolution (1.6 GB)
    assert ops.padded_conv_fits(81, 240, 416, 192)                           # untiled level 2: 3.2 GB
    assert not ops.padded_conv_fits(81, 240, 416, 384, history=False)        # untiled resample convolution: 6.3 GB
    assert not ops.padded_conv_fits(81, 120, 208, 96)                        # 96 channels: not this kernel's level


def test_launcher_takes_the_reference_launchers_arguments():
    """scripts/inference_goal_force.py parses exactly what scripts/inference/inference_goal_force.sh / inference_canny_edge_control.sh
    pass (INF:38-56), derives the output directory and the file-name root as INF:73-76, 178-186, and defaults the model paths to the
    ones the reference hard-codes (INF:81-106)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gf_launcher", os.path.join(ROOT, "scripts", "inference_goal_force.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = m.parse_args(["--device_id", "3", "--world_size", "8", "--seed", "5", "--control_signal_type", "canny_edge",
                      "--model_ckpt_path", "checkpoints/wan2.2_controlnet_canny_edge/step-500.safetensors", "--example_paths", "a.csv", "b.csv"])
    assert (a.device_id, a.world_size, a.seed, a.control_signal_type, a.example_paths) == (3, 8, 5, "canny_edge", ["a.csv", "b.csv"])
    assert m.output_location(a) == ("checkpoints/wan2.2_controlnet_canny_edge/step-500-videos", "500")
    assert len(a.dit_high) == 6 and a.dit_high[5].endswith("high_noise_model/diffusion_pytorch_model-00006-of-00006.safetensors")
    assert a.vae == "./models/Wan-AI/Wan2.1-T2V-1.3B/Wan2.1_VAE.pth" and a.tokenizer.endswith("google/umt5-xxl")
    assert m.parse_args(["--controlnet_checkpoint", "x/step-7.safetensors", "--example_paths", "a.csv"]).model_ckpt_path == "x/step-7.safetensors"
    with pytest.raises(SystemExit):
        m.parse_args(["--example_paths", "a.csv"])                       # --model_ckpt_path is required (INF:51) unless --synthetic
    with pytest.raises(SystemExit):
        m.parse_args(["--control_signal_type", "depth", "--synthetic", "--example_paths", "a.csv"])
    data = dict(file_id="scene", x_pos=0.4423, y_pos=0.225, target_x_pos=0.655, target_y_pos=0.2375, masses={"projectile": -1, "target": 2.0},
                force=-1.0, angle=-1.0, target_indirect_force=350.0, target_indirect_angle=0.0)
    assert m.goal_force_name("500", data, 3) == ("step-500_scene__prj_coords_0.44_0.23__tgt_coords_0.66_0.24__prj_mass_-1.0__tgt_mass_2.0"
                                                  "__prj_force_-1.0__prj_angle_-1.0__tgt_indirect_force_350.0__tgt_indirect_angle_0.0__seed_3")
    assert (m.NUM_FRAMES, m.NUM_FRAMES_CANNY, m.CONTROLNET_NUM_LAYERS) == (81, 49, 10)


def test_canny_video_operator_crops_resizes_and_counts_frames_like_the_reference(tmp_path):
    """ControlSignalDataset_CannyEdge.default_video_operator (DS:441-461): ImageCropAndResize (DS:136-170) and LoadVideo's frame count
    (DS:188-194) restated on PIL; containers readable here (frame directory, .npy, image); mp4 refused by name without imageio."""
    from PIL import Image
    from goal_force_amd._lib import GoalForceError
    from goal_force_amd.canny import ControlSignalDataset_CannyEdge
    rng = np.random.default_rng(0)
    clip = rng.integers(0, 255, (15, 50, 90, 3), dtype=np.uint8)
    np.save(tmp_path / "c.npy", clip)
    op = ControlSignalDataset_CannyEdge.default_video_operator(base_path=str(tmp_path), max_pixels=921600, height=48, width=80,
                                                               num_frames=49, time_division_factor=4, time_division_remainder=1)
    frames = op("c.npy")
    assert len(frames) == 13 and frames[0].size == (80, 48)              # 15 available -> 13 = 4k + 1
    # one frame by hand: scale = max(80/90, 48/50) = 0.96 -> resize to (round(90*.96), round(50*.96)) = (86, 48), crop 3 columns left
    want = Image.fromarray(clip[0]).resize((86, 48), Image.BILINEAR).crop((3, 0, 83, 48))
    assert np.array_equal(np.array(frames[0]), np.array(want))
    assert [op.count(n) for n in (100, 49, 48, 6, 5, 1)] == [49, 49, 45, 5, 5, 1]
    d = tmp_path / "dir"
    d.mkdir()
    for i in range(5):
        Image.fromarray(clip[i]).save(d / f"{i:03d}.png")
    got = op("dir")
    assert len(got) == 5 and np.array_equal(np.array(got[4]), np.array(Image.fromarray(clip[4]).resize((86, 48), Image.BILINEAR).crop((3, 0, 83, 48))))
    free = ControlSignalDataset_CannyEdge.default_video_operator(base_path=str(tmp_path), max_pixels=40 * 40)
    assert free("dir/000.png")[0].size == (48, 16)                        # 90x50 > 1600 px: scale 1.677 -> 53x29 -> multiples of 16
    try:
        import imageio  # noqa: F401
    except ImportError:
        (tmp_path / "v.mp4").write_bytes(b"")
        with pytest.raises(GoalForceError, match="imageio"):
            op("v.mp4")


def test_training_cli_and_dataset_mux_match_the_reference_objects(tmp_path):
    """g18 = the reference's own `wan_parser()` (utils.py:854-900) and `get_dataset(args)` (train.py:126-197) on a synthetic data tree:
    every option of the command line with its dest / default / type / nargs / required / action, and what the three training-mode sets read
    from their CSV listings (rows kept, force / goal-force / mass ranges, masking probabilities)."""
    from goal_force_amd import training as tr
    g = np.load(os.path.join(GOLDEN, "g18_training_cli.npz"))
    rows = []
    for a in tr.wan_parser()._actions:
        if not a.option_strings or a.dest == "help":
            continue
        rows.append([a.option_strings[0], a.dest, repr(a.default), getattr(a.type, "__name__", "None"), repr(a.nargs), repr(bool(a.required)),
                     type(a).__name__])
    assert rows == g["parser"].tolist()
    gi.write_training_tree(str(tmp_path))
    args = tr.wan_parser().parse_args(gi.training_cli(str(tmp_path)))
    ds = tr.get_dataset(args, device="cpu")
    b, d, p = ds.datasets
    assert [type(x).__name__ for x in ds.datasets] == ["ControlSignalDataset_Balls", "ControlSignalDataset_Dominos", "ControlSignalDataset_Plants"]
    assert [len(b), len(d), len(p), len(ds)] == g["lengths"].tolist()
    assert [b.min_force, b.max_force, b.min_indirect_force, b.max_indirect_force, b.min_mass, b.max_mass] == g["balls_ranges"].tolist()
    assert [d.min_force, d.max_force, d.min_indirect_force, d.max_indirect_force, d.min_mass, d.max_mass] == g["dominos_ranges"].tolist()
    assert [p.min_force, p.max_force] == g["plants_ranges"].tolist()
    assert [b.p_mask_out_direct_force, b.p_mask_out_indirect_force, b.p_mask_out_masses, d.p_mask_out_direct_force, d.p_mask_out_indirect_force,
            d.p_mask_out_masses] == g["masks"].tolist()
    assert [",".join(x.df[x.media_type].tolist()) for x in ds.datasets] == g["kept_rows"].tolist()
    assert not b.is_validation_dataset and b.media_type == "video"
    verdicts = []
    for name, item, nf in gi.shape_check_cases():
        try:
            verdicts.append(str(bool(tr.data_is_correct_shape_and_type(item, "direct_force_and_goal_force_and_mass", nf))))
        except Exception as e:      # noqa: BLE001
            verdicts.append(type(e).__name__)
    assert verdicts == g["shape_check_verdicts"].tolist(), list(zip([c[0] for c in gi.shape_check_cases()], verdicts, g["shape_check_verdicts"]))
    assert [repr(tr.safe_collate(x)) for x in ([None, None], [None, 7, 8], [3], [])] == g["safe_collate"].tolist()


def test_training_mode_frame_selection_video_loading_and_the_pixel_roundtrip(tmp_path):
    """Host logic of the training-mode sets that needs no GPU: which frames of a clip are trained on (balls `[::2][-n:]` DS:985, dominos
    `[14:][0:n]` DS:1463), the clip containers readable here, and the reference's ToTensor -> 2x-1 -> (x+1)/2 -> ToPILImage chain, which
    is NOT the identity in fp32: the levels 1 .. 63 come back one lower (goal_force_amd/force_map.py::reference_pixel_roundtrip)."""
    from PIL import Image
    from goal_force_amd import force_map as fm
    frames = list(range(200))
    b = object.__new__(fm.ControlSignalDataset_Balls)
    d = object.__new__(fm.ControlSignalDataset_Dominos)
    b.num_frames = d.num_frames = 81
    assert b.select_frames(frames) == list(range(38, 200, 2)) and d.select_frames(frames) == list(range(14, 95))
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    (back,) = fm.reference_pixel_roundtrip([Image.fromarray(ramp)])
    got = np.array(back)[..., 0].reshape(-1).astype(int)
    want = np.arange(256)
    want[1:64] -= 1
    assert got.tolist() == want.tolist(), "levels 1..63 lose one level in the reference's fp32 chain, every other level survives"
    clip = np.random.default_rng(0).integers(0, 255, (7, 8, 12, 3), dtype=np.uint8)
    np.save(tmp_path / "c.npy", clip)
    fr = fm.load_video_frames(str(tmp_path / "c.npy"))
    assert len(fr) == 7 and np.array_equal(np.array(fr[3]), clip[3])
    (tmp_path / "dir").mkdir()
    for i in range(3):
        Image.fromarray(clip[i]).save(tmp_path / "dir" / f"{i:02d}.png")
    fr = fm.load_video_frames(str(tmp_path / "dir"))
    assert len(fr) == 3 and np.array_equal(np.array(fr[2]), clip[2])


def test_pipeline_is_a_module_with_the_base_pipeline_helpers():
    """BasePipeline's API shell (UTIL:13-157; SURVEY §2 #5 "keep signatures"): the pipeline is a torch.nn.Module whose children are the
    models — `named_children`, `freeze_except` (what `--trainable_models controlnet` calls, utils.py:563), `to`, `step`,
    `blend_with_mask`, `vae_output_to_image`, and the reference's positional order of `preprocess_image` / `vae_output_to_video`."""
    import inspect
    from goal_force_amd._lib import GoalForceError
    from goal_force_amd.controlnet import ControlNet
    from goal_force_amd.dit import WanModel
    from goal_force_amd.pipeline import WanVideoPipeline
    cfg = gi.TINY
    dit = WanModel(has_image_input=False, require_clip_embedding=False, **cfg)
    cn = ControlNet(1, dim=cfg["dim"], num_heads=cfg["num_heads"], ffn_dim=cfg["ffn_dim"])
    pipe = WanVideoPipeline.from_modules(dit, None, cn, None, device="cpu")
    assert isinstance(pipe, torch.nn.Module) and [n for n, _ in pipe.named_children()] == ["dit", "controlnet"]
    pipe.freeze_except(["controlnet"])
    assert all(p.requires_grad for p in cn.parameters()) and not any(p.requires_grad for p in dit.parameters())
    assert cn.training and not dit.training and sum(p.numel() for p in pipe.parameters() if p.requires_grad) == sum(p.numel() for p in cn.parameters())
    pipe.freeze_except([])
    assert not any(p.requires_grad for p in pipe.parameters())
    assert pipe.to("cpu") is pipe and pipe.device == torch.device("cpu")
    with pytest.raises(GoalForceError):
        pipe.to(torch.float16)
    pipe.dit2 = None
    assert pipe.dit2 is None and pipe.dit is dit
    a, b, m = torch.full((2, 2), 2.0), torch.full((2, 2), 6.0), torch.tensor([[0.0, 1.0], [0.5, 0.25]])
    assert torch.equal(pipe.blend_with_mask(a, b, m), a * (1 - m) + b * m)
    pipe.scheduler.set_timesteps(4, shift=5.0)
    lat, pred = torch.randn(1, 16, 1, 2, 2).to(BF), torch.randn(1, 16, 1, 2, 2).to(BF)
    assert torch.equal(pipe.step(pipe.scheduler, lat, 2, pred), pipe.scheduler.step(pred, pipe.scheduler.timesteps[2], lat))
    x = (torch.rand(1, 3, 5, 7) * 2.4 - 1.2).to(BF)
    img = pipe.vae_output_to_image(x)
    want = ((x.mean(dim=0).permute(1, 2, 0) + 1) * (255 / 2)).clip(0, 255).to(torch.uint8).numpy()      # UTIL:80-82 in the tensor's dtype
    assert img.size == (7, 5) and np.array_equal(np.array(img), want)
    for name, order in (("preprocess_image", ["image", "torch_dtype", "device", "pattern", "min_value", "max_value"]),
                        ("preprocess_video", ["video", "torch_dtype", "device", "pattern", "min_value", "max_value"]),
                        ("vae_output_to_video", ["vae_output", "pattern", "min_value", "max_value"]),
                        ("vae_output_to_image", ["vae_output", "pattern", "min_value", "max_value"]),
                        ("generate_noise", ["shape", "seed", "rand_device", "rand_torch_dtype", "device", "torch_dtype"]),
                        ("step", ["scheduler", "latents", "progress_id", "noise_pred", "input_latents", "inpaint_mask", "kwargs"])):
        assert list(inspect.signature(getattr(pipe, name)).parameters) == order, name
