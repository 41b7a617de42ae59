"""End-to-end run of the inference driver at the real resolution (832x480x81f) with random weights and a reduced
depth/step count: CSV row -> force map (HIP) -> VAE encode of the force map and of the first frame (HIP) ->
denoise loop with ControlNet + CFG + expert switch (HIP) -> tiled VAE decode (HIP) -> 81 frames."""
import os
import subprocess
import sys

import numpy as np
import pytest

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
