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


def test_inference_driver_end_to_end(tmp_path):
    from PIL import Image
    ex = tmp_path / "example"
    (ex / "images").mkdir(parents=True)
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (480, 832, 3), dtype=np.uint8)).save(ex / "images" / "scene.png")
    (ex / "row.csv").write_text(CSV)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "inference_goal_force.py"), "--example_paths",
                        str(ex / "row.csv"), "--synthetic", "--layers", "2", "--num_inference_steps", "3",
                        "--output_dir", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    frames = sorted(os.listdir(out / "scene_seed0"))
    assert len(frames) == 81
    im = Image.open(out / "scene_seed0" / frames[40])
    assert im.size == (832, 480)
    a = np.asarray(im).astype(np.float32)
    assert np.isfinite(a).all() and a.std() > 1.0   # not a constant / NaN frame
