"""Generates tests/golden/example_pendulum.npz in the BUILD container (the reference tree does not travel to the GPU box):

    python tests/golden/make_example_fixture.py

Data only: the first-frame image of the reference's pendulum example (datasets/examples/multi-object-collision/images/
_pendulum.png, opened with PIL and resized to 832x480 with LANCZOS as the dataset does it, DS:942-1026) as a
uint8 [480, 832, 3] array, and the numeric fields of its CSV row (_pendulum_obj1_prompt1.csv).  Used by `bench.py --inputs
example`: the image conditioning `y` and the force-map control latents of the timed run are then VAE encodings of these
structured inputs instead of seeded noise (profiles/r03/README.md, data-sensitivity table)."""
import csv
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference/datasets/examples/multi-object-collision"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_pendulum.npz")


def main():
    with open(os.path.join(REF, "_pendulum_obj1_prompt1.csv")) as f:
        row = next(csv.DictReader(f))
    img = Image.open(os.path.join(REF, "images", row["image"])).convert("RGB").resize((832, 480), resample=Image.Resampling.LANCZOS)   # DS:942-1026 / force_map.get_batch
    fields = ["projectile_force_angle", "projectile_force_magnitude", "projectile_coordx", "projectile_coordy", "projectile_mass",
              "target_indirect_force_angle", "target_indirect_force_magnitude", "target_coordx", "target_coordy", "target_mass",
              "width", "height"]
    np.savez_compressed(OUT, image=np.asarray(img, dtype=np.uint8), fields=np.array(fields),
                        values=np.array([float(row[k]) for k in fields], dtype=np.float64), caption=np.array(row["caption"]))
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
