#!/bin/bash
# round 6, call I: ONE video end to end through the launcher at production size — CSV row -> force map -> 2 tiled VAE encodes -> 50-step loop
# (40 + 10 blocks, both experts, CFG) -> tiled VAE decode -> 81 PNG frames — wall clock of the whole process, random-init weights
O=gpurun_out/r06
mkdir -p $O /tmp/e2e/example/images
python3 - <<'PY'
import numpy as np
from PIL import Image
rng = np.random.default_rng(0)
Image.fromarray(rng.integers(0, 255, (480, 832, 3), dtype=np.uint8)).save('/tmp/e2e/example/images/scene.png')
open('/tmp/e2e/example/row.csv','w').write("image,projectile_force_angle,projectile_force_magnitude,projectile_coordx,projectile_coordy,projectile_mass,target_indirect_force_angle,target_indirect_force_magnitude,target_coordx,target_coordy,target_mass,width,height,caption\nscene.png,-1.0,-1.0,368,108,-1,0.0,350.0,545,114,2.0,832,480,\"The pendulum swings, striking and toppling the red block.\"\n")
PY
( time python scripts/inference_goal_force.py --device_id 0 --world_size 1 --seed 0 --control_signal_type goal_force --example_paths /tmp/e2e/example/row.csv --synthetic --num_inference_steps 50 --output_dir /tmp/e2e/out ) > $O/e2e_one_video_50steps.log 2>&1
echo "rc=$?" >> $O/e2e_one_video_50steps.log
tail -8 $O/e2e_one_video_50steps.log; ls /tmp/e2e/out | head -3; ls /tmp/e2e/out/*/ | wc -l
