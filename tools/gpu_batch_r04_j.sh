#!/bin/bash
# Round 4, batch J (GPU box, repo root): the uncond forward of a CFG step on a second HIP stream (GF_CFG_STREAMS=1) against one stream,
# same box, with and without the per-launch events; bit-identity by the latents' sha256 in the bench line.
O=gpurun_out/r04
mkdir -p $O
A="--steps 4 --warmup 1 --no-cpu-baseline --config5-steps 0"
python bench.py $A --no-launch-events > $O/bench_streams1_noev.json.log 2>&1
GF_CFG_STREAMS=1 python bench.py $A --no-launch-events > $O/bench_streams2_noev.json.log 2>&1
python bench.py $A > $O/bench_streams1.json.log 2>&1
GF_CFG_STREAMS=1 python bench.py $A > $O/bench_streams2.json.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04/bench_streams*.json.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d['value'], d.get('roofline', {}).get('avg_launch_ms'), str(d.get('self_check', {}).get('latents', {}).get('sha256'))[:16])
    except Exception as e:
        print(f, 'FAILED', e, open(f).read()[-600:])
PY
