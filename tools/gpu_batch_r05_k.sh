#!/bin/bash
# round 5, call K: the N = 2 path at FULL size on the one GPU of the box (two ranks share cuda:0, gloo transport: same code path as RCCL, another
# transport): pre-flight at production size, CFG pair, VAE tiles split over the pair, frame all-gather; latents / frames must equal the N = 1 run's
O=gpurun_out/r05
mkdir -p $O
( time GF_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_n2_gloo_fullsize.json.log 2> $O/bench_n2_gloo_fullsize.stderr.log
echo "rc=$?" >> $O/bench_n2_gloo_fullsize.stderr.log
grep "pre-flight\|rc=\|real" $O/bench_n2_gloo_fullsize.stderr.log | cut -c1-400
( timeout 600 python bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --config5-steps 0 --peaky-steps 0 --no-preloop ) > $O/bench_n1_4steps.json.log 2> /dev/null
python3 - <<'PY'
import json
a=json.loads([l for l in open('gpurun_out/r05/bench_n2_gloo_fullsize.json.log') if l.startswith('{')][0])
b=json.loads([l for l in open('gpurun_out/r05/bench_n1_4steps.json.log') if l.startswith('{')][0])
print("N=2 latents == N=1:", a["self_check"]["latents"]["sha256"]==b["self_check"]["latents"]["sha256"], "frames:", a["self_check"]["frames_uint8"]["sha256"]==b["self_check"]["frames_uint8"]["sha256"])
print("preflight", {k:(v["bytes"], round(v["seconds"]*1e3,2)) for k,v in a["preflight"]["steps"].items()}, a["preflight"]["min_hbm_free_gb"])
PY
