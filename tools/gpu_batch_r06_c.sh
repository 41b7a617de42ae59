#!/bin/bash
# round 6, call C: where the 8-rank launch loses 70 s while building one-layer models (CPU thread oversubscription?): phases with the
# default thread count and with OMP_NUM_THREADS=4
O=gpurun_out/r06
mkdir -p $O
nproc; python -c "import torch; print('torch threads', torch.get_num_threads())"
( time GF_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 8 --layers 1 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_n8_c_default.json.log 2> $O/bench_n8_c_default.stderr.log
grep "bench.py \[\|real" $O/bench_n8_c_default.stderr.log | cut -c1-200
( time OMP_NUM_THREADS=4 GF_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 8 --layers 1 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_n8_c_omp4.json.log 2> $O/bench_n8_c_omp4.stderr.log
grep "bench.py \[\|real" $O/bench_n8_c_omp4.stderr.log | cut -c1-200
python3 - <<'PY'
import json
a=json.loads([l for l in open('gpurun_out/r06/bench_n8_c_default.json.log') if l.startswith('{')][0])
b=json.loads([l for l in open('gpurun_out/r06/bench_n8_c_omp4.json.log') if l.startswith('{')][0])
print("same frames:", [d['frames_uint8_sha256'][:10] for d in a['self_check']['per_sample']]==[d['frames_uint8_sha256'][:10] for d in b['self_check']['per_sample']])
PY
