#!/bin/bash
# round 5, call H: loop-level parity at production size on the round's final build (4-step schedule: 2 high-noise + 2 low-noise steps, bf16 and fp8,
# PSNR of the decoded frames through the new VAE kernels), then the counters again (gf_conv_direct.hip was re-organised: new source hash)
O=gpurun_out/r05
mkdir -p $O
( time timeout 1500 python tests/fullsize_parity.py --steps 4 --fp8 --out $O/fullsize_parity_4step.json ) > $O/fullsize_parity_4step.log 2>&1
grep -v "^MIOpen\|amdgpu.ids" $O/fullsize_parity_4step.log | grep "vs fp32\|PSNR\|real" | cut -c1-330
bash tools/profile_r05.sh > $O/pmc_all.log 2>&1
python3 tools/pmc_static.py $O/pmc profiles/r05/pmc > $O/pmc_static.log 2>&1; cat $O/pmc_static.log
cp profiles/pmc_static.json $O/pmc_static.json
