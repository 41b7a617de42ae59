#!/bin/bash
# round 5, call E: counters of every kernel bench.py prints a traffic figure for (VAE convs, self-attention, block GEMMs bf16 / e4m3)
O=gpurun_out/r05
mkdir -p $O
bash tools/profile_r05.sh > $O/pmc_all.log 2>&1
ls $O/pmc | wc -l
python3 tools/pmc_static.py $O/pmc profiles/r05/pmc > $O/pmc_static.log 2>&1; cat $O/pmc_static.log
cp profiles/pmc_static.json $O/pmc_static.json
