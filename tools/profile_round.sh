#!/bin/bash
# The evidence set of a round (run on the GPU box; copy gpurun_out/prof_round/* to profiles/rNN_*):
#   bench_line.json          the bench JSON line of a plain run
#   kernel_stats.csv         rocprofv3 --kernel-trace --stats of the graph-replayed bench (kernel durations overlap across the two streams)
#   pmc_bench.json           HBM traffic per kernel (tools/pmc_bench.sh)
#   pmc_conv_step.txt        MFMA-busy / issue / stall / LDS / L2 counters of the conv and weight-gradient kernels inside the step (tools/pmc_step.sh)
#   layers.txt               per-layer conv launch durations (tools/layer_profile.py)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_round; rm -rf $O; mkdir -p $O
R=$PWD
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.out 2> $O/bench.err; grep '^{' $O/bench.out | tail -1 > $O/bench_line.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --no-detect > $O/prof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench.json $O/pmc_bench.json
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; cp gpurun_out/pmc_step/pmc_conv_step.txt $O/pmc_conv_step.txt
rm -rf gpurun_out/pmc_step/g* gpurun_out/pmcb
Y5M_OVERLAP=0 timeout 600 python tools/layer_profile.py 64 640 > $O/layers.txt 2>&1
ls -la $O
