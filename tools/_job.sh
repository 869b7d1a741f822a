#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/job16; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/gputest.txt 2>&1
tail -5 $O/gputest.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 600 $O/bench_line.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --no-detect > $O/prof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
bash tools/pmc_bench.sh > $O/pmc.txt 2>&1; cp gpurun_out/pmc_bench.json $O/pmc_bench.json; rm -rf gpurun_out/pmcb
Y5M_OVERLAP=0 timeout 900 python tools/layer_profile.py > $O/layers.txt 2>&1
tail -3 $O/layers.txt
