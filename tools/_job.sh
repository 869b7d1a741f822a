#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/job30; mkdir -p $O
timeout 2700 python -m pytest tests/ -q -m gpu > $O/gputest.txt 2>&1
tail -4 $O/gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 300 $O/bench_line.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --no-detect > $O/prof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
Y5M_OVERLAP=0 timeout 900 python tools/layer_profile.py > $O/layers.txt 2>&1
tail -1 $O/layers.txt
