#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/job21; mkdir -p $O
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-detect --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('final_loss'))" | tee -a $O/ab.txt; }
for i in 1 2 3; do run base; Y5M_WGRAD_DEFER_HALO=1 run defer_halo; done
