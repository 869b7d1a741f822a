#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/job28; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_conv.py -q -x -k "gemm8 or bn_accumulator" > $O/t.txt 2>&1
tail -5 $O/t.txt
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-detect --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for i in 1 2 3; do
Y5M_CONV_GEMM8=0 run gemm8=0
run gemm8=2-default
done
timeout 900 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -3
