#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/job42; mkdir -p $O
timeout 2700 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp.py -q > $O/t.txt 2>&1
tail -4 $O/t.txt
