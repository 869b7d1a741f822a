#!/bin/bash
# A/B of the BatchNorm statistics path inside the full train step (B=64 @ 640^2, bf16), on the GPU box:
#   Y5M_BN_FUSE=0  partial rows + bn_reduce_finalize_kernel launches (three launches per CBL and direction)
#   Y5M_BN_FUSE=1  f64 accumulator rows + fused consumers (csrc/y5m_bnfuse.h; 2 / 3 = forward / backward only)
# then a rocprofv3 kernel-stats summary of both (written under gpurun_out/ab_bn_fuse/).
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=$PWD/gpurun_out/ab_bn_fuse; mkdir -p $O
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-detect --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for i in 1 2; do for f in 0 1 2 3; do Y5M_BN_FUSE=$f run "Y5M_BN_FUSE=$f"; done; done
R=$PWD
for f in 0 1; do
  (cd /tmp && Y5M_BN_FUSE=$f timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$f -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-detect > $O/prof$f.log 2>&1)
  find $O/prof$f -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_fuse$f.csv
  rm -rf $O/prof$f
done
