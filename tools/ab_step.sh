#!/bin/bash
# A/B of environment configurations INSIDE the graph-replayed train step (B=64 @ 640^2, bf16), alternating rounds in fresh
# processes (knobs are read once per process). usage: tools/ab_step.sh <rounds> "<name>|<ENV=.. ENV=..>" ...
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=$PWD/gpurun_out/ab_step; mkdir -p $O
R=$1; shift
for r in $(seq 1 $R); do
  for c in "$@"; do
    name=${c%%|*}; envs=${c#*|}
    out=$(env $envs timeout 600 python tools/step_ablation.py none 2>&1 | grep "ms/step" | awk '{print $(NF-1)}')
    echo "round $r  $name  [$envs]  $out ms/step" | tee -a $O/ab.txt
  done
done
