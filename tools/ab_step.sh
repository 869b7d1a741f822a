#!/bin/bash
# A/B of environment configurations INSIDE the graph-replayed train step (B=64 @ 640^2, bf16), alternating rounds in fresh
# processes (knobs are read once per process). usage: [AB_TAG=tag] tools/ab_step.sh <rounds> "<name>|<ENV=.. ENV=..>" ...
# lines go to gpurun_out/ab_step/ab[_tag].txt; tools/ab_summary.py turns such a file into means, spread and a verdict per variant
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
O=$PWD/gpurun_out/ab_step; mkdir -p $O
R=$1; shift
for r in $(seq 1 $R); do
  for c in "$@"; do
    name=${c%%|*}; envs=${c#*|}
    out=$(env $envs timeout 600 python tools/step_ablation.py none 2>&1 | grep "ms/step" | awk '{print $(NF-1)}')
    echo "round $r  $name  [$envs]  $out ms/step" | tee -a $O/ab${AB_TAG:+_$AB_TAG}.txt
  done
done
