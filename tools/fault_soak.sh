#!/bin/bash
# Soak of the graph-replay fix (NOTES.md): N consecutive runs of tests/test_gpu_model.py (the multi_scale all-sizes scenario
# in-process behind ~45 other tests that create and drop graphs) and of tools/graph_first_replay.py (5 rounds x 15 captured plans).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
N=${1:-10}; OUT=gpurun_out/soak; mkdir -p $OUT
# every box reports the hostname "runc": tell them apart by the GPU's unique id
BOX=$(rocm-smi --showuniqueid 2>/dev/null | grep -o "0x[0-9a-f]*" | head -n 1); BOX=${BOX:-$(hostname)}
ok1=0; ok2=0
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -p no:cacheprovider > $OUT/model_$i.log 2>&1 && ok1=$((ok1+1))
  timeout 300 python tools/graph_first_replay.py 5 f32 4 > $OUT/replay_$i.log 2>&1 && grep -q "'first_replay': 0, 'second_replay': 0" $OUT/replay_$i.log && ok2=$((ok2+1))
  echo "run $i: model file $(tail -n 1 $OUT/model_$i.log | cut -c1-80) | replay $(tail -n 1 $OUT/replay_$i.log | cut -c1-100)"
done
echo "soak on $(hostname) gpu $BOX ($(date -u +%FT%TZ)): tests/test_gpu_model.py green $ok1 / $N, graph_first_replay clean $ok2 / $N" | tee $OUT/summary_$BOX.txt
