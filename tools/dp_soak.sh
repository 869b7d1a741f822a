#!/bin/bash
# Soak of tools/dp_parity.py (two ranks over gloo on this box's one GPU): N runs, every CHECK line kept.
# usage: tools/dp_soak.sh N outdir [extra env assignments...]
N=${1:-20}; OUT=${2:-gpurun_out/dp_soak}; shift 2
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 Y5M_DIST_BACKEND=gloo
for kv in "$@"; do export "$kv"; done
fails=0
for i in $(seq 1 "$N"); do
  port=$((20000 + RANDOM % 20000))
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
      tools/dp_parity.py > "$OUT/run_$i.log" 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails + 1)); echo "run $i rc $rc" >> "$OUT/summary.txt"; fi
  grep -h "^CHECK\|^WARN\|FAILED\|dp parity ok" "$OUT/run_$i.log" | sed "s/^/run $i: /" >> "$OUT/summary.txt"
done
echo "soak: $N runs, $fails failed ($*)" | tee -a "$OUT/summary.txt"
