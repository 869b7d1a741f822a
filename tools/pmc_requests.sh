#!/bin/bash
# L2 request granularity per kernel of the train step: bytes per TCC read / write request
# (HBM-side bytes from FETCH_SIZE / WRITE_SIZE, requests from TCC_READ_sum / TCC_WRITE_sum; separate --pmc passes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcr; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-detect --no-roofline --no-graph"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_READ_sum TCC_WRITE_sum"; do
  d=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$d -o p -- $CMD > $OUT/$d.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmcr/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for k, d in agg.items():
    rb, wb = d.get("FETCH_SIZE", 0) * 2048, d.get("WRITE_SIZE", 0) * 1024
    rq, wq = d.get("TCC_READ_sum", 0), d.get("TCC_WRITE_sum", 0)
    rows.append((rb + wb, k, rb, wb, rq, wq))
print(f"{'kernel':70s} {'HBM rd MB':>10s} {'wr MB':>9s} {'B/rd req':>9s} {'B/wr req':>9s}")
for tot, k, rb, wb, rq, wq in sorted(rows, reverse=True)[:28]:
    print(f"{k[:70]:70s} {rb/1e6:10.1f} {wb/1e6:9.1f} {rb/max(rq,1):9.1f} {wb/max(wq,1):9.1f}")
PY
