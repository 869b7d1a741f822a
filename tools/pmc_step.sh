#!/bin/bash
# PMC counters of the conv / weight-gradient kernels INSIDE the train step (B=64 @ 640^2, bf16; eager launches, the step's own
# two-stream schedule), one counter group per pass, no trace domains besides --kernel-trace (gpurun refuses other mixes).
# Output: gpurun_out/pmc_step/pmc_conv_step.txt  (copy to profiles/rNN_pmc_conv_step.txt)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_step; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-detect --no-roofline --no-graph"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- $CMD > $OUT/g$i.log 2>&1
done
python - <<'PY' | tee gpurun_out/pmc_step/pmc_conv_step.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmc_step/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
print("# rocprofv3 --pmc <one group per pass> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-graph ... (tools/pmc_step.sh)")
print("# B=64 @ 640x640 bf16 train step, eager launches on the step's own two streams; counters SUMMED over every launch of the")
print("# kernel in the run (3 steps); FETCH_SIZE / WRITE_SIZE in KiB as reported (gfx950: x2 on FETCH_SIZE for wide reads).")
print("# derived (ratios of sums, launch counts cancel). GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the launches = GUI / 8.")
print("#   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI / 8); lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x GUI / 8)")
print("#   issue / stall / parked = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES; lds_stall = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES")
print("#   l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); HBM MB/launch = (FETCH_SIZE x 2048 + WRITE_SIZE x 1024) / launches; L2 req MB/launch = TCC_REQ x 128 / launches")
keep = ("conv_", "wgrad", "bwd_pw", "bn_bwd", "bn_act")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if not any(s in k for s in keep):
        continue
    n = max(calls[k].values())
    wc = max(d.get("SQ_WAVE_CYCLES", 0), 1.0)
    cyc = max(d.get("GRBM_GUI_ACTIVE", 0), 1.0) / 8
    print(f"\n{k}   launches {n}")
    print(f"   mfma_busy {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc):.3f}  lds_busy {d.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * cyc):.3f}  issue {d.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}  "
          f"stall {d.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  parked {d.get('SQ_WAIT_ANY', 0) / wc:.3f}  "
          f"lds_stall {d.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}  "
          f"l2_hit {d.get('TCC_HIT_sum', 0) / max(d.get('TCC_HIT_sum', 0) + d.get('TCC_MISS_sum', 0), 1):.3f}  "
          f"HBM MB/launch {(d.get('FETCH_SIZE', 0) * 2048 + d.get('WRITE_SIZE', 0) * 1024) / n / 1e6:.1f}  "
          f"L2 req MB/launch {d.get('TCC_REQ_sum', 0) * 128 / n / 1e6:.1f}")
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:.4g}")
PY
