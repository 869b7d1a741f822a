#!/bin/bash
# PMC breakdown of the halo conv kernel (run on the GPU box): tools/pmc_halo.sh "<conv_bench args>" [Y5M_LIB]
cd /tmp && export TMPDIR=/tmp
ARGS="$1"; LIB="$2"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pmc
  Y5M_LIB=$LIB rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc -o h --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_bench.py $ARGS > /tmp/pmc.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "conv_halo" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print({c: round(v / cnt[c]) for c, v in agg.items()})
PY
done
