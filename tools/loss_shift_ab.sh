#!/bin/bash
# Attribution of the bf16 first-step loss shift (1.51e-5 -> 6.94e-5 relative to the reference after the
# hardware reciprocal and __expf went into every SiLU): the first-step loss at B=64 @ 640^2 with the default build, with the
# IEEE division, with the accurate expf, and with both. Builds are made here if missing (hipcc cross-compiles on the CPU box:
# run this once THERE so that build/exp/*.so travel with the snapshot). Reference f32 loss: tests/golden g13 (303.6076).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/loss_shift; mkdir -p $O build/exp
for v in "ieee|-DY5M_IEEE_RCP" "exp|-DY5M_ACCURATE_EXP" "both|-DY5M_IEEE_RCP -DY5M_ACCURATE_EXP"; do
  n=${v%%|*}; f=${v#*|}
  [ -f build/exp/liby5m_$n.so ] || make -s -C yolov5m_amd/csrc -j8 OUT=../../build/exp/liby5m_$n.so BUILD=../../build/exp/$n EXTRA="$f" > $O/build_$n.log 2>&1
done
[ "$1" == "build" ] && exit 0
for n in default ieee exp both; do
  lib=""; [ $n != default ] && lib="Y5M_LIB=$PWD/build/exp/liby5m_$n.so"
  for dt in bf16 f32; do
    echo "$n $dt $(env $lib DT=$dt GRAPH=0 timeout 600 python tools/loss_trace.py 1 64 640 2>/dev/null | tail -1)" | tee -a $O/summary.txt
  done
done
