#!/bin/bash
# A/B of wgrad_rows_kernel's scalar addressing (round 4, built without a GPU: tools/isa_audit.py counts 400 -> 253 instructions per chunk
# for <2,2,2> and 399 -> 346 for <1,4,1>, VALU 214 -> 38 / 213 -> 53) against the round-3 kernel, alone on the chip and inside the step.
# The round-3 form is rebuilt from git (commit 746228a) into build/exp/liby5m_rows_r3.so; run `tools/ab_wgrad_rows.sh build` on the CPU box
# first so that the library travels with the snapshot.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ab_wgrad_rows; mkdir -p $O build/exp/rows_r3_src
if [ ! -f build/exp/liby5m_rows_r3.so ]; then
  cp yolov5m_amd/csrc/*.hip yolov5m_amd/csrc/*.h yolov5m_amd/csrc/Makefile build/exp/rows_r3_src/
  git show 746228a:yolov5m_amd/csrc/y5m_conv_wgrad.hip > build/exp/rows_r3_src/y5m_conv_wgrad.hip
  sed -i 's#-I../../include#-I../../../include#; s#../../include/y5m.h#../../../include/y5m.h#' build/exp/rows_r3_src/Makefile
  sed -i 's#"../../include/y5m.h"#"../../../include/y5m.h"#' build/exp/rows_r3_src/y5m_common.h
  make -s -C build/exp/rows_r3_src -j8 OUT=../liby5m_rows_r3.so BUILD=../rows_r3_obj > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
fi
[ "$1" == "build" ] && exit 0
for lib in "" "Y5M_LIB=$PWD/build/exp/liby5m_rows_r3.so"; do
  echo "== ${lib:-HEAD}" | tee -a $O/alone.txt
  for a in "64 48 320 320 96 3 2" "64 48 160 160 48 3 1"; do env $lib python tools/conv_bench.py wgrad $a 30 2>/dev/null | tee -a $O/alone.txt; done
done
bash tools/ab_step.sh 3 "head|" "rows_r3|Y5M_LIB=$PWD/build/exp/liby5m_rows_r3.so" 2>&1 | tail -8 | tee $O/step.txt
