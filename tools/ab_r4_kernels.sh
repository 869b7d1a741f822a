#!/bin/bash
# A/B of the five kernel changes of round 4 -- made WITHOUT a GPU, from the static ISA audit (tools/isa_audit.py) and verified on the CPU
# executor only -- against their round-3 forms, alone on the chip and inside the replayed step:
#   wgrad_rows_kernel  per-run address arithmetic on the scalar unit          (400 -> 253 / 399 -> 346 instructions per chunk)
#   bn_act_kernel      residual rows loaded raw: no vmcnt(0) per row            (428 -> 373, 120 -> 104 VGPRs)
#   bwd_stem_kernel    straight-line streaming loop: counted waits              (904 -> 703, 21 -> 0 full waits per chunk)
#   bwd_pw_kernel      unconditional re-requests: vmcnt(6) instead of vmcnt(0)  (725 -> 557 for C = 96)
#   bn_bwd_reduce      act as a template parameter, raw loads pinned together   (636 -> 404, 128 -> 109 VGPRs; in the "bnact" = y5m_nn.hip variant)
# One experiment library per kernel (that file taken from commit 746228a = end of round 3, everything else HEAD) + one with all four.
# `tools/ab_r4_kernels.sh build` on the CPU box builds them into build/exp/ so that they travel with the snapshot.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ab_r4_kernels; mkdir -p $O
build_variant() {   # name, files taken from round 3...
  local name=$1; shift
  [ -f build/exp/liby5m_r3_$name.so ] && return
  local d=build/exp/r3_${name}_src; rm -rf $d; mkdir -p $d
  cp yolov5m_amd/csrc/*.hip yolov5m_amd/csrc/*.h yolov5m_amd/csrc/Makefile $d/
  for f in "$@"; do git show 746228a:yolov5m_amd/csrc/$f > $d/$f; done
  sed -i 's#-I../../include#-I../../../include#; s#../../include/y5m.h#../../../include/y5m.h#' $d/Makefile
  sed -i 's#"../../include/y5m.h"#"../../../include/y5m.h"#' $d/y5m_common.h
  make -s -C $d -j8 OUT=../liby5m_r3_$name.so BUILD=../r3_${name}_obj > $O/build_$name.log 2>&1 || { echo "build of $name failed"; tail -5 $O/build_$name.log; }
  rm -rf $d build/exp/r3_${name}_obj                 # (only the library travels)
}
build_variant rows y5m_conv_wgrad.hip
build_variant bnact y5m_nn.hip
build_variant stem y5m_bwd_stem.hip
build_variant bwdpw y5m_bwd_pw.hip
build_variant all y5m_conv_wgrad.hip y5m_nn.hip y5m_bwd_stem.hip y5m_bwd_pw.hip
[ "$1" == "build" ] && exit 0
for lib in "" "Y5M_LIB=$PWD/build/exp/liby5m_r3_rows.so"; do
  echo "== wgrad_rows alone: ${lib:-HEAD}" | tee -a $O/alone.txt
  for a in "64 48 320 320 96 3 2" "64 48 160 160 48 3 1"; do env $lib python tools/conv_bench.py wgrad $a 30 2>/dev/null | tee -a $O/alone.txt; done
done
for lib in "" "Y5M_LIB=$PWD/build/exp/liby5m_r3_bwdpw.so"; do
  echo "== bwd_pw alone: ${lib:-HEAD}" | tee -a $O/alone.txt
  env $lib python tools/bwd_pw_bench.py 2>/dev/null | tail -8 | tee -a $O/alone.txt
done
bash tools/ab_step.sh 3 "head|" "r3_all|Y5M_LIB=$PWD/build/exp/liby5m_r3_all.so" "r3_rows|Y5M_LIB=$PWD/build/exp/liby5m_r3_rows.so" \
     "r3_bnact|Y5M_LIB=$PWD/build/exp/liby5m_r3_bnact.so" "r3_stem|Y5M_LIB=$PWD/build/exp/liby5m_r3_stem.so" \
     "r3_bwdpw|Y5M_LIB=$PWD/build/exp/liby5m_r3_bwdpw.so" 2>&1 | tail -20 | tee $O/step.txt
