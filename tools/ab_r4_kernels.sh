#!/bin/bash
# A/B of the five kernel rewrites of round 4 -- made WITHOUT a GPU, from the static ISA audit (tools/isa_audit.py) and verified on the CPU
# executor only -- against their round-3 forms, alone on the chip and inside the replayed step. Both forms are compiled into liby5m.so
# (a template parameter per kernel); Y5M_R4_KERNELS is the bit mask of the round-4 forms in use, default 0 (csrc/y5m_common.h):
#   1  wgrad_rows_kernel  per-run address arithmetic on the scalar unit          (400 -> 253 / 399 -> 346 instructions per chunk)
#   2  bn_act_kernel      residual rows loaded raw: no vmcnt(0) per row            (428 -> 373, 120 -> 104 VGPRs)
#   4  bwd_stem_kernel    straight-line streaming loop: counted waits              (904 -> 703, 21 -> 0 full waits per chunk)
#   8  bwd_pw_kernel      unconditional re-requests: vmcnt(6) instead of vmcnt(0)  (725 -> 557 for C = 96)
#  16  bn_bwd_reduce      act as a template parameter, raw loads pinned together   (636 -> 404, 128 -> 109 VGPRs)
# Decision rule: a bit whose form is not faster INSIDE THE STEP stays 0; the winners become the default mask.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ab_r4_kernels; mkdir -p $O
# correctness first: the op-level cases of the five kernels with every round-4 form on
Y5M_R4_KERNELS=31 timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "wgrad_rows or bwd_pw or bwd_stem or bn_act" 2>&1 | tail -3 | tee $O/r4_forms_op_tests.txt
Y5M_R4_KERNELS=31 timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -k "per_layer or train_step_grads or graph_replay" 2>&1 | tail -3 | tee -a $O/r4_forms_op_tests.txt
for m in 0 1; do
  echo "== wgrad_rows alone: Y5M_R4_KERNELS=$m" | tee -a $O/alone.txt
  for a in "64 48 320 320 96 3 2" "64 48 160 160 48 3 1"; do Y5M_R4_KERNELS=$m python tools/conv_bench.py wgrad $a 30 2>/dev/null | tee -a $O/alone.txt; done
done
for m in 0 8; do
  echo "== bwd_pw alone: Y5M_R4_KERNELS=$m" | tee -a $O/alone.txt
  Y5M_R4_KERNELS=$m python tools/bwd_pw_bench.py 2>/dev/null | tail -8 | tee -a $O/alone.txt
done
bash tools/ab_step.sh ${1:-3} "r3_forms|Y5M_R4_KERNELS=0" "r4_all|Y5M_R4_KERNELS=31" "r4_rows|Y5M_R4_KERNELS=1" "r4_bnact|Y5M_R4_KERNELS=2" \
     "r4_stem|Y5M_R4_KERNELS=4" "r4_bwdpw|Y5M_R4_KERNELS=8" "r4_bnred|Y5M_R4_KERNELS=16" 2>&1 | tail -24 | tee $O/step.txt
