#!/bin/bash
# what each kernel family costs INSIDE the overlapped, graph-replayed step (B = 64 @ 640^2): the family's C-ABI entry points
# are stubbed out one group at a time (tools/step_ablation.py; results are wrong by construction, only the time is read)
cd "$(dirname "$0")/.."
for s in none y5m_wgrad,y5m_unpack_wgrad y5m_bn_bwd y5m_bn_act y5m_bn_finalize \
         y5m_bn_bwd,y5m_bn_act,y5m_bn_finalize y5m_conv,y5m_conv_multi \
         y5m_wgrad,y5m_unpack_wgrad,y5m_bn_bwd,y5m_bn_act,y5m_bn_finalize; do
  timeout 300 python tools/step_ablation.py $s 2>&1 | tail -1
done
