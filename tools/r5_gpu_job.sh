#!/bin/bash
# Round 5 -- everything that waits for a GPU, ONE gpurun call, in the order it matters (a short lease still yields GPUTEST + BENCH):
#   /usr/local/graft/bin/gpurun --timeout 9000 -- 'bash tools/r5_gpu_job.sh'
# (rounds 4 and 5 never got a box: every call was refused -- "GPU use for this repository has been closed from outside the build")
# Blocks 1-3 ~25 min, 4 ~25 min, 5-6 ~30 min, 7-8 ~20 min. A shorter --timeout cuts the tail.
O=gpurun_out/r5; mkdir -p $O
# 1. HEAD's GPU suite once WITHOUT -x (every failure listed), smoke, the bench line with all its legs
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > $O/suite_1.txt; tail -1 $O/suite_1.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo; tail -3 $O/bench.err
# 2. the round's rocprof evidence set (kernel stats, HBM PMC, conv PMC inside the step, per-layer table) -> copy to profiles/r05_*
bash tools/profile_round.sh > $O/prof_round.log 2>&1; tail -12 $O/prof_round.log
# 3. suite twice more (three consecutive greens with the commit hash)
for i in 2 3; do timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/suite_$i.txt; tail -1 $O/suite_$i.txt; done
# 4. A/B inside the step, one bit at a time: the five round-4 kernel forms (Y5M_R4_KERNELS) -- op tests with all five on first.
#    Decision rule: what is not faster INSIDE THE STEP stays off.
AB_TAG=r4 bash tools/ab_r4_kernels.sh 3 > $O/ab_r4_kernels.log 2>&1; cat gpurun_out/ab_r4_kernels/r4_forms_op_tests.txt gpurun_out/ab_r4_kernels/step.txt
#    + the round-5 LDS-tiled SPPF pooling (Y5M_POOL_TILE=1: forward 2 launches -> 1, backward 6 -> 1; bit-identical results):
#      its GPU test, the pooling kernels' times under rocprofv3, and the step A/B
Y5M_POOL_TILE=0 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "sppf_pool" 2>&1 | tail -2 | tee $O/pool_tile_test.txt
(cd /tmp && export TMPDIR=/tmp && for m in 0 1; do Y5M_POOL_TILE=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/pool_prof$m -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-detect > /dev/null 2>&1; done)
for m in 0 1; do f=$(find $O/pool_prof$m -name "*kernel_stats.csv" | head -1); echo "== Y5M_POOL_TILE=$m"; grep -i "sppf\|maxpool" "$f" | cut -c1-160; done | tee $O/pool_tile_kernels.txt; rm -rf $O/pool_prof0 $O/pool_prof1
timeout 600 python -m pytest tests/test_gpu_detect_loss.py -m gpu -q -k "sparse_head" 2>&1 | tail -2 | tee $O/head_pack16_test.txt
AB_TAG=pool bash tools/ab_step.sh 3 "default|" "pool_tile|Y5M_POOL_TILE=1" "head_pack16|Y5M_HEAD_PACK16=1" "both|Y5M_POOL_TILE=1 Y5M_HEAD_PACK16=1" 2>&1 | tail -12 | tee $O/ab_pool_tile.txt
python tools/ab_summary.py gpurun_out/ab_step/ab_r4.txt gpurun_out/ab_step/ab_pool.txt | tee $O/ab_summary.txt
#    + the halo kernel's two-stage weight ring for images 45..88 pixels wide (Y5M_CONV_HALO_NS2=1; BASELINE configs[4]: the ten
#      192 -> 192 3x3 layers of the 80x80 stage at 1280x1280 = 21 % of the forward's FLOPs, today on the tiled kernel): its GPU test,
#      the layer alone, and the detect leg's forward_1280 with the knob off / on
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "halo_two_stage or halo_wide" 2>&1 | tail -2 | tee $O/halo_ns2_test.txt
for m in 0 1; do echo "== 192 -> 192 3x3 @ 80x80, B = 128, alone: Y5M_CONV_HALO_NS2=$m" | tee -a $O/halo_ns2_alone.txt
  Y5M_CONV_HALO_NS2=$m python tools/conv_bench.py fwd 128 192 80 80 192 3 1 30 2>/dev/null | tee -a $O/halo_ns2_alone.txt
  Y5M_CONV_HALO_NS2=$m python tools/conv_bench.py dgrad 128 192 80 80 192 3 1 30 2>/dev/null | tee -a $O/halo_ns2_alone.txt; done
for m in 0 1; do Y5M_CONV_HALO_NS2=$m timeout 900 python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('Y5M_CONV_HALO_NS2=$m forward_1280', d['detect'].get('forward_1280'))" | tee -a $O/halo_ns2_detect.txt; done
# 5. dp_parity soak: which bound fires, and the distribution of every checked value (20 standalone runs)
bash tools/dp_soak.sh 20 $O/dp_soak > $O/dp_soak.txt 2>&1; tail -1 $O/dp_soak.txt
# 6. the graph-destroy hunt (stand-alone HIP reproducer + the known reproducer under the heap checker)
bash tools/graph_hunt_r4.sh > $O/graph_hunt.log 2>&1; cat gpurun_out/graph_hunt_r4/summary.txt
# 7. attribution of the bf16 first-step loss shift (IEEE division / accurate expf builds)
bash tools/loss_shift_ab.sh > $O/loss_shift.log 2>&1; cat gpurun_out/loss_shift/summary.txt
# 8. two-rank bench line on this box's one GPU (gloo; the RCCL path itself needs two devices: the driver's SCALE run)
Y5M_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err; head -c 300 $O/bench_2ranks_gloo.json; echo
