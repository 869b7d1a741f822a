#!/bin/bash
# Everything that waits for a GPU, as numbered blocks (rounds 4-6 never got a box: every gpurun call was refused -- "GPU use for this
# repository has been closed from outside the build"). Run the blocks that fit the lease, most important first:
#   /usr/local/graft/bin/gpurun --timeout 4200 -- 'bash tools/gpu_job.sh 1 2'        # suite + smoke + bench, rocprof evidence set (~50 min)
#   /usr/local/graft/bin/gpurun --timeout 4800 -- 'bash tools/gpu_job.sh 3 4'        # suite x2, every staged A/B (~55 min)
#   /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash tools/gpu_job.sh 5 6 7 8'    # dp soak, graph-destroy hunt, loss-shift A/B, 2-rank line (~50 min)
# Results under gpurun_out/r6/. What to do with them: NOTES.md, "When a GPU is reachable again".
O=gpurun_out/r6; mkdir -p $O
cat .git_head > $O/head.txt 2>/dev/null
b1() {  # HEAD's GPU suite once WITHOUT -x (every failure listed), smoke, the bench line with all its legs, the YOLO_LOSS line
  timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > $O/suite_1.txt; tail -1 $O/suite_1.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
  timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo; tail -3 $O/bench.err
  timeout 600 python bench.py --loss yolo --steps 20 --warmup 5 --no-roofline --no-detect --no-cpu-baseline > $O/bench_yolo_loss.json 2> $O/bench_yolo_loss.err; head -c 400 $O/bench_yolo_loss.json; echo
}
b2() {  # the round's rocprof evidence set (kernel stats, HBM PMC, conv PMC inside the step, per-layer table) -> copy to profiles/r06_*
  bash tools/profile_round.sh > $O/prof_round.log 2>&1; tail -12 $O/prof_round.log
  timeout 600 python tools/eval_layers.py 32 640 > $O/eval_layers_b32_640.txt 2>&1; timeout 900 python tools/eval_layers.py 128 1280 > $O/eval_layers_b128_1280.txt 2>&1
}
b3() {  # suite twice more (three consecutive greens with the commit hash)
  for i in 2 3; do timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/suite_$i.txt; tail -1 $O/suite_$i.txt; done
}
b4() {  # A/B inside the step, one bit at a time: the five round-4 kernel forms (Y5M_R4_KERNELS) -- op tests with all five on first.
        # Decision rule: what is not faster INSIDE THE STEP is deleted with its `if constexpr` branch.
  AB_TAG=r4 bash tools/ab_r4_kernels.sh 3 > $O/ab_r4_kernels.log 2>&1; cat gpurun_out/ab_r4_kernels/r4_forms_op_tests.txt gpurun_out/ab_r4_kernels/step.txt
  # + the round-5 LDS-tiled SPPF pooling (Y5M_POOL_TILE=1: forward 2 launches -> 1, backward 6 -> 1; bit-identical results)
  Y5M_POOL_TILE=0 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "sppf_pool" 2>&1 | tail -2 | tee $O/pool_tile_test.txt
  (cd /tmp && export TMPDIR=/tmp && for m in 0 1; do Y5M_POOL_TILE=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/pool_prof$m -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-detect > /dev/null 2>&1; done)
  for m in 0 1; do f=$(find $O/pool_prof$m -name "*kernel_stats.csv" | head -1); echo "== Y5M_POOL_TILE=$m"; grep -i "sppf\|maxpool" "$f" | cut -c1-160; done | tee $O/pool_tile_kernels.txt; rm -rf $O/pool_prof0 $O/pool_prof1
  timeout 600 python -m pytest tests/test_gpu_detect_loss.py -m gpu -q -k "sparse_head" 2>&1 | tail -2 | tee $O/head_pack16_test.txt
  AB_TAG=pool bash tools/ab_step.sh 3 "default|" "pool_tile|Y5M_POOL_TILE=1" "head_pack16|Y5M_HEAD_PACK16=1" "both|Y5M_POOL_TILE=1 Y5M_HEAD_PACK16=1" 2>&1 | tail -12 | tee $O/ab_pool_tile.txt
  python tools/ab_summary.py gpurun_out/ab_step/ab_r4.txt gpurun_out/ab_step/ab_pool.txt | tee $O/ab_summary.txt
  # + the halo kernel's two-stage weight ring for images 45..88 pixels wide (Y5M_CONV_HALO_NS2=1; configs[4]'s 80x80 stage)
  timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -k "halo_two_stage or halo_wide" 2>&1 | tail -2 | tee $O/halo_ns2_test.txt
  for m in 0 1; do echo "== 192 -> 192 3x3 @ 80x80, B = 128, alone: Y5M_CONV_HALO_NS2=$m" | tee -a $O/halo_ns2_alone.txt
    Y5M_CONV_HALO_NS2=$m python tools/conv_bench.py fwd 128 192 80 80 192 3 1 30 2>/dev/null | tee -a $O/halo_ns2_alone.txt
    Y5M_CONV_HALO_NS2=$m python tools/conv_bench.py dgrad 128 192 80 80 192 3 1 30 2>/dev/null | tee -a $O/halo_ns2_alone.txt; done
  for m in 0 1; do Y5M_CONV_HALO_NS2=$m timeout 900 python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('Y5M_CONV_HALO_NS2=$m forward_1280', d['detect'].get('forward_1280'))" | tee -a $O/halo_ns2_detect.txt; done
  # + pack-once (opt-in since round 6): the eval forward leg with and without it
  for m in 0 1; do Y5M_PACK_ONCE=$m timeout 600 python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('Y5M_PACK_ONCE=$m forward', d.get('forward'))" | tee -a $O/pack_once.txt; done
}
b5() { bash tools/dp_soak.sh 20 $O/dp_soak > $O/dp_soak.txt 2>&1; tail -1 $O/dp_soak.txt; }     # dp_parity soak: which bound fires, distributions
b6() { bash tools/graph_hunt_r4.sh > $O/graph_hunt.log 2>&1; cat gpurun_out/graph_hunt_r4/summary.txt; }     # stand-alone graph-destroy reproducer + heap checker
b7() { bash tools/loss_shift_ab.sh > $O/loss_shift.log 2>&1; cat gpurun_out/loss_shift/summary.txt; }        # bf16 first-step loss shift: IEEE division / accurate expf builds
b8() {  # two-rank bench line on this box's one GPU (gloo; the RCCL path itself needs two devices: the driver's SCALE run)
  Y5M_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err; head -c 300 $O/bench_2ranks_gloo.json; echo
}
for b in "${@:-1 2}"; do echo "=== block $b"; b$b; done
