#!/bin/bash
# Round 4 -- everything that waits for a GPU, in the order it matters, ONE gpurun call (about 110 GPU-minutes; the blocks run in order of importance, a shorter --timeout cuts the tail):
#   /usr/local/graft/bin/gpurun --timeout 9000 -- 'bash tools/r4_gpu_job.sh'
# (round 4 itself never got a box: every call was refused, "GPU use for this repository has been closed from outside the build")
O=gpurun_out/r4; mkdir -p $O
# 1. HEAD's GPU suite, three times in a row (three consecutive greens with the commit hash)
for i in 1 2 3; do
  python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/suite_$i.txt; tail -1 $O/suite_$i.txt
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
# 2. dp_parity soak: which bound fires, and the distribution of every checked value (20 standalone runs)
bash tools/dp_soak.sh 20 $O/dp_soak > $O/dp_soak.txt 2>&1; tail -1 $O/dp_soak.txt
# 3. bench line + rocprof evidence of the same command
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
bash tools/profile_round.sh > $O/prof_round.log 2>&1
# 4. A/B of the four round-4 kernel changes (wgrad_rows, bn_act, bwd_stem, bwd_pw) against their round-3 forms (alone on the chip and inside the step)
bash tools/ab_r4_kernels.sh > $O/ab_r4_kernels.log 2>&1; cat gpurun_out/ab_r4_kernels/alone.txt gpurun_out/ab_r4_kernels/step.txt

# 5. attribution of the bf16 first-step loss shift (IEEE division / accurate expf builds)
bash tools/loss_shift_ab.sh > $O/loss_shift.log 2>&1; cat gpurun_out/loss_shift/summary.txt
# 6. the graph-destroy hunt (stand-alone HIP reproducer + the known reproducer under the heap checker)
bash tools/graph_hunt_r4.sh > $O/graph_hunt.log 2>&1; cat gpurun_out/graph_hunt_r4/summary.txt
