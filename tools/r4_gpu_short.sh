#!/bin/bash
# The 20-minute cut of tools/r4_gpu_job.sh (for a box that turns up late in a round): suite once, smoke, bench line, 5 dp_parity runs,
# one A/B round of the round-4 kernel changes inside the step.
O=gpurun_out/r4s; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/suite.txt; tail -1 $O/suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
bash tools/dp_soak.sh 5 $O/dp_soak > $O/dp_soak.txt 2>&1; tail -1 $O/dp_soak.txt
bash tools/ab_step.sh 1 "head|" "r3_all|Y5M_LIB=$PWD/build/exp/liby5m_r3_all.so" 2>&1 | tail -3 | tee $O/ab.txt
