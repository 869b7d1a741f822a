#!/bin/bash
# Round 5, first GPU call: HEAD's GPU suite once (no -x, every failure listed), smoke, the bench line, the round's rocprof evidence set.
O=gpurun_out/r5a; mkdir -p $O
python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > $O/suite_1.txt; tail -1 $O/suite_1.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo; tail -3 $O/bench.err
bash tools/profile_round.sh > $O/prof_round.log 2>&1; tail -12 $O/prof_round.log
