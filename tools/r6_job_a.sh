#!/bin/bash
# Round 6, GPU call A: HEAD's GPU suite WITHOUT -x, smoke, the bench line, then the round's rocprof evidence set.
O=gpurun_out/r6a; mkdir -p $O
git_head=$(cat .git_head 2>/dev/null); echo "HEAD $git_head" > $O/head.txt
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > $O/suite_1.txt; tail -1 $O/suite_1.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo; tail -3 $O/bench.err
bash tools/profile_round.sh > $O/prof_round.log 2>&1; tail -12 $O/prof_round.log
