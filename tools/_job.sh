#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/job18; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/gputest.txt 2>&1
tail -8 $O/gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
