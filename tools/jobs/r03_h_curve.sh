#!/bin/bash
# round 3, call H: the 20-step curve test with the oracle loop on 8 torch threads (duration), smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03h; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -x -q --durations=3 tests/test_gpu_bench_config.py -k loss_curve > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 8 $O/pytest.log; tail -n 8 $O/smoke.log
