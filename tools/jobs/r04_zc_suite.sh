#!/bin/bash
# round 4, call ZC: the whole GPU suite on the final tree with the 16-thread cap of tests/conftest.py (durations), smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zc; mkdir -p $O; export TMPDIR=/tmp
uptime > $O/host.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
cat $O/rc.log $O/host.txt; tail -n 22 $O/pytest_gpu.log; tail -n 3 $O/smoke.log
