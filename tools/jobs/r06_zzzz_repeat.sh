#!/bin/bash
# round 6, last call: the whole GPU suite three times on the final tree (flake hunt: the suite's atomics-order-dependent comparisons), smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzzz; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_$r.log 2>&1; echo "pytest_gpu run $r rc=$?" >> $O/rc.log; tail -n 4 $O/pytest_gpu_$r.log >> $O/rc.log; done
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
cat $O/rc.log; grep -h "^FAILED\|^E  " $O/pytest_gpu_*.log | head -20
