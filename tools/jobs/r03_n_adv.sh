#!/bin/bash
# round 3, call N: the 36-head adversarial parity tests with their bounds at the measured values
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03n; mkdir -p $O; export TMPDIR=/tmp
timeout 420 python -m pytest -m gpu -x -q tests/test_gpu_adv.py -k c3 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/pytest.log
