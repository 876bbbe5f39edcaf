#!/bin/bash
# round 3, call P: attention GPU tests on the final build (kernel signature changed by the opt-in XCD map)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03p; mkdir -p $O
timeout 80 python -m pytest -m gpu -x -q tests/test_gpu_kernels.py -k "attention" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 2 $O/pytest.log
