#!/bin/bash
# round 4, call ZH: tests/test_gpu_fp16.py in full on the final tree (durations)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zh; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fp16.py -q --durations=10 > $O/pytest_fp16.txt 2>&1; echo "rc=$?" >> $O/rc.log
cat $O/rc.log; tail -16 $O/pytest_fp16.txt
