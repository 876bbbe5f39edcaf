#!/bin/bash
# round 4, call V: GPU tests of the IEEE-half build (tests/test_gpu_fp16.py) + the bf16 kernel / step tests that share kernel_cases.py
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04v; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fp16.py -q -x -s --durations=8 > $O/pytest_fp16.txt 2>&1; echo "fp16 rc=$?" >> $O/rc.log
cp gpurun_out/fp16_*.json $O/ 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm.py -q -x > $O/pytest_bf16_kernels.txt 2>&1; echo "bf16 kernels rc=$?" >> $O/rc.log
cat $O/rc.log; tail -40 $O/pytest_fp16.txt; tail -3 $O/pytest_bf16_kernels.txt
