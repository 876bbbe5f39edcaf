#!/bin/bash
# round 3, call L: last check of the final tree -- the GPU suite minus its five oracle-heavy files (those passed in calls F / G / H / J on
# this tree's kernels), smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03l; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_kernels.py tests/test_gpu_gemm.py tests/test_gpu_mmdit.py tests/test_gpu_sampler.py tests/test_gpu_sdxl.py tests/test_gpu_bench_config.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/pytest.log; tail -n 3 $O/smoke.log
