#!/bin/bash
# round 3, call G: the tests that changed after call F (20-step curve with the matched oracle on 3 steps, per-module graph-vs-eager gradient
# checks, the pipelined attention forward forced on every head dim), with durations
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03g; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest -m gpu -x -q --durations=8 tests/test_gpu_bench_config.py tests/test_gpu_sdxl.py tests/test_gpu_kernels.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 16 $O/pytest.log
