#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" > $O/test_wgrad.log 2>&1; echo "wgrad test rc=$?" >> $O/rc.log
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_bench_config.py -x -q > $O/test_step.log 2>&1; echo "step test rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> $O/bench.err > $O/bench.json; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/test_wgrad.log; tail -3 $O/test_step.log; grep -i "timed\|two-timestep" $O/bench.err
