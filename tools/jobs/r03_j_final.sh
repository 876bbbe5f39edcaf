#!/bin/bash
# round 3, call J: the by-shape conv K order on hardware (GEMM tests incl. the 8x8 case, the full-size step parity test, the bench line)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03j; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_gemm.py > $O/pytest_gemm.log 2>&1; echo "gemm rc=$?" >> $O/rc.log
timeout 900 python -m pytest -m gpu -x -q -s "tests/test_gpu_step.py::test_full_step_vs_oracle[0.02]" > $O/pytest_step.log 2>&1; echo "step rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 2 $O/pytest_gemm.log; grep "b_std" $O/pytest_step.log | cut -c1-500; tail -n 2 $O/pytest_step.log; grep "timed\|two-timestep" $O/bench.err; cut -c1-300 $O/bench.json
