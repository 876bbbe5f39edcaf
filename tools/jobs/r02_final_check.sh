#!/bin/bash
# last check of the round on the final tree: the GPU suite (minus the two full-size adversarial cases that had just passed on this tree and
# the b_std = 0 variant of the full-size step parity, whose CPU oracle costs 3.5 GPU-box minutes), smoke, bench line, kernel-trace summary
cd $GRAFT_REPO_ROOT; O=gpurun_out/final2; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_adv.py::test_adv_step_full_size --deselect "tests/test_gpu_step.py::test_full_step_vs_oracle[0.0]" > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 4 $O/pytest_gpu.log
