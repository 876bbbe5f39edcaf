#!/bin/bash
# round 4, call ZA: the live-oracle GPU tests under the 16-thread cap of tests/conftest.py (they were 9-12 s on one box, 70-103 s on another)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04za; mkdir -p $O; export TMPDIR=/tmp
nproc > $O/host.txt; uptime >> $O/host.txt
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_sdxl.py tests/test_gpu_mmdit.py -q --durations=10 > $O/pytest_live_oracle.txt 2>&1; echo "rc=$?" >> $O/rc.log
cat $O/rc.log $O/host.txt; tail -16 $O/pytest_live_oracle.txt
