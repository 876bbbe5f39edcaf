#!/bin/bash
# round 4, call ZB: the SDXL-topology step in both builds (refactored shared case)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zb; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sdxl.py "tests/test_gpu_fp16.py::test_fp16_sdxl_topology_step_vs_oracle" -q -s > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/rc.log
cat $O/rc.log; grep "sdxl-topology step" $O/pytest.txt; tail -3 $O/pytest.txt
