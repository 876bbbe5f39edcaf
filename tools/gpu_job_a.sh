#!/bin/bash
# round-2 GPU job A: packed-attention fix at full width, both full-size tests, full-size SDXL / SD3 timing + kernel traces
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/a; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" > gpurun_out/a/attn.log 2>&1; echo "attn rc=$?" >> gpurun_out/a/rc.log
timeout 900 python -m pytest tests/test_gpu_zy_sdxl_fullsize.py tests/test_gpu_zz_sd3_fullsize.py -x -q -s > gpurun_out/a/fullsize.log 2>&1; echo "fullsize rc=$?" >> gpurun_out/a/rc.log
timeout 600 python tools/sdxl_step_probe.py 4 > gpurun_out/a/sdxl_probe.log 2>&1; echo "sdxl rc=$?" >> gpurun_out/a/rc.log
timeout 600 python tools/sd3_step_probe.py 2 graph > gpurun_out/a/sd3_probe.log 2>&1; echo "sd3 rc=$?" >> gpurun_out/a/rc.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/a/prof_sd3 -- python $GRAFT_REPO_ROOT/tools/sd3_step_probe.py 2 > $GRAFT_REPO_ROOT/gpurun_out/a/sd3_prof.log 2>&1; echo "sd3prof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/a/rc.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/a/prof_sdxl -- python $GRAFT_REPO_ROOT/tools/sdxl_step_probe.py 4 > $GRAFT_REPO_ROOT/gpurun_out/a/sdxl_prof.log 2>&1; echo "sdxlprof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/a/rc.log
cd $GRAFT_REPO_ROOT
# keep only the stats CSVs (traces are large)
find gpurun_out/a -name "*kernel_trace.csv" -delete; find gpurun_out/a -name "*.db" -delete
cat gpurun_out/a/rc.log
