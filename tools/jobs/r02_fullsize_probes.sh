#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/g; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/adv_step_probe.py 8 > $O/adv_probe.txt 2>&1; echo "adv rc=$?" >> $O/rc.log
timeout 600 python tools/sdxl_step_probe.py 4 > $O/sdxl_probe.txt 2>&1; echo "sdxl rc=$?" >> $O/rc.log
timeout 600 python tools/sd3_step_probe.py 2 graph > $O/sd3_probe.txt 2>&1; echo "sd3 rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sd3 -o p -- python $GRAFT_REPO_ROOT/tools/sd3_step_probe.py 2 > /dev/null 2>&1); python tools/prof_summary.py $(find /tmp/prof_sd3 -name "*.db" | head -1) 30 > $O/sd3_kernel_stats.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sdxl -o p -- python $GRAFT_REPO_ROOT/tools/sdxl_step_probe.py 4 > /dev/null 2>&1); python tools/prof_summary.py $(find /tmp/prof_sdxl -name "*.db" | head -1) 30 > $O/sdxl_kernel_stats.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_adv -o p -- python $GRAFT_REPO_ROOT/tools/adv_step_probe.py 8 > /dev/null 2>&1); python tools/prof_summary.py $(find /tmp/prof_adv -name "*.db" | head -1) 30 > $O/adv_kernel_stats.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_bench_config.py -q -k split > $O/test_split.log 2>&1; echo "split rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/adv_probe.txt; tail -4 $O/sdxl_probe.txt; tail -5 $O/sd3_probe.txt; tail -3 $O/test_split.log
