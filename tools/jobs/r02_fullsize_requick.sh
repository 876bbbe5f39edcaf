#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/x; mkdir -p $O
timeout 300 python tools/sdxl_step_probe.py 4 > $O/sdxl_probe.txt 2>&1; echo "sdxl rc=$?" >> $O/rc.log
timeout 300 python tools/sd3_step_probe.py 2 graph > $O/sd3_probe.txt 2>&1; echo "sd3 rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/sdxl_probe.txt; tail -n 4 $O/sd3_probe.txt
