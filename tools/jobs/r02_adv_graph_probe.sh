#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/z; mkdir -p $O
timeout 500 python tools/adv_step_probe.py 8 graph > $O/adv_probe.txt 2>&1; echo "adv probe rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 9 $O/adv_probe.txt
