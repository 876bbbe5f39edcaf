#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sdxl.py -x -q -k "topology_step" > $O/test_sdxl.log 2>&1; echo "sdxl test rc=$?" >> $O/rc.log
timeout 400 python tools/sdxl_step_probe.py 4 graph > $O/sdxl_probe.txt 2>&1; echo "sdxl probe rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 5 $O/test_sdxl.log; tail -n 6 $O/sdxl_probe.txt
