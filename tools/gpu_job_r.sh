#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_adv.py -x -q -k "graph_replay" > $O/test_adv_graph.log 2>&1; echo "adv graph test rc=$?" >> $O/rc.log
timeout 500 python tools/adv_step_probe.py 8 graph > $O/adv_probe.txt 2>&1; echo "adv probe rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 12 $O/test_adv_graph.log; tail -n 9 $O/adv_probe.txt
