#!/bin/bash
# round 3, call I: LoRA-gradient floor at the real SD1.5 size (tools/grad_floor.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03i; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python tools/grad_floor.py > $O/grad_floor.log 2>&1; echo "grad_floor rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 12 $O/grad_floor.log | cut -c1-400
