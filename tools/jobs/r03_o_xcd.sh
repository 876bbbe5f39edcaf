#!/bin/bash
# round 3, call O: A/B of the XCD-aware block map of the forward attention kernel (last GPU seconds of the round)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03o; mkdir -p $O
timeout 100 python tools/attn_xcd_ab.py > $O/xcd_ab.txt 2>&1; echo "rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/xcd_ab.txt
