#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/k; mkdir -p $O
timeout 120 tools/probes/mfmavalu > $O/mfmavalu.txt 2>&1; echo "mfmavalu rc=$?" >> $O/rc.log
cat $O/mfmavalu.txt
