#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/k; mkdir -p $O
timeout 120 tools/probes/dmabench > $O/dmabench.txt 2>&1; echo "dmabench rc=$?" >> $O/rc.log
cat $O/dmabench.txt
