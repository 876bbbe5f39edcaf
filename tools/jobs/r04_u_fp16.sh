#!/bin/bash
# round 4, call U: first run of the IEEE-half build (lib/libpcm_hip_f16.so): one SD1.5-size step against the fp32-oracle fixture, bf16 beside fp16
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04u; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/fp16_step_probe.py $O/fp16_step_probe.json > $O/fp16_step_probe.txt 2> $O/fp16_step_probe.err; echo "probe rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/fp16_step_probe.txt; tail -5 $O/fp16_step_probe.err
