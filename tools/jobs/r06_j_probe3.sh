#!/bin/bash
# round 6, call J: the student's two-timestep pass as ONE 2B pass against two B passes on two streams
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06j; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/two_stream_probe.py --reps 5 > $O/two_stream.txt 2>&1
grep -a "ms (eager" $O/two_stream.txt | head -8
