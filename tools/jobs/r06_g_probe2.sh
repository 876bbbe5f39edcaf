#!/bin/bash
# round 6, call G: two-stream overlap probe incl. the cross-step pairing (student backward of step k || teacher pass of step k+1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/two_stream_probe.py --reps 5 > $O/two_stream.txt 2>&1
timeout 600 python tools/two_stream_probe.py --reps 5 >> $O/two_stream.txt 2>&1
grep -a "ms (eager" $O/two_stream.txt
