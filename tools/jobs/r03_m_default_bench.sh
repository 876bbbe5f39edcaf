#!/bin/bash
# round 3, call M: the driver's exact bench command (default flags: N = 1, cpu_baseline leg with its 3 timed oracle steps), wall time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03m; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/time.log; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/time.log; tail -n 4 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d['cpu_baseline']))"
