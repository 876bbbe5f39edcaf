#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/e; mkdir -p $O; export TMPDIR=/tmp; rm -f $O/rc.log
timeout 900 python -m pytest tests/test_gpu_bench_config.py -q -s > $O/test_bench_config.log 2>&1; echo "bench_config rc=$?" >> $O/rc.log
timeout 300 python tools/wgrad_probe.py > $O/wgrad_probe.txt 2>&1
PCM_WGRAD_SIDE=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_noside.json 2> $O/bench_noside.err; echo "bench noside rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -5 $O/test_bench_config.log; cat $O/wgrad_probe.txt; grep timed $O/bench_noside.err $O/bench.err
