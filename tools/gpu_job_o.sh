#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "geglu" > $O/test_geglu.log 2>&1; echo "geglu test rc=$?" >> $O/rc.log
timeout 600 python -m pytest tests/test_gpu_bench_config.py -x -q -k "graph" > $O/test_graph.log 2>&1; echo "graph test rc=$?" >> $O/rc.log
PCM_FUSE_GEGLU=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2> $O/bench_unfused.err > $O/bench_unfused.json; echo "bench0 rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> $O/bench.err > $O/bench.json; echo "bench rc=$?" >> $O/rc.log
PCM_FUSE_GEGLU=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2> $O/bench_unfused2.err > $O/bench_unfused2.json
cat $O/rc.log; tail -2 $O/test_geglu.log; tail -2 $O/test_graph.log; grep -i "timed\|two-timestep" $O/bench_unfused.err $O/bench.err $O/bench_unfused2.err
