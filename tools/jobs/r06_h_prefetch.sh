#!/bin/bash
# round 6, call H: cross-step teacher prefetch (capture(pipeline=True)): bitwise test, bench A/B prefetch on / off / round-5 tree, whole suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_config.py -q -x -k "pipelined or graph_replay_equals" --durations=3 > $O/pytest_pipe.log 2>&1; echo "pytest pipe rc=$?" >> $O/rc.log
for r in 1 2; do
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_prefetch_$r.json 2>> $O/bench_c2.err; echo "bench rc=$?" >> $O/rc.log
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-prefetch > $O/bench_c2_noprefetch_$r.json 2>> $O/bench_c2.err; echo "bench nopf rc=$?" >> $O/rc.log
  (cd tools/probes/base_tree && timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline) > $O/bench_c2_base_$r.json 2>> $O/bench_c2.err; echo "bench base rc=$?" >> $O/rc.log
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_default.json 2>> $O/bench_c2.err; echo "bench default rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $O/bench_c2_eager_prefetch.json 2>> $O/bench_c2.err; echo "bench eager rc=$?" >> $O/rc.log
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 5 $O/pytest_pipe.log; tail -n 6 $O/pytest_gpu.log; for f in $O/bench_*.json; do echo -n "$f: "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done; tail -n 5 $O/bench_c2.err
