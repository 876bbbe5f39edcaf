#!/bin/bash
# round 6, call I: LoRA weight gradients on a side stream re-measured on this tree (PCM_WGRAD_SIDE=1, with / without the teacher prefetch), the default
# bench command with its CPU leg (wall time of the whole command), kernel-trace summary of the eager step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2; do
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_prefetch_$r.json 2>> $O/bench_c2.err
  PCM_WGRAD_SIDE=1 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_prefetch_wgradside_$r.json 2>> $O/bench_c2.err
  PCM_WGRAD_SIDE=1 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-prefetch > $O/bench_c2_noprefetch_wgradside_$r.json 2>> $O/bench_c2.err
done
( time timeout 900 python bench.py > $O/bench_c2_default_flags.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt; echo "bench default rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph --no-prefetch > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_i -name "*.db" | head -1) 70 > $O/kernel_stats_bench_bs16.txt 2>&1
cat $O/rc.log; cat $O/bench_default_time.txt; for f in $O/bench_*.json; do echo -n "$f: "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done
