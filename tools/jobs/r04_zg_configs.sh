#!/bin/bash
# round 4, call ZG: bench lines of the other BASELINE configs on the final tree (c3 adversarial SD1.5, c4 SDXL, c5 SD3-medium), c2 beside them on the same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zg; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.log
for c in c3 c4 c5; do
  timeout 600 python bench.py --config $c --steps 8 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.log
done
cat $O/rc.log; for c in c2 c3 c4 c5; do grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/bench_$c.json | sed "s/^/$c /"; done
