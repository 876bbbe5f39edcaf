#!/bin/bash
# round 6, call M: teacher prefetch for the adversarial SD1.5 step (capture_adv(pipeline=True)) and the SD3 step (SD3Distiller.capture(pipeline=True)):
# bitwise tests, then bench A/B prefetch on / off for c3 and c5 (c4 and the default line re-checked on the same box)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_deterministic_adv.py tests/test_gpu_mmdit.py tests/test_gpu_adv.py -q -x --durations=5 > $O/pytest_adv.log 2>&1; echo "pytest adv rc=$?" >> $O/rc.log
for r in 1 2; do
  for c in c3 c5; do
    timeout 600 python bench.py --config $c --steps 10 --warmup 4 > $O/bench_${c}_prefetch_$r.json 2>> $O/bench_$c.err; echo "$c prefetch rc=$?" >> $O/rc.log
    timeout 600 python bench.py --config $c --steps 10 --warmup 4 --no-prefetch > $O/bench_${c}_noprefetch_$r.json 2>> $O/bench_$c.err; echo "$c noprefetch rc=$?" >> $O/rc.log
  done
done
timeout 600 python bench.py --config c3 --batch 2 --steps 10 --warmup 4 > $O/bench_c3_b2_prefetch.json 2>> $O/bench_c3.err; echo "c3 b2 rc=$?" >> $O/rc.log
timeout 600 python bench.py --config c3 --batch 2 --steps 10 --warmup 4 --no-prefetch > $O/bench_c3_b2_noprefetch.json 2>> $O/bench_c3.err; echo "c3 b2 nopf rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2.json 2>> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 12 $O/pytest_adv.log; for f in $O/bench_*.json; do echo -n "$f: "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done; tail -n 5 $O/bench_c3.err $O/bench_c5.err
