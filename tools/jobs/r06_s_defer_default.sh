#!/bin/bash
# round 6, call S: weight-gradient jobs collected across modules as the default (PCM_WGRAD_DEFER=32): bitwise tests (UNet and MMDiT paths), C5 A/B,
# whole GPU suite, smoke, default bench lines of every config
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_mmdit.py -q -x -m gpu -k "collected_across" > $O/pytest_defer.log 2>&1; echo "pytest defer rc=$?" >> $O/rc.log
for r in 1 2; do for n in 0 32; do PCM_WGRAD_DEFER=$n timeout 600 python bench.py --config c5 --steps 10 --warmup 4 > $O/bench_c5_defer${n}_$r.json 2>> $O/bench.err; echo "c5 defer $n rc=$?" >> $O/rc.log; done; done
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout 900 python bench.py > $O/bench_c2_default_flags.json 2> $O/bench_c2.err; echo "bench c2 (default flags) rc=$?" >> $O/rc.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_cmd.json 2>> $O/bench_c2.err; echo "bench c2 (driver's command) rc=$?" >> $O/rc.log
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --steps 10 --warmup 4 > $O/bench_$c.json 2>> $O/bench.err; echo "$c rc=$?" >> $O/rc.log; done
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench2_rehearsal_c2.json 2> $O/bench2_c2.err; echo "2-rank rehearsal c2 rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/pytest_defer.log; tail -n 9 $O/pytest_gpu.log; tail -n 3 $O/smoke.log
for f in $O/bench_c5_defer*.json $O/bench_c3.json $O/bench_c4.json $O/bench_c5.json; do echo -n "$(basename $f): "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done
for f in default_flags driver_cmd; do python - $O/bench_c2_$f.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], "frac", r["frac"], "avg_launch_us", r.get("avg_launch_us"), "fwd2t", r["student_fwd_2t"]["ms"], r["student_fwd_2t"]["frac"], "step_frac", r["step_frac"], "traffic_source", r["traffic_source"][:60], "cpu", (d.get("cpu_baseline") or {}).get("value"), d.get("build_id"))
PY
done
cut -c1-300 $O/bench2_rehearsal_c2.json
