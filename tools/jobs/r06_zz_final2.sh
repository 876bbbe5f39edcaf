#!/bin/bash
# round 6, call ZZ: after job Z's PMC table was committed (so the default line names a table stamped with ITS library's source id): the default
# bench command again, the driver's own command line, a kernel-trace summary of the ONE-CHAIN eager step (--no-prefetch: per-kernel durations
# comparable with bench.py's event-timed figures, which come from an un-overlapped instrumented step), and the GPU suite once more with every
# torch.empty buffer poisoned (PCM_POISON_EMPTY=1: a kernel that reads an element before writing it turns the result into NaN)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zz; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_c2_default_flags.json 2> $O/bench_c2.err; echo "bench c2 (default flags) rc=$?" >> $O/rc.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_cmd.json 2>> $O/bench_c2.err; echo "bench c2 (driver's command) rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_zz -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph --no-prefetch > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_zz -name "*.db" | head -1) 70 > $O/kernel_stats_bench_bs16_one_chain.txt 2>&1; echo "summary rc=$?" >> $O/rc.log
PCM_POISON_EMPTY=1 timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_poisoned.log 2>&1; echo "pytest_gpu poisoned rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 6 $O/pytest_gpu_poisoned.log; head -n 6 $O/kernel_stats_bench_bs16_one_chain.txt | cut -c1-150
for f in default_flags driver_cmd; do python - $O/bench_c2_$f.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], "frac", r["frac"], "avg_launch_us", r.get("avg_launch_us"), "fwd2t", r["student_fwd_2t"]["ms"], r["student_fwd_2t"]["frac"], "traffic_source", r["traffic_source"][:90], "idle", d["config"].get("host_ms_per_step_idle_queue"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
done
