#!/bin/bash
# round 6, call ZZZ: the tree after the last kernel-source edit (gemm_ws.hip / pcm_common.h: tools-build code only, but the source id changes):
# whole GPU suite, smoke, PMC table stamped with THIS id, then the default bench command and the driver's command line against it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzz; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; print([capi.Lib(p).build_id for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)])" > $O/libs.log 2>&1 || { cat $O/libs.log; exit 7; }
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
PCM_JOB_OUT=r06zzz bash tools/jobs/r06_pmc.sh > $O/pmc_job.log 2>&1
cp $O/pmc_step_table.txt profiles/r06_zzz_pmc_step_table.txt
timeout 900 python bench.py > $O/bench_c2_default_flags.json 2> $O/bench_c2.err; echo "bench c2 (default flags) rc=$?" >> $O/rc.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_cmd.json 2>> $O/bench_c2.err; echo "bench c2 (driver's command) rc=$?" >> $O/rc.log
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --steps 10 --warmup 4 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.log; done
cat $O/libs.log $O/rc.log; tail -n 14 $O/pytest_gpu.log; tail -n 4 $O/smoke.log; head -n 4 $O/pmc_step_table.txt | cut -c1-170
for f in default_flags driver_cmd; do python - $O/bench_c2_$f.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], "frac", r["frac"], "avg_launch_us", r.get("avg_launch_us"), "fwd2t", r["student_fwd_2t"]["ms"], r["student_fwd_2t"]["frac"], "traffic_source", r["traffic_source"][:100], "idle", d["config"].get("host_ms_per_step_idle_queue"), "cpu", (d.get("cpu_baseline") or {}).get("value"), d.get("build_id"))
PY
done
for f in c3 c4 c5; do grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/bench_$f.json | sed "s/^/$f /"; done
