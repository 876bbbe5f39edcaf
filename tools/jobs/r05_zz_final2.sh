#!/bin/bash
# round 5, call ZZ: the driver's default bench command (roofline + cpu_baseline legs), per-shape GEMM table and rocprofv3 kernel-trace summary on the
# FINAL tree (after the time_emb_proj / text K|V launch-count reductions), C3 / C4 / C5 lines beside it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05zz; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 900 python bench.py > $O/bench_c2_default_flags.json 2> $O/bench_c2.err; echo "bench c2 (default flags) rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_12_steps.json 2>> $O/bench_c2.err; echo "bench c2 12 steps rc=$?" >> $O/rc.log
for c in c3 c4 c5; do
  timeout 600 python bench.py --config $c --steps 8 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.log
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_z -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_z -name "*.db" | head -1) 70 > $O/kernel_stats_bench_bs16.txt 2>&1; echo "summary rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/smoke.log; cut -c1-400 $O/bench_c2_default_flags.json
for f in c2_12_steps c3 c4 c5; do grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/bench_$f.json | sed "s/^/$f /"; done; head -5 $O/kernel_stats_bench_bs16.txt | cut -c1-150
