#!/bin/bash
# round 4, call Z: final tree -- whole GPU suite (durations), smoke, the driver's default bench command (roofline + cpu_baseline legs), per-shape
# GEMM table, rocprofv3 kernel-trace summary of the eager step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04z; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 900 python bench.py > $O/bench_c2_default_flags.json 2> $O/bench_c2.err; echo "bench c2 (default flags) rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_12_steps.json 2>> $O/bench_c2.err; echo "bench c2 12 steps rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_z -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_z -name "*.db" | head -1) 70 > $O/kernel_stats_bench_bs16.txt 2>&1; echo "summary rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 28 $O/pytest_gpu.log; tail -n 4 $O/smoke.log; cut -c1-600 $O/bench_c2_default_flags.json; cut -c1-300 $O/bench_c2_12_steps.json
