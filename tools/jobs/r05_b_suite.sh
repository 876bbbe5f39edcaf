#!/bin/bash
# round 5, call B: the tree with the UNet on the pre-scaled-query attention kernels and the lean product library: whole GPU suite (the
# rounding-matched fixtures are still the round-4 ones at this point: their three tests are expected to move), smoke, C2 bench line with
# the per-shape GEMM table, rocprofv3 kernel-trace summary
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05b; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?" >> $O/rc.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_b -name "*.db" | head -1) 70 > $O/kernel_stats_bench_bs16.txt 2>&1; echo "summary rc=$?" >> $O/rc.log
cp gpurun_out/*.json $O/ 2>/dev/null
cat $O/rc.log; tail -n 25 $O/pytest_gpu.log; tail -n 3 $O/smoke.log; head -30 $O/kernel_stats_bench_bs16.txt | cut -c1-150; cut -c1-260 $O/bench_c2.json
