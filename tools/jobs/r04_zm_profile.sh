#!/bin/bash
# round 4, call ZM: rocprofv3 kernel-trace summary and the roofline leg (per-shape GEMM table) of the end-of-round tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zm; mkdir -p $O; export TMPDIR=/tmp
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?" >> $O/rc.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_zm -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_zm -name "*.db" | head -1) 70 > $O/kernel_stats_bench_bs16.txt 2>&1; echo "summary rc=$?" >> $O/rc.log
cat $O/rc.log; head -8 $O/kernel_stats_bench_bs16.txt | cut -c1-150; cut -c1-200 $O/bench_c2.json
