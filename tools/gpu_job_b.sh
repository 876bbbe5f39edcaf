#!/bin/bash
# SD1.5 bs-16 picture: per-shape GEMM table (with plan codes), rocprofv3 kernel trace summary
cd $GRAFT_REPO_ROOT; O=gpurun_out/b2; mkdir -p $O; export TMPDIR=/tmp
PCM_GEMM_TABLE=$O/gemm_table.txt timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python tools/prof_summary.py $DB 70 > $O/kernel_stats.txt 2>&1
cat $O/rc.log
