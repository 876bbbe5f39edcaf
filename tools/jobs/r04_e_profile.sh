#!/bin/bash
# round 4, call E: the step after the epilogue rewrite -- teacher shared-prefix A/B, bench line with the roofline leg + per-shape GEMM table,
# rocprofv3 kernel-trace summary of the eager step, the C2-as-benchmarked fixture test
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04e; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  PCM_DEDUP_TEACHER=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_nodedup_$i.json 2> $O/bench_nodedup_$i.err; echo "nodedup $i rc=$?" >> $O/rc.log
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_dedup_$i.json 2> $O/bench_dedup_$i.err; echo "dedup $i rc=$?" >> $O/rc.log
done
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_e -name "*.db" | head -1) 70 > $O/kernel_stats_bench_bs16.txt 2>&1; echo "summary rc=$?" >> $O/rc.log
timeout 900 python -m pytest tests/test_gpu_bench_config.py -x -q -k "not curve" --durations=5 > $O/pytest_bench_config.txt 2>&1; echo "pytest bench_config rc=$?" >> $O/rc.log
cat $O/rc.log; for f in nodedup_1 dedup_1 nodedup_2 dedup_2; do echo "$f: $(cut -c80-170 $O/bench_$f.json)"; done; cut -c1-1500 $O/bench.json; head -45 $O/kernel_stats_bench_bs16.txt; tail -15 $O/pytest_bench_config.txt
