#!/bin/bash
# round 3, call C: bench lines of all four BASELINE configs on the round-3 tree (c2 with the MFMA-bound / HBM-bound launch classes of the
# dominant kernel; c3 / c4 / c5 with the whole-step roofline from the FLOP walk), the 2-rank rehearsal of the SEGMENTED adversarial capture
# (both ranks on device 0, gloo: graph replay vs eager launches must report the same losses), rocprofv3 kernel-trace summaries of c2 and c3
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03c; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.log
timeout 600 python bench.py --config c3 --steps 8 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/rc.log
timeout 600 python bench.py --config c4 --steps 6 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?" >> $O/rc.log
timeout 600 python bench.py --config c5 --steps 6 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/rc.log
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 900 python bench.py --config c3 --gpus 2 --batch 4 --steps 4 --warmup 2 > $O/bench_c3_2rank_graph.json 2> $O/bench_c3_2rank_graph.err; echo "c3 2-rank graph rc=$?" >> $O/rc.log
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 900 python bench.py --config c3 --gpus 2 --batch 4 --steps 4 --warmup 2 --no-graph > $O/bench_c3_2rank_eager.json 2> $O/bench_c3_2rank_eager.err; echo "c3 2-rank eager rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_c2.log 2>&1); echo "prof c2 rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_c2 -name "*.db" | head -1) 70 > $O/kernel_stats_c2.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o b -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 4 --warmup 2 --no-graph > $GRAFT_REPO_ROOT/$O/prof_c3.log 2>&1); echo "prof c3 rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_c3 -name "*.db" | head -1) 50 > $O/kernel_stats_c3.txt 2>&1
cat $O/rc.log; for f in c2 c3 c4 c5 c3_2rank_graph c3_2rank_eager; do echo "== $f"; tail -n 2 $O/bench_$f.err; cut -c1-1500 $O/bench_$f.json; done; head -30 $O/kernel_stats_c3.txt
