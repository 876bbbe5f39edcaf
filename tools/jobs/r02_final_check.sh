#!/bin/bash
# last check of the round on the final tree: the whole GPU suite, smoke, bench line (no cpu_baseline leg: profiles/r02_h_bench_line.json has it),
# rocprofv3 kernel-trace summary
cd $GRAFT_REPO_ROOT; O=gpurun_out/final2; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout 600 python bench.py --no-cpu-baseline 2> $O/bench.err > $O/bench.json; echo "bench rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_f -name "*.db" | head -1) 60 > $O/kernel_stats.txt 2>&1
cat $O/rc.log; tail -n 3 $O/pytest_gpu.log; tail -n 2 $O/smoke.log; cat $O/bench.json | cut -c1-900
