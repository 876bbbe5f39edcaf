#!/bin/bash
# round 4, call L: the dense conv3x3 weight-gradient kernel (wgrad_dense.hip) on hardware -- kernel tests, per-geometry A/B against the
# Cout/64 rank-64 launches it replaces (shipped M-split rule and forced splits), the C3 tests and bench line, kernel-trace summary of C3
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04l; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "wgrad" > $O/pytest_wgrad.txt 2>&1; echo "pytest wgrad rc=$?" >> $O/rc.log
timeout 300 python tools/wgrad_dense_ab.py > $O/ab_rule.txt 2> $O/ab_rule.err; echo "ab rule rc=$?" >> $O/rc.log
for ms in 1 2 4 8; do
  PCM_WGRAD_DENSE_MSPLIT=$ms timeout 300 python tools/wgrad_dense_ab.py > $O/ab_ms$ms.txt 2> $O/ab_ms$ms.err; echo "ab ms$ms rc=$?" >> $O/rc.log
done
timeout 600 python -m pytest tests/test_gpu_adv.py -q --durations=5 > $O/pytest_adv.txt 2>&1; echo "pytest adv rc=$?" >> $O/rc.log
timeout 600 python bench.py --config c3 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o b -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_c3.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_l -name "*.db" | head -1) 50 > $O/kernel_stats_c3.txt 2>&1; echo "summary rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 5 $O/pytest_wgrad.txt; cat $O/ab_rule.txt; for ms in 1 2 4 8; do echo "== msplit $ms"; cat $O/ab_ms$ms.txt; done; tail -n 8 $O/pytest_adv.txt; cut -c1-300 $O/bench_c3.json; head -20 $O/kernel_stats_c3.txt | cut -c1-160
