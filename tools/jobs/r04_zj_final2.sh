#!/bin/bash
# round 4, call ZJ: the tree as it stands at the end of the round -- whole GPU suite (durations), smoke, C2 and C5 bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zj; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/rc.log
timeout 300 python bench.py --config c5 --steps 8 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 18 $O/pytest_gpu.log; tail -n 3 $O/smoke.log; cut -c1-330 $O/bench_c2.json; cut -c1-330 $O/bench_c5.json
