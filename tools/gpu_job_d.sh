#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_config.py -x -q -s > $O/test_bench_config.log 2>&1; echo "test rc=$?" >> $O/rc.log
timeout 300 python tools/error_budget.py --tiny > $O/error_budget_tiny.log 2>&1; echo "eb tiny rc=$?" >> $O/rc.log
timeout 900 python tools/error_budget.py > $O/error_budget.log 2>&1; echo "eb rc=$?" >> $O/rc.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -30 $O/test_bench_config.log
