#!/bin/bash
# round 3, call F: the whole GPU suite on the final tree (with per-test durations), smoke, the default bench line (roofline with launch
# classes, host time on an idle queue, per-shape GEMM table), the PMC traffic passes on the dominant kernel's largest launch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03f; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --durations=30 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/tools/pmc_gemm8p.py > $GRAFT_REPO_ROOT/$O/pmc_f.log 2>&1); echo "pmc_f rc=$?" >> $O/rc.log
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/tools/pmc_gemm8p.py > $GRAFT_REPO_ROOT/$O/pmc_w.log 2>&1); echo "pmc_w rc=$?" >> $O/rc.log
python tools/pmc_traffic_json.py $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; echo "pmcjson rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 45 $O/pytest_gpu.log; tail -n 3 $O/smoke.log; tail -n 3 $O/bench.err; cut -c1-600 $O/bench.json; cat $O/pmc_traffic.json | head -30
