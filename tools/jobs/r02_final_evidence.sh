#!/bin/bash
# round-2 final evidence: the whole GPU suite, smoke, the default bench line (with the C1 cpu_baseline), rocprofv3 kernel-trace summary of
# the same workload, the PMC traffic passes on the dominant kernel, and the three "next row" configurations at full size
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout 900 python bench.py 2> $O/bench.err > $O/bench.json; echo "bench rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_f -name "*.db" | head -1) 60 > $O/kernel_stats.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/tools/pmc_gemm8p.py > $GRAFT_REPO_ROOT/$O/pmc_f.log 2>&1); echo "pmc_f rc=$?" >> $O/rc.log
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/tools/pmc_gemm8p.py > $GRAFT_REPO_ROOT/$O/pmc_w.log 2>&1); echo "pmc_w rc=$?" >> $O/rc.log
python tools/pmc_traffic_json.py $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; echo "pmcjson rc=$?" >> $O/rc.log
timeout 300 python tools/sdxl_step_probe.py 4 > $O/sdxl_probe.txt 2>&1; echo "sdxl rc=$?" >> $O/rc.log
timeout 300 python tools/sd3_step_probe.py 2 graph > $O/sd3_probe.txt 2>&1; echo "sd3 rc=$?" >> $O/rc.log
timeout 300 python tools/adv_step_probe.py 8 > $O/adv_probe.txt 2>&1; echo "adv rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; cat $O/bench.json; tail -3 $O/sdxl_probe.txt $O/sd3_probe.txt $O/adv_probe.txt
