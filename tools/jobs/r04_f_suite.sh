#!/bin/bash
# round 4, call F: the whole GPU suite with the oracle side coming from the committed fixtures (durations), smoke, bench lines of all four
# configs, PMC traffic passes on the dominant kernel's largest launch, per-shape GEMM table
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04f; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?" >> $O/rc.log
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?" >> $O/rc.log; done
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/tools/pmc_gemm8p.py > $GRAFT_REPO_ROOT/$O/pmc_f.log 2>&1); echo "pmc_f rc=$?" >> $O/rc.log
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/tools/pmc_gemm8p.py > $GRAFT_REPO_ROOT/$O/pmc_w.log 2>&1); echo "pmc_w rc=$?" >> $O/rc.log
python tools/pmc_traffic_json.py $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; echo "pmcjson rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 40 $O/pytest_gpu.log; tail -n 3 $O/smoke.log; for c in c2 c3 c4 c5; do cut -c1-330 $O/bench_$c.json; done
