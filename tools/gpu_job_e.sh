#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/e; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_config.py -q -s > $O/test_bench_config.log 2>&1; echo "bench_config rc=$?" >> $O/rc.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q > $O/test_kernels.log 2>&1; echo "kernels rc=$?" >> $O/rc.log
timeout 300 python tools/attn_probe.py > $O/attn_probe.txt 2>&1
timeout 300 python tools/attn_ablate.py > $O/attn_ablate.txt 2>&1
timeout 300 python tools/wgrad_probe.py > $O/wgrad_probe.txt 2>&1
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/test_kernels.log; cat $O/attn_probe.txt $O/attn_ablate.txt $O/wgrad_probe.txt; tail -4 $O/bench.err
