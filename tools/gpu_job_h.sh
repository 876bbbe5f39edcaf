#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q > $O/test_gemm.log 2>&1; echo "gemm rc=$?" >> $O/rc.log
timeout 600 python tools/gemm_conv_order_ab.py > $O/conv_order_ab.txt 2>&1
PCM_GEMM_CONV_MD=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2> $O/bench_nomd.err > $O/bench_nomd.json
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> $O/bench.err > $O/bench.json
cat $O/rc.log; tail -2 $O/test_gemm.log; cat $O/conv_order_ab.txt; grep timed $O/bench_nomd.err $O/bench.err
