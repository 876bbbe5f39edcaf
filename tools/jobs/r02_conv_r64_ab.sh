#!/bin/bash
# halo-window conv LoRA down-projection (conv_r64.hip): GPU tests, per-shape timing against the generic tile, bench with / without
cd $GRAFT_REPO_ROOT; O=gpurun_out/w; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "conv_lora" > $O/test.log 2>&1; echo "test rc=$?" >> $O/rc.log
timeout 300 python tools/conv_r64_ab.py > $O/ab.txt 2>&1; echo "ab rc=$?" >> $O/rc.log
PCM_CONV_R64=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2> $O/bench_off.err > $O/bench_off.json
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2> $O/bench_on.err > $O/bench_on.json
PCM_CONV_R64=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2> $O/bench_off2.err > $O/bench_off2.json
cat $O/rc.log; tail -n 3 $O/test.log; cat $O/ab.txt; grep -h "timed" $O/bench_off.err $O/bench_on.err $O/bench_off2.err
