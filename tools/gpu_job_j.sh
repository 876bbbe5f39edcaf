#!/bin/bash
# gemm8p: persistent tile loop + register-side bias / unrolled store loop, A/B against the round-2 baseline build
cd $GRAFT_REPO_ROOT; O=gpurun_out/j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q > $O/test_gemm.log 2>&1; echo "gemm rc=$?" >> $O/rc.log
timeout 600 python tools/gemm_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/ab.txt 2>&1; echo "ab rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> $O/bench.err > $O/bench.json; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -2 $O/test_gemm.log; cat $O/ab.txt; grep -i "timed\|two-timestep" $O/bench.err; cat $O/bench.json | cut -c1-400
