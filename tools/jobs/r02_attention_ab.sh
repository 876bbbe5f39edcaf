#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or attn" > $O/test_attn.log 2>&1; echo "attn test rc=$?" >> $O/rc.log
timeout 600 python tools/attn_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/attn_ab.txt 2>&1; echo "ab rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2> $O/bench.err > $O/bench.json; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/test_attn.log; cat $O/attn_ab.txt; grep -i "timed\|two-timestep" $O/bench.err
