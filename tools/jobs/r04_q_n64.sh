#!/bin/bash
# round 4, call Q: the generalised K-split rank-64 projection kernel (32 rows per block, 128-column pieces) against the previous commit's
# library on the N = 64 shapes of all four configs; GEMM tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -k "rank64" > $O/pytest_n64.txt 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 600 python tools/n64_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/n64_ab.txt 2> $O/n64_ab.err; echo "ab rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest_n64.txt; cat $O/n64_ab.txt; tail -3 $O/n64_ab.err
