#!/bin/bash
# round 4, call G: why did the epilogues WITHOUT a residual get 3-6 % slower with the rewritten epilogue?  Three libraries on the same shapes:
# A = previous commit (old epilogue), B = this tree, C = this tree with the unrolled residual path compiled out (-DPCM_EPI_NO_RES_PATH: 6 KB
# less code in the 55 KB kernel)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04g; mkdir -p $O; export TMPDIR=/tmp
L=phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so
PCM_GEMM_BIG=4 timeout 600 python tools/gemm_ab_libs.py tools/probes/libpcm_base.so $L tools/probes/libpcm_nores.so > $O/ab_libs_8p.txt 2>&1; echo "ab libs rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/ab_libs_8p.txt
