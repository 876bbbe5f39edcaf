#!/bin/bash
# round 4, call I: forward attention with the S / O accumulators in ACCUMULATOR registers (inline-asm MFMAs with 'a' operands, 63 accvgpr
# reads per tile, two workgroups per CU) against the shipped VGPR-form kernel (three per CU): libpcm_agacc.so = this tree + -DPCM_ATTN_AGPR_ACC
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04i; mkdir -p $O; export TMPDIR=/tmp
L=phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so
timeout 600 python tools/attn_ab_libs.py $L tools/probes/libpcm_agacc.so > $O/attn_ab_agpr.txt 2>&1; echo "attn ab rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/attn_ab_agpr.txt
