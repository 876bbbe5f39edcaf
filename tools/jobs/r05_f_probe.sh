#!/bin/bash
# round 5, call F: (1) dK/dV kernel of the pre-scaled-query attention at 3 waves per SIMD (168 VGPRs, 38 spilled) against the shipped 2
# (tools/probes/libpcm_dkdvlb3.so as the alternate library of tools/attn_ps_ab.py); (2) graph vs eager with the REGISTER-staged attention
# kernels (tools/probes/libpcm_nodma.so): call B's suite saw 1.65e-5 on the gradient there, call D with the DMA staging 1.2e-7
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05f; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/attn_ps_ab.py 3 tools/probes/libpcm_dkdvlb3.so > $O/attn_ps_ab_dkdv_3waves.txt 2>&1; echo "ab rc=$?" >> $O/rc.log
timeout 400 python tools/graph_vs_eager.py 16 tools/probes/libpcm_nodma.so > $O/graph_vs_eager_register_staging.txt 2>&1; echo "gve rc=$?" >> $O/rc.log
cat $O/rc.log; cut -c1-60,330-420 $O/attn_ps_ab_dkdv_3waves.txt; cat $O/graph_vs_eager_register_staging.txt
