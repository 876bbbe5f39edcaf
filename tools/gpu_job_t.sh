#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/t; mkdir -p $O
timeout 600 python tools/attn_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/attn_ab.txt 2>&1
cat $O/attn_ab.txt
