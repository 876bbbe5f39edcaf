#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/c; mkdir -p $O
timeout 300 python tools/gemm8p_ablate.py > $O/gemm8p_ablate.txt 2>&1
cat $O/gemm8p_ablate.txt
