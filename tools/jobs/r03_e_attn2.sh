#!/bin/bash
# round 3, call E: the pipelined forward after the packed-fma / streamed-V^T-read changes, against the first kernel (variants 0 and 1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_kernels.py -k "attention" > $O/pytest_attn.log 2>&1; echo "attn rc=$?" >> $O/rc.log
timeout 600 python tools/attn_fwd_variants.py 0,1 > $O/attn_variants.txt 2>&1; echo "variants rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 2 $O/pytest_attn.log; cat $O/attn_variants.txt
