#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/u; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q > $O/test_gemm.log 2>&1; echo "gemm rc=$?" >> $O/rc.log
timeout 600 python tools/gemm_conv_order_ab.py > $O/conv_ab.txt 2>&1; echo "ab rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 2 $O/test_gemm.log; cat $O/conv_ab.txt | cut -c1-230
