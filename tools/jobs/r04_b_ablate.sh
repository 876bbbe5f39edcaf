#!/bin/bash
# round 4, call B: what bounds the short-K launches -- ablations of gemm8p / gemm4w (no stores / no epilogue / no MFMA / no DMA), the start
# stagger of gemm4w's odd workgroup slot, cycle stamps of one gemm4w tile
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04b; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/gemm4w_ablate.py > $O/gemm4w_ablate.txt 2> $O/gemm4w_ablate.err; echo "ablate rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/gemm4w_ablate.txt; tail -5 $O/gemm4w_ablate.err
