#!/bin/bash
# round 4, call N (second pass: new short-K rule as the planner arm, forced phased-tile plans, SD3 / SDXL shapes): plan sweep on the small-M GEMM shapes of the deep UNet levels (tools/gemm_small_m.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04n; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/gemm_small_m.py > $O/small_m_sweep2.txt 2> $O/small_m_sweep.err; echo "sweep rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/small_m_sweep2.txt; tail -3 $O/small_m_sweep.err
