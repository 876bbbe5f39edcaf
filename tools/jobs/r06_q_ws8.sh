#!/bin/bash
# round 6, call Q: the weights-stationary kernel again -- (1) the four-wave form with UNTRACKED LDS accesses in its epilogue (hipcc had put
# s_waitcnt vmcnt(0) in front of every staging read / write, draining the prefetched tile each step), (2) the eight-wave form (two waves per SIMD,
# 40 columns each): GPU parity (bitwise against the phased tile), per-shape A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06q; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -k "weights_stationary" > $O/pytest_ws.log 2>&1; echo "pytest ws rc=$?" >> $O/rc.log
timeout 900 python tools/gemm_ws_ab.py > $O/gemm_ws_ab.txt 2>&1; echo "ab rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/pytest_ws.log; cat $O/gemm_ws_ab.txt
