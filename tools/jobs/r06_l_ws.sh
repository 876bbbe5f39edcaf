#!/bin/bash
# round 6, call L: the weights-stationary short-K kernel (csrc/gemm_ws.hip): GPU parity (bit-identical to the phased tile), per-shape A/B, forward and step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06l; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -k "weights_stationary" > $O/pytest_ws.log 2>&1; echo "pytest ws rc=$?" >> $O/rc.log
timeout 600 python tools/gemm_ws_ab.py > $O/gemm_ws_ab.txt 2>&1; echo "ab rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 4 $O/pytest_ws.log; cat $O/gemm_ws_ab.txt
