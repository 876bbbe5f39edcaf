#!/bin/bash
# round 5, call D: (1) where the captured-graph forward+backward differs from the eager one (tools/graph_vs_eager.py, atomics vs reproducible
# reductions); (2) the whole GPU suite on the tree with the regenerated rounding-matched fixtures, DMA-staged attention, batched text K/V;
# (3) the batched cross-attention K/V projection of the frozen pass off / on (PCM_TEXT_KV), whole C2 step, interleaved twice
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
timeout 400 python tools/graph_vs_eager.py 16 > $O/graph_vs_eager.txt 2>&1; echo "gve rc=$?" >> $O/rc.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
for i in 1 2; do
  PCM_TEXT_KV=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_textkv_off_$i.txt
  PCM_TEXT_KV=1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_textkv_on_$i.txt
done
cp gpurun_out/*.json $O/ 2>/dev/null
cat $O/rc.log; cat $O/graph_vs_eager.txt; tail -n 22 $O/pytest_gpu.log; tail -n 3 $O/smoke.log; for f in $O/step_textkv_*; do echo $f; cat $f; done
