#!/bin/bash
# round 5, call C: (1) LDS-DMA double-buffered staging of the pre-scaled-query attention kernels against their register staging: per kernel
# (tools/attn_ps_ab.py, tools build, hook) and on the whole C2 step (product library vs tools/probes/libpcm_nodma.so = the same tree built with
# -DPCM_ATTN_PS_DMA_DEFAULT=0), interleaved twice; (2) run-to-run repeatability of the LoRA gradient, atomics vs reproducible forms
# (tools/grad_repeatability.py); (3) the attention kernel tests on both stagings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.TOOLS_LIB, \"tools/probes/libpcm_nodma.so\")]; print(\"libs load\")" || exit 7
timeout 300 python tools/attn_ps_ab.py 3 > $O/attn_ps_ab.txt 2>&1; echo "attn ab rc=$?" >> $O/rc.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_product_lib.py -q -k "attention or product" > $O/pytest_attention.txt 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 300 python tools/grad_repeatability.py 16 > $O/grad_repeatability.txt 2>&1; echo "repeat rc=$?" >> $O/rc.log
for i in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_dma_$i.txt; echo "bench dma $i rc=$?" >> $O/rc.log
  timeout 300 python tools/bench_with_lib.py tools/probes/libpcm_nodma.so --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_nodma_$i.txt; echo "bench nodma $i rc=$?" >> $O/rc.log
done
cat $O/rc.log; cat $O/attn_ps_ab.txt; tail -n 3 $O/pytest_attention.txt; cat $O/grad_repeatability.txt; for f in $O/step_*; do echo $f; cat $f; done
