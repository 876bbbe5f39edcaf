#!/bin/bash
# round 5, call ZT: every resnet's time_emb_proj of a pass as ONE GEMM + scatter (PCM_TEMB_BATCH, pcm_amd/model.py UNet._temb_all) off / on:
# whole C2 step interleaved twice, the whole GPU suite with it on, C3 / C4 lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05zt; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
for i in 1 2; do
  PCM_TEMB_BATCH=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_temb_off_$i.txt
  PCM_TEMB_BATCH=1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_temb_on_$i.txt
done
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
for c in c3 c4; do
  PCM_TEMB_BATCH=0 timeout 400 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 | cut -c1-200 > $O/${c}_temb_off.txt
  PCM_TEMB_BATCH=1 timeout 400 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 | cut -c1-200 > $O/${c}_temb_on.txt
done
cat $O/rc.log; for f in $O/step_temb_* $O/c3_* $O/c4_*; do echo "$f $(grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f)"; done; tail -n 12 $O/pytest_gpu.log
