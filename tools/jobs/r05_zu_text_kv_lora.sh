#!/bin/bash
# round 5, call ZU: LoRA pass -- pass-wide rank-64 down-projection of the text + one K|V GEMM per cross-attention block (PCM_TEXT_KV_LORA,
# pcm_amd/model.py UNet._text_kv_t / _attn_fwd) off / on: whole C2 step interleaved twice, then the whole GPU suite with it on
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05zu; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
for i in 1 2; do
  PCM_TEXT_KV_LORA=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_off_$i.txt
  PCM_TEXT_KV_LORA=1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_on_$i.txt
done
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
cat $O/rc.log; for f in $O/step_*; do echo "$f $(grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f)"; done; tail -n 10 $O/pytest_gpu.log
