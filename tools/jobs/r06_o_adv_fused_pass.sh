#!/bin/bash
# round 6, call O: adversarial step with the online + target student forwards as ONE 2B-sample pass (as the plain step does): parity tests, C3 A/B;
# LoRA weight gradients on a second stream for the SD3 backward (PCM_SD3_WGRAD_SIDE=1) and the C3 student backward (PCM_WGRAD_SIDE=1): A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06o; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_adv.py tests/test_gpu_deterministic_adv.py tests/test_gpu_sdxl.py tests/test_gpu_fp16.py -q -x -k "adv or sdxl" --durations=5 > $O/pytest_adv.log 2>&1; echo "pytest adv rc=$?" >> $O/rc.log
for r in 1 2; do
  timeout 600 python bench.py --config c3 --steps 10 --warmup 4 > $O/bench_c3_fused_$r.json 2>> $O/bench_c3.err; echo "c3 fused rc=$?" >> $O/rc.log
  PCM_ADV_FUSE_PASSES=0 timeout 600 python bench.py --config c3 --steps 10 --warmup 4 > $O/bench_c3_two_passes_$r.json 2>> $O/bench_c3.err; echo "c3 two passes rc=$?" >> $O/rc.log
done
timeout 600 python bench.py --config c3 --batch 2 --steps 10 --warmup 4 > $O/bench_c3_b2_fused.json 2>> $O/bench_c3.err; echo "c3 b2 fused rc=$?" >> $O/rc.log
PCM_ADV_FUSE_PASSES=0 timeout 600 python bench.py --config c3 --batch 2 --steps 10 --warmup 4 > $O/bench_c3_b2_two_passes.json 2>> $O/bench_c3.err; echo "c3 b2 two rc=$?" >> $O/rc.log
for r in 1 2; do
  timeout 600 python bench.py --config c5 --steps 10 --warmup 4 > $O/bench_c5_wgrad_inline_$r.json 2>> $O/bench_c5.err; echo "c5 inline rc=$?" >> $O/rc.log
  PCM_SD3_WGRAD_SIDE=1 timeout 600 python bench.py --config c5 --steps 10 --warmup 4 > $O/bench_c5_wgrad_side_$r.json 2>> $O/bench_c5.err; echo "c5 wgrad side rc=$?" >> $O/rc.log
done
PCM_WGRAD_SIDE=1 timeout 600 python bench.py --config c3 --steps 10 --warmup 4 > $O/bench_c3_fused_wgrad_side.json 2>> $O/bench_c3.err; echo "c3 wgrad side rc=$?" >> $O/rc.log
PCM_SD3_WGRAD_SIDE=1 timeout 600 python -m pytest tests/test_gpu_mmdit.py -q -x > $O/pytest_sd3_wgrad_side.log 2>&1; echo "pytest sd3 wgrad side rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/pytest_sd3_wgrad_side.log; tail -n 9 $O/pytest_adv.log; for f in $O/bench_*.json; do echo -n "$f: "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done; grep -v "amdgpu.ids\|model ready" $O/bench_c3.err | tail -n 5
