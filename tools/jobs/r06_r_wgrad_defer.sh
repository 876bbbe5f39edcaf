#!/bin/bash
# round 6, call R: LoRA weight-gradient jobs of a backward collected across modules into few pcm_lora_wgrad_multi_bf16 launches (PCM_WGRAD_DEFER=n
# modules per flush; host side only): parity tests with it on, step A/B on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06r; mkdir -p $O; export TMPDIR=/tmp
PCM_WGRAD_DEFER=32 timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_bench_config.py tests/test_gpu_sdxl.py tests/test_gpu_adv.py -q -x -m gpu > $O/pytest_defer.log 2>&1; echo "pytest (defer 32) rc=$?" >> $O/rc.log
for r in 1 2; do for n in 0 16 32 64 200; do
  PCM_WGRAD_DEFER=$n timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_defer${n}_$r.json 2>> $O/bench.err; echo "c2 defer $n rc=$?" >> $O/rc.log
done; done
for n in 0 32; do PCM_WGRAD_DEFER=$n timeout 600 python bench.py --config c3 --steps 10 --warmup 4 > $O/bench_c3_defer$n.json 2>> $O/bench.err; PCM_WGRAD_DEFER=$n timeout 600 python bench.py --config c4 --steps 10 --warmup 4 > $O/bench_c4_defer$n.json 2>> $O/bench.err; done
cat $O/rc.log | sort | uniq -c; tail -n 4 $O/pytest_defer.log; for f in $O/bench_*.json; do echo -n "$(basename $f): "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done
