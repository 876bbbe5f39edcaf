#!/bin/bash
# round 6, call P: SD3 step with the online + target forwards as ONE 2B-sample pass (PCM_SD3_ONLINE_TARGET=fused, default) against two passes on two
# streams (=side, job N's winner) and one launch chain (=serial): parity tests (incl. the full-size step against the fp32 oracle), C5 A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06p; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_mmdit.py tests/test_gpu_zz_sd3_fullsize.py tests/test_gpu_fp16.py -q -x -k "sd3 or mmdit or prefetch" --durations=5 > $O/pytest_sd3.log 2>&1; echo "pytest sd3 rc=$?" >> $O/rc.log
for r in 1 2; do
  for m in fused side serial; do
    PCM_SD3_ONLINE_TARGET=$m timeout 600 python bench.py --config c5 --steps 10 --warmup 4 > $O/bench_c5_${m}_$r.json 2>> $O/bench_c5.err; echo "c5 $m rc=$?" >> $O/rc.log
  done
done
for m in fused side; do
  PCM_SD3_ONLINE_TARGET=$m timeout 600 python bench.py --config c5 --batch 4 --steps 8 --warmup 3 > $O/bench_c5_b4_$m.json 2>> $O/bench_c5.err; echo "c5 b4 $m rc=$?" >> $O/rc.log
done
cat $O/rc.log; tail -n 9 $O/pytest_sd3.log; for f in $O/bench_*.json; do echo -n "$f: "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done; grep -v "amdgpu.ids\|model ready" $O/bench_c5.err | tail -n 5
