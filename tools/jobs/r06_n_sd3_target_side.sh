#!/bin/bash
# round 6, call N: SD3 step with the no-grad target pass on a second stream beside the online pass (PCM_SD3_ONLINE_TARGET=side; it was the default when this job ran) vs one launch chain (=serial): bitwise test,
# full-size parity test, C5 A/B on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06n; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_mmdit.py tests/test_gpu_zz_sd3_fullsize.py -q -x --durations=5 > $O/pytest_sd3.log 2>&1; echo "pytest sd3 rc=$?" >> $O/rc.log
for r in 1 2; do
  PCM_SD3_ONLINE_TARGET=side timeout 600 python bench.py --config c5 --steps 10 --warmup 4 > $O/bench_c5_side_$r.json 2>> $O/bench_c5.err; echo "c5 side rc=$?" >> $O/rc.log
  PCM_SD3_ONLINE_TARGET=serial timeout 600 python bench.py --config c5 --steps 10 --warmup 4 > $O/bench_c5_noside_$r.json 2>> $O/bench_c5.err; echo "c5 noside rc=$?" >> $O/rc.log
done
PCM_SD3_ONLINE_TARGET=side timeout 600 python bench.py --config c5 --batch 4 --steps 8 --warmup 3 > $O/bench_c5_b4_side.json 2>> $O/bench_c5.err; echo "c5 b4 side rc=$?" >> $O/rc.log
PCM_SD3_ONLINE_TARGET=serial timeout 600 python bench.py --config c5 --batch 4 --steps 8 --warmup 3 > $O/bench_c5_b4_noside.json 2>> $O/bench_c5.err; echo "c5 b4 noside rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 9 $O/pytest_sd3.log; for f in $O/bench_*.json; do echo -n "$f: "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done; grep -v "amdgpu.ids\|model ready" $O/bench_c5.err | tail -n 5
