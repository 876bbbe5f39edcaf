#!/bin/bash
# round 6, call E: reproducible reductions of the adversarial trainers (C3 / C5), build-id checks, GroupNorm-backward residual fusion A/B, whole suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; print([capi.Lib(p).build_id for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)])" || exit 7
timeout 900 python -m pytest tests/test_gpu_deterministic_adv.py -q -x --durations=5 > $O/pytest_det.log 2>&1; echo "pytest det rc=$?" >> $O/rc.log
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 --deselect tests/test_gpu_deterministic_adv.py > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
for r in 1 2; do
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_new_$r.json 2>> $O/bench_c2.err; echo "bench rc=$?" >> $O/rc.log
  PCM_GN_BWD_ADD=0 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_noGNADD_$r.json 2>> $O/bench_c2.err; echo "bench noadd rc=$?" >> $O/rc.log
  (cd tools/probes/base_tree && timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline) > $O/bench_c2_base_$r.json 2>> $O/bench_c2.err; echo "bench base rc=$?" >> $O/rc.log
done
timeout 600 python bench.py --config c3 --steps 8 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?" >> $O/rc.log
timeout 600 python bench.py --config c3 --steps 8 --warmup 3 --deterministic > $O/bench_c3_det.json 2>> $O/bench_c3.err; echo "c3 det rc=$?" >> $O/rc.log
timeout 600 python bench.py --config c5 --steps 8 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?" >> $O/rc.log
timeout 600 python bench.py --config c5 --steps 8 --warmup 3 --deterministic > $O/bench_c5_det.json 2>> $O/bench_c5.err; echo "c5 det rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 6 $O/pytest_det.log; tail -n 8 $O/pytest_gpu.log; for f in $O/bench_*.json; do echo -n "$f: "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done
