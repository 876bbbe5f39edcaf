#!/bin/bash
# round 6, call B: GroupNorm statistics from the producing contraction's epilogue (pcm_gemm_epi.chstats) + concat-free skips (out2): whole GPU
# suite, then the two-timestep student forward A/B on one box -- round-5 tree | new tree | new tree with each fusion switched off -- and the bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
for r in 1 2; do
  (cd tools/probes/base_tree && timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen) 2>&1 | grep -a "ms (eager" | sed "s/^/base_r05   /" >> $O/fwd2t_ab.txt
  timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new        /" >> $O/fwd2t_ab.txt
  PCM_GN_FUSE=0 timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new_noGN   /" >> $O/fwd2t_ab.txt
  PCM_CAT_FUSE=0 timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new_noCAT  /" >> $O/fwd2t_ab.txt
done
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?" >> $O/rc.log
(cd tools/probes/base_tree && timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline) > $O/bench_c2_base.json 2>> $O/bench_c2.err; echo "bench base rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 15 $O/pytest_gpu.log; cat $O/fwd2t_ab.txt; cut -c1-300 $O/bench_c2.json; cut -c1-300 $O/bench_c2_base.json
