#!/bin/bash
# round 6, call D: after call C -- attention forward restored (the accumulator read behind the loop is half-build only), GroupNorm statistics from the
# epilogue opt-in (PCM_GN_FUSE=1), concat-free skips on: forward A/B on one box, whole GPU suite, bench line new vs round-5 tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
for r in 1 2; do
  (cd tools/probes/base_tree && timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen) 2>&1 | grep -a "ms (eager" | sed "s/^/base_r05   /" >> $O/fwd2t_ab.txt
  timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new        /" >> $O/fwd2t_ab.txt
  PCM_GN_FUSE=1 timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new_GNfuse /" >> $O/fwd2t_ab.txt
  PCM_CAT_FUSE=0 timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new_noCAT  /" >> $O/fwd2t_ab.txt
done
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
for r in 1 2; do
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_new_$r.json 2>> $O/bench_c2.err; echo "bench rc=$?" >> $O/rc.log
  (cd tools/probes/base_tree && timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline) > $O/bench_c2_base_$r.json 2>> $O/bench_c2.err; echo "bench base rc=$?" >> $O/rc.log
done
(cd /tmp && PCM_GN_FUSE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d1 -o f -- python $GRAFT_REPO_ROOT/tools/fwd2t_trace.py --reps 5 > $GRAFT_REPO_ROOT/$O/prof1.log 2>&1)
python tools/prof_summary.py $(find /tmp/prof_d1 -name "*.db" | head -1) 40 > $O/kernel_stats_fwd2t_gnfuse.txt 2>&1
cat $O/rc.log; tail -n 12 $O/pytest_gpu.log; cat $O/fwd2t_ab.txt; for f in $O/bench_c2_*.json; do echo $f; cut -c1-260 $f; done
