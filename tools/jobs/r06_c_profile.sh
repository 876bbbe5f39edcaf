#!/bin/bash
# round 6, call C: where call B's +3 ms on the two-timestep forward come from -- per-kernel tables of the forward alone: new tree with both fusions,
# with both switched off (the kernels' own regression against round 5, if any), and the timing A/B again after the compile-time "extras" flag
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -k "epilogue_fusions or gemm_big" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
for r in 1 2; do
  (cd tools/probes/base_tree && timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen) 2>&1 | grep -a "ms (eager" | sed "s/^/base_r05   /" >> $O/fwd2t_ab.txt
  timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new        /" >> $O/fwd2t_ab.txt
  PCM_GN_FUSE=0 timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new_noGN   /" >> $O/fwd2t_ab.txt
  PCM_GN_FUSE=0 PCM_CAT_FUSE=0 timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen 2>&1 | grep -a "ms (eager" | sed "s/^/new_none   /" >> $O/fwd2t_ab.txt
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c1 -o f -- python $GRAFT_REPO_ROOT/tools/fwd2t_trace.py --reps 5 > $GRAFT_REPO_ROOT/$O/prof1.log 2>&1)
python tools/prof_summary.py $(find /tmp/prof_c1 -name "*.db" | head -1) 40 > $O/kernel_stats_fwd2t_new.txt 2>&1
(cd /tmp && PCM_GN_FUSE=0 PCM_CAT_FUSE=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o f -- python $GRAFT_REPO_ROOT/tools/fwd2t_trace.py --reps 5 > $GRAFT_REPO_ROOT/$O/prof2.log 2>&1)
python tools/prof_summary.py $(find /tmp/prof_c2 -name "*.db" | head -1) 40 > $O/kernel_stats_fwd2t_new_nofuse.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o f -- python $GRAFT_REPO_ROOT/tools/probes/base_tree/tools/fwd2t_trace.py --reps 5 > $GRAFT_REPO_ROOT/$O/prof3.log 2>&1)
python tools/prof_summary.py $(find /tmp/prof_c3 -name "*.db" | head -1) 40 > $O/kernel_stats_fwd2t_base.txt 2>&1
cat $O/rc.log; tail -3 $O/pytest_gemm.log; cat $O/fwd2t_ab.txt
