#!/bin/bash
# round 6, call A: the round-5 tree on this round's box -- libraries load, default-flag bench line (no CPU leg), the two-timestep student
# forward on its own (event time + rocprofv3 per-kernel table of the forward ALONE: where the north star's 38 ms go), frozen 2B pass beside it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06a; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?" >> $O/rc.log
timeout 300 python tools/fwd2t_trace.py --reps 5 --frozen --json $O/fwd2t.json > $O/fwd2t.log 2>&1; echo "fwd2t rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o f -- python $GRAFT_REPO_ROOT/tools/fwd2t_trace.py --reps 5 > $GRAFT_REPO_ROOT/$O/prof_fwd2t.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_a -name "*.db" | head -1) 60 > $O/kernel_stats_fwd2t.txt 2>&1
cat $O/rc.log; cat $O/fwd2t.log | tail -3; cut -c1-400 $O/bench_c2.json; head -45 $O/kernel_stats_fwd2t.txt
