#!/bin/bash
# round 3, call K: A/B of the by-shape conv K order on one box: bench with PCM_GEMM_CONV_CO=0 (tap-outer everywhere, the round-2 behaviour) vs the
# default (chunk-outer on 8x8 maps), interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03k; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  PCM_GEMM_CONV_CO=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_tapouter_$i.json 2> $O/bench_tapouter_$i.err; echo "tap-outer $i rc=$?" >> $O/rc.log
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_auto_$i.json 2> $O/bench_auto_$i.err; echo "auto $i rc=$?" >> $O/rc.log
done
cat $O/rc.log; for f in tapouter_1 auto_1 tapouter_2 auto_2; do echo "$f: $(grep timed $O/bench_$f.err)"; done
