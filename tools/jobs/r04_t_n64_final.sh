#!/bin/bash
# round 4, call T: rank-64 K-split kernel as committed (rows per block by M and K) against the library of the commit before the kernel
# (tools/probes/libpcm_base.so = 14a80b0): per shape with the shipped rule and PCM_N64_RF = 1 / 2 / 4 forced, and whole C2 steps on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04t; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_gemm.py -q -x -k "rank64" > $O/pytest_n64.txt 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 200 python tools/n64_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/n64_ab_rule.txt 2> $O/n64_ab.err; echo "ab rule rc=$?" >> $O/rc.log
for rf in 1 2 4; do
  PCM_N64_RF=$rf timeout 200 python tools/n64_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/n64_ab_rf$rf.txt 2>> $O/n64_ab.err; echo "ab rf$rf rc=$?" >> $O/rc.log
done
for r in 1 2; do
  timeout 300 python tools/bench_with_lib.py tools/probes/libpcm_base.so --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_base.json 2>> $O/c2_base.err; echo "c2 base rc=$?" >> $O/rc.log
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_new.json 2>> $O/c2_new.err; echo "c2 new rc=$?" >> $O/rc.log
done
cat $O/rc.log; tail -2 $O/pytest_n64.txt; echo "== rule"; cat $O/n64_ab_rule.txt; for rf in 1 2 4; do echo "== PCM_N64_RF=$rf"; cat $O/n64_ab_rf$rf.txt; done
for f in $O/c2_base.json $O/c2_new.json; do echo "$f: $(grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f | sed 's/"unit".*"ms_per_step"/ms/' | tr '\n' ';')"; done
