#!/bin/bash
# round 4, call S: rows per block of the rank-64 K-split kernel (PCM_N64_RF = 1 / 2 / 4 forced, and the shipped rule) against the base library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_gemm.py -q -x -k "rank64" > $O/pytest_n64.txt 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 300 python tools/n64_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/n64_ab_rule.txt 2> $O/n64_ab.err; echo "ab rule rc=$?" >> $O/rc.log
for rf in 1 2 4; do
  PCM_N64_RF=$rf timeout 300 python tools/n64_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/n64_ab_rf$rf.txt 2>> $O/n64_ab.err; echo "ab rf$rf rc=$?" >> $O/rc.log
done
cat $O/rc.log; tail -2 $O/pytest_n64.txt; echo "== rule"; cat $O/n64_ab_rule.txt; for rf in 1 2 4; do echo "== PCM_N64_RF=$rf"; cat $O/n64_ab_rf$rf.txt; done
