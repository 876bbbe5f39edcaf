#!/bin/bash
# round 4, call R: rank-64 K-split kernel, second pass (16 waves for long K at 16 rows per block) against the library of the commit before
# the change (tools/probes/libpcm_base.so), per shape and as whole steps of C2 / C4 / C5 on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04r; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_gemm.py -q -x -k "rank64" > $O/pytest_n64.txt 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 600 python tools/n64_ab_libs.py tools/probes/libpcm_base.so phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so > $O/n64_ab.txt 2> $O/n64_ab.err; echo "ab rc=$?" >> $O/rc.log
AB_ONLY="(8192, 1" timeout 300 python tools/gemm_small_m.py > $O/sweep_4w.txt 2> $O/sweep_4w.err; echo "sweep rc=$?" >> $O/rc.log
for r in 1 2; do
  timeout 300 python tools/bench_with_lib.py tools/probes/libpcm_base.so --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_base.json 2>> $O/c2_base.err; echo "c2 base rc=$?" >> $O/rc.log
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_new.json 2>> $O/c2_new.err; echo "c2 new rc=$?" >> $O/rc.log
done
for c in c4 c5 c3; do for r in 1 2; do
  timeout 300 python tools/bench_with_lib.py tools/probes/libpcm_base.so --config $c --steps 8 --warmup 3 >> $O/${c}_base.json 2>> $O/${c}_base.err; echo "$c base rc=$?" >> $O/rc.log
  timeout 300 python bench.py --config $c --steps 8 --warmup 3 >> $O/${c}_new.json 2>> $O/${c}_new.err; echo "$c new rc=$?" >> $O/rc.log
done; done
cat $O/rc.log; tail -2 $O/pytest_n64.txt; cat $O/n64_ab.txt; cut -c1-200 $O/sweep_4w.txt
for f in $O/c?_base.json $O/c?_new.json; do echo "$f: $(grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f | sed 's/"unit".*"ms_per_step"/ms/' | tr '\n' ';')"; done
