#!/bin/bash
# round 4, call H: piece loops instantiated per flag set (manual unswitching) vs the previous commit's old epilogue; + SD3 / SDXL shapes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04h; mkdir -p $O; export TMPDIR=/tmp
L=phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so
PCM_GEMM_BIG=4 timeout 600 python tools/gemm_ab_libs.py tools/probes/libpcm_base.so $L > $O/ab_libs_8p.txt 2>&1; echo "ab libs rc=$?" >> $O/rc.log
for i in 1 2; do timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?" >> $O/rc.log; done
timeout 600 python bench.py --config c5 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?" >> $O/rc.log
timeout 900 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_gemm.py -q -x --durations=5 -k "not c2_as" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/ab_libs_8p.txt; for i in 1 2 c5; do cut -c1-200 $O/bench_$i.json; echo; done; tail -12 $O/pytest.txt
