#!/bin/bash
# round 4, call C: the rewritten tile epilogue (bias row in LDS, residual / row-vector pieces requested a pass ahead) against the previous
# build (tools/probes/libpcm_base.so = csrc of the previous commit), gemm8p and gemm4w; ablations + stamps again
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04c; mkdir -p $O; export TMPDIR=/tmp
L=phased-consistency-model_amd/pcm_amd/lib/libpcm_hip.so
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q > $O/pytest_gemm.txt 2>&1; echo "pytest gemm rc=$?" >> $O/rc.log
PCM_GEMM_BIG=4 timeout 600 python tools/gemm_ab_libs.py tools/probes/libpcm_base.so $L > $O/ab_libs_8p.txt 2>&1; echo "ab libs 8p rc=$?" >> $O/rc.log
PCM_GEMM_BIG=3 timeout 600 python tools/gemm_ab_libs.py tools/probes/libpcm_base.so $L > $O/ab_libs_4w.txt 2>&1; echo "ab libs 4w rc=$?" >> $O/rc.log
timeout 900 python tools/gemm_4w_ab.py > $O/gemm4w_ab.txt 2> $O/gemm4w_ab.err; echo "ab rc=$?" >> $O/rc.log
timeout 900 python tools/gemm4w_ablate.py > $O/gemm4w_ablate.txt 2> $O/gemm4w_ablate.err; echo "ablate rc=$?" >> $O/rc.log
for i in 1 2; do
  PCM_GEMM_BIG=4 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_no4w_$i.json 2> $O/bench_no4w_$i.err; echo "no4w $i rc=$?" >> $O/rc.log
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_4w_$i.json 2> $O/bench_4w_$i.err; echo "4w $i rc=$?" >> $O/rc.log
done
cat $O/rc.log; tail -3 $O/pytest_gemm.txt; cat $O/ab_libs_8p.txt $O/ab_libs_4w.txt; for f in no4w_1 4w_1 no4w_2 4w_2; do echo "$f: $(cat $O/bench_$f.json | cut -c1-160)"; done
