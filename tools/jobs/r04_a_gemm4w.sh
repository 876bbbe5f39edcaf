#!/bin/bash
# round 4, call A: the short-K kernel (gemm4w.hip) on hardware -- kernel tests, per-shape A/B against gemm8p, whole-step A/B (planner with /
# without gemm4w), and the first fixture-backed parity tests (oracle side from tests/golden/step_*.safetensors)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04a; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "short_k" > $O/pytest_gemm4w.txt 2>&1; echo "pytest gemm4w rc=$?" >> $O/rc.log
timeout 900 python tools/gemm_4w_ab.py > $O/gemm4w_ab.txt 2> $O/gemm4w_ab.err; echo "ab rc=$?" >> $O/rc.log
AB_SET=lin2b timeout 300 python tools/gemm_big_ab.py > $O/small_vs_big.txt 2>&1; echo "small-vs-big rc=$?" >> $O/rc.log
for i in 1 2; do
  PCM_GEMM_BIG=4 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_no4w_$i.json 2> $O/bench_no4w_$i.err; echo "no4w $i rc=$?" >> $O/rc.log
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_4w_$i.json 2> $O/bench_4w_$i.err; echo "4w $i rc=$?" >> $O/rc.log
done
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_rounding_matched.py -x -q --durations=8 > $O/pytest_fixtures.txt 2>&1; echo "pytest fixtures rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest_gemm4w.txt; tail -4 $O/gemm4w_ab.txt; for f in no4w_1 4w_1 no4w_2 4w_2; do echo "$f: $(cat $O/bench_$f.json | cut -c1-200)"; done; tail -12 $O/pytest_fixtures.txt
