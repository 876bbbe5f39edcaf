#!/bin/bash
# round 4, call O: the re-fitted GEMM planner (short-K small-M rule + time model for the phased tile's K split) on hardware: sweep with the
# planner arm, GEMM tests, bench lines of all four configs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04o; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/gemm_small_m.py > $O/small_m_sweep3.txt 2> $O/small_m_sweep.err; echo "sweep rc=$?" >> $O/rc.log
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > $O/pytest_gemm.txt 2>&1; echo "pytest gemm rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes_c2.txt timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?" >> $O/rc.log
for c in c3 c4 c5; do
  PCM_GEMM_TABLE=$O/gemm_shapes_$c.txt timeout 600 python bench.py --config $c --steps 8 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?" >> $O/rc.log
done
cat $O/rc.log; tail -n 3 $O/pytest_gemm.txt; cut -c1-150 $O/small_m_sweep3.txt; for c in c2 c3 c4 c5; do cut -c1-220 $O/bench_$c.json; done
