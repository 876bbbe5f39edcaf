#!/bin/bash
# round 4, call M: C3 after the dense head wgrad + the row-batched colsum / gn_param_grad / rowdot_bwd loops: kernel tests of the touched
# kernels, bench line + per-shape GEMM table of a D + G pair (where the small-M launches of the bs-8 step go), same for C4 / C5
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04m; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_adv.py -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes_c3.txt timeout 600 python bench.py --config c3 --steps 12 --warmup 4 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes_c4.txt timeout 600 python bench.py --config c4 --steps 8 --warmup 3 > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 rc=$?" >> $O/rc.log
PCM_GEMM_TABLE=$O/gemm_shapes_c5.txt timeout 600 python bench.py --config c5 --steps 8 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 4 $O/pytest.txt; for c in c3 c4 c5; do cut -c1-250 $O/bench_$c.json; done; head -45 $O/gemm_shapes_c3.txt
