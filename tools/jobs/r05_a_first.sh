#!/bin/bash
# round 5, call A: (1) the pre-scaled-query attention kernels against the ones they replace (kernel-level A/B + accuracy), (2) the new GPU
# tests of this round: reproducible reductions, bitwise-reproducible step, deterministic loss curve (hard 1e-3 bound), full-size SDXL / SD3
# whole-step parity, prescaled attention cases, (3) a baseline bench line of the tree for this box.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05a; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/attn_ps_ab.py 3 > $O/attn_ps_ab.txt 2>&1; echo "attn ab rc=$?" >> $O/rc.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "reproducible or prescaled or attention" --durations=5 > $O/pytest_kernels.txt 2>&1; echo "kernels rc=$?" >> $O/rc.log
timeout 600 python -m pytest tests/test_gpu_step.py -q -k "deterministic" > $O/pytest_det_step.txt 2>&1; echo "det step rc=$?" >> $O/rc.log
timeout 900 python -m pytest tests/test_gpu_bench_config.py -q -k "loss_curve" > $O/pytest_curve.txt 2>&1; echo "curve rc=$?" >> $O/rc.log
timeout 900 python -m pytest tests/test_gpu_zy_sdxl_fullsize.py tests/test_gpu_zz_sd3_fullsize.py -q -s > $O/pytest_fullsize.txt 2>&1; echo "fullsize rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_c2.txt 2>&1; echo "bench rc=$?" >> $O/rc.log
cp gpurun_out/*.json $O/ 2>/dev/null
cat $O/rc.log; cat $O/attn_ps_ab.txt; tail -5 $O/pytest_kernels.txt $O/pytest_det_step.txt $O/pytest_curve.txt; tail -15 $O/pytest_fullsize.txt; tail -3 $O/bench_c2.txt
