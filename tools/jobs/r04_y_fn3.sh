#!/bin/bash
# round 4, call Y (needs tools/probes/patches/r04_gemm8p_256x192_tile.patch applied and PCM_GEMM_FN3 read by the planner): the 256 x 192 member of the phased-tile family (gemm8p <F0 = 1>): kernel tests, per-shape sweep of forced plans on the
# shapes the time model moves to it, whole C2 steps with PCM_GEMM_FN3 = 0 / 1 on one box; the half build's tests after the assert fix
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04y; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -x > $O/pytest_gemm.txt 2>&1; echo "gemm rc=$?" >> $O/rc.log
timeout 900 python -m pytest tests/test_gpu_fp16.py -q -x -k "kernels or gemm or attention" > $O/pytest_fp16.txt 2>&1; echo "fp16 rc=$?" >> $O/rc.log
AB_SHAPES="8192,1280,1280;16384,640,640;8192,1280,5120;16384,640,5120;16384,640,1920;8192,1280,2560;8192,1280,640;131072,192,320;65536,192,960;4096,2560,1280;8192,1280,1920;2048,1280,5120;4096,1280,3840" timeout 600 python tools/gemm_small_m.py > $O/sweep.txt 2> $O/sweep.err; echo "sweep rc=$?" >> $O/rc.log
for r in 1 2; do
  PCM_GEMM_FN3=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_fn3_off.json 2>> $O/c2_off.err; echo "c2 off rc=$?" >> $O/rc.log
  PCM_GEMM_FN3=1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_fn3_on.json 2>> $O/c2_on.err; echo "c2 on rc=$?" >> $O/rc.log
done
cat $O/rc.log; tail -3 $O/pytest_gemm.txt; tail -3 $O/pytest_fp16.txt; cut -c1-260 $O/sweep.txt
for f in $O/c2_fn3_off.json $O/c2_fn3_on.json; do echo "$f: $(grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f | sed 's/"unit".*"ms_per_step"/ms/' | tr '\n' ';')"; done
