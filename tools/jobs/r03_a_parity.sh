#!/bin/bash
# round 3, call A: the new / sharpened parity tests on hardware (attention rel-L2 bounds, rounding-point-matched oracle, 36-head adversarial
# step with real learning rates, one-sample full-size SDXL / SD3 oracle parity), a baseline bench line, and the self-spawn 2-rank rehearsal
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03a; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest -m gpu -x -q -s tests/test_gpu_kernels.py -k "attention or discriminator" > $O/pytest_attn.log 2>&1; echo "attn rc=$?" >> $O/rc.log
timeout 1500 python -m pytest -m gpu -x -q -s tests/test_gpu_rounding_matched.py > $O/pytest_rm.log 2>&1; echo "rounding_matched rc=$?" >> $O/rc.log
timeout 1500 python -m pytest -m gpu -x -q -s tests/test_gpu_adv.py -k c3 > $O/pytest_adv.log 2>&1; echo "adv_c3 rc=$?" >> $O/rc.log
timeout 1200 python -m pytest -m gpu -x -q -s tests/test_gpu_zy_sdxl_fullsize.py tests/test_gpu_zz_sd3_fullsize.py > $O/pytest_full.log 2>&1; echo "fullsize rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
# `python bench.py --gpus 2` with no torchrun: bench.py spawns its own two ranks (both on device 0 here, gloo in place of RCCL)
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/pytest_attn.log $O/pytest_rm.log $O/pytest_adv.log $O/pytest_full.log; tail -n 3 $O/bench.err; cut -c1-400 $O/bench.json; tail -n 4 $O/bench2.err; cut -c1-700 $O/bench2.json
