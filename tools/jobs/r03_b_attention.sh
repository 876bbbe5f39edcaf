#!/bin/bash
# round 3, call B: the software-pipelined attention forward on hardware: parity tests, A/B of the four forward variants of the same build
# (tools/attn_fwd_variants.py), the G step of the 36-head adversarial parity case, a bench line with the new forward
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03b; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_kernels.py -k "attention" > $O/pytest_attn.log 2>&1; echo "attn rc=$?" >> $O/rc.log
timeout 900 python tools/attn_fwd_variants.py 0,1,2,3 > $O/attn_variants.txt 2>&1; echo "variants rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
timeout 900 python -m pytest -m gpu -x -q -s "tests/test_gpu_adv.py::test_adv_step_c3_shape_full_size[1]" > $O/pytest_adv.log 2>&1; echo "adv_c3_g rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 3 $O/pytest_attn.log; cat $O/attn_variants.txt; grep -i "timed\|two-timestep" $O/bench.err; tail -n 3 $O/pytest_adv.log
