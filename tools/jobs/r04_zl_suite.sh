#!/bin/bash
# round 4, call ZL: the whole GPU suite on the end-of-round tree (after the loss-curve bound fix), smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zl; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 14 $O/pytest_gpu.log; tail -n 3 $O/smoke.log; python -c "import json;d=json.load(open('gpurun_out/loss_curve_20_real_size.json'));print('curve', d['mean_abs_hip_vs_matched'], d['matched_floor_mean_of_3'])"
