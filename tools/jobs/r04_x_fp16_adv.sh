#!/bin/bash
# round 4, call X: the adversarial step (BASELINE configs[2] shape, 36 heads) through the half build against the fp32-oracle fixture; bf16 adv tests beside it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04x; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fp16.py -q -x -s -k "adv" > $O/pytest_fp16_adv.txt 2>&1; echo "fp16 adv rc=$?" >> $O/rc.log
cp gpurun_out/fp16_adv_*.json $O/ 2>/dev/null
timeout 1200 python -m pytest tests/test_gpu_adv.py -q -x > $O/pytest_bf16_adv.txt 2>&1; echo "bf16 adv rc=$?" >> $O/rc.log
cat $O/rc.log; tail -30 $O/pytest_fp16_adv.txt | cut -c1-1500; tail -3 $O/pytest_bf16_adv.txt
