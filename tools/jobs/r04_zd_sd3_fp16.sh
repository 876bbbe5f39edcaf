#!/bin/bash
# round 4, call ZD: the SD3 / MMDiT trainers through the half build (narrow configs, live oracle) + the bf16 MMDiT tests beside them
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zd; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fp16.py -q -s -k "sd3" > $O/pytest_fp16_sd3.txt 2>&1; echo "fp16 sd3 rc=$?" >> $O/rc.log
timeout 600 python -m pytest tests/test_gpu_mmdit.py -q -s > $O/pytest_bf16_sd3.txt 2>&1; echo "bf16 sd3 rc=$?" >> $O/rc.log
cat $O/rc.log; grep -h "loss\|grad" $O/pytest_fp16_sd3.txt | cut -c1-200; echo ---; grep -h "loss\|grad" $O/pytest_bf16_sd3.txt | cut -c1-200; tail -2 $O/pytest_fp16_sd3.txt
