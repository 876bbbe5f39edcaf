#!/bin/bash
# round 4, call ZE: SDXL and SD3-medium at their real sizes through the half build (one sample vs the fp32-oracle fixture + a loss-scaled step)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04ze; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fp16.py -q -s -k "full_size_one_sample" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/rc.log
cp gpurun_out/fp16_s*_fullsize_oracle_parity.json $O/ 2>/dev/null
cat $O/rc.log; grep -h "^fp16\|full-size" $O/pytest.txt | cut -c1-300; tail -3 $O/pytest.txt
