#!/bin/bash
# round 4, call W: the half build's GPU tests on the final form, and the C2 bench line in both formats on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04w; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fp16.py -q -x --durations=8 > $O/pytest_fp16.txt 2>&1; echo "fp16 rc=$?" >> $O/rc.log
for r in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_bf16.json 2>> $O/c2_bf16.err; echo "c2 bf16 rc=$?" >> $O/rc.log
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --precision fp16 >> $O/c2_fp16.json 2>> $O/c2_fp16.err; echo "c2 fp16 rc=$?" >> $O/rc.log
done
cat $O/rc.log; tail -14 $O/pytest_fp16.txt
for f in $O/c2_bf16.json $O/c2_fp16.json; do echo "$f: $(grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f | sed 's/"unit".*"ms_per_step"/ms/' | tr '\n' ';') $(grep -o '"dtype": "[a-z0-9]*"' $f | tr '\n' ' ') $(grep -o '"loss_last": [0-9.e-]*' $f | tr '\n' ' ')"; done
tail -3 $O/c2_fp16.err
