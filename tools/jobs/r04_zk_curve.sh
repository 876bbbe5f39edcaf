#!/bin/bash
# round 4, call ZK: the bf16 20-step loss curve test with the batch-row kernel off / on, twice each (run-to-run variation from the gradient atomics vs the kernel's effect)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zk; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2; do for s in 0 1; do
  PCM_GEMM_SMALLM=$s timeout 300 python -m pytest tests/test_gpu_bench_config.py -q -k "loss_curve_20_steps_real_size" > $O/curve_s${s}_r$r.txt 2>&1; echo "smallm=$s run $r rc=$?" >> $O/rc.log
  cp gpurun_out/loss_curve_20_real_size.json $O/curve_s${s}_r$r.json
done; done
cat $O/rc.log; for f in $O/curve_s*_r*.json; do echo "$f $(python -c "import json;d=json.load(open('$f'));print(d['mean_abs_hip_vs_matched'], d['mean_abs_hip_vs_fp32'], d['last5_hip_vs_matched'])")"; done
