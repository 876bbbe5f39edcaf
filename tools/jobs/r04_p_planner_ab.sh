#!/bin/bash
# round 4, call P: whole-step A/B of the re-fitted GEMM planner against the previous rules (PCM_GEMM_PLAN_LEGACY=1) on ONE box, all four configs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2; do for leg in 1 0; do
  PCM_GEMM_PLAN_LEGACY=$leg timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/c2_legacy${leg}_$r.json 2> $O/c2_legacy${leg}_$r.err; echo "c2 legacy=$leg run $r rc=$?" >> $O/rc.log
done; done
for c in c3 c4 c5; do for leg in 1 0 1 0; do
  PCM_GEMM_PLAN_LEGACY=$leg timeout 300 python bench.py --config $c --steps 8 --warmup 3 >> $O/${c}_legacy${leg}.json 2>> $O/${c}_legacy${leg}.err; echo "$c legacy=$leg rc=$?" >> $O/rc.log
done; done
cat $O/rc.log; for f in $O/c2_legacy*.json $O/c3_*.json $O/c4_*.json $O/c5_*.json; do echo "$f: $(grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f | sed 's/"unit".*"ms_per_step"/ms/' | tr '\n' ';')"; done
