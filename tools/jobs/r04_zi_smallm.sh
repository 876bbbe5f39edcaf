#!/bin/bash
# round 4, call ZI: the batch-row projection kernel (gemm_smallm.hip, M <= 16): GPU tests, per-shape times with the kernel off / on, whole steps of C2 and C5 off / on
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zi; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -k "batch_row or plain_gemm" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/rc.log
PCM_GEMM_SMALLM=0 timeout 300 python tools/gemm_smallm_probe.py > $O/probe_off.txt 2> $O/probe.err; echo "probe off rc=$?" >> $O/rc.log
timeout 300 python tools/gemm_smallm_probe.py > $O/probe_on.txt 2>> $O/probe.err; echo "probe on rc=$?" >> $O/rc.log
for r in 1 2; do
  PCM_GEMM_SMALLM=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_off.json 2>> $O/c2.err; echo "c2 off rc=$?" >> $O/rc.log
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline >> $O/c2_on.json 2>> $O/c2.err; echo "c2 on rc=$?" >> $O/rc.log
done
PCM_GEMM_SMALLM=0 timeout 300 python bench.py --config c5 --steps 8 --warmup 3 >> $O/c5_off.json 2>> $O/c5.err; echo "c5 off rc=$?" >> $O/rc.log
timeout 300 python bench.py --config c5 --steps 8 --warmup 3 >> $O/c5_on.json 2>> $O/c5.err; echo "c5 on rc=$?" >> $O/rc.log
cat $O/rc.log; tail -2 $O/pytest.txt; paste -d'|' $O/probe_off.txt $O/probe_on.txt | cut -c1-150
for f in c2_off c2_on c5_off c5_on; do echo "$f: $(grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/$f.json | sed 's/"unit".*"ms_per_step"/ms/' | tr '\n' ';')"; done
