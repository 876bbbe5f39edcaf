#!/bin/bash
# round 6, call Z: final tree -- whole GPU suite (durations), smoke, the driver's default bench command (roofline + cpu_baseline legs), per-shape
# GEMM table, rocprofv3 kernel-trace summary of the eager step, the three --pmc passes (stamped table), the other BASELINE configs, the
# reproducible-reductions / half / fp16-teacher lines, the two-timestep student forward on its own
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06z; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; print([capi.Lib(p).build_id for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)])" > $O/libs.log 2>&1 || { cat $O/libs.log; exit 7; }
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
PCM_JOB_OUT=r06z bash tools/jobs/r06_pmc.sh > $O/pmc_job.log 2>&1
PCM_GEMM_TABLE=$O/gemm_shapes.txt timeout 900 python bench.py > $O/bench_c2_default_flags.json 2> $O/bench_c2.err; echo "bench c2 (default flags) rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_c2_12_steps.json 2>> $O/bench_c2.err; echo "bench c2 12 steps rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-prefetch > $O/bench_c2_no_prefetch.json 2>> $O/bench_c2.err; echo "bench c2 no prefetch rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --deterministic > $O/bench_c2_deterministic.json 2>> $O/bench_c2.err; echo "bench c2 deterministic rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --teacher-fp16 > $O/bench_c2_teacher_fp16.json 2>> $O/bench_c2.err; echo "bench c2 teacher fp16 rc=$?" >> $O/rc.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --precision fp16 > $O/bench_c2_fp16.json 2>> $O/bench_c2.err; echo "bench c2 fp16 rc=$?" >> $O/rc.log
for c in c3 c4 c5; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 4 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.log
done
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench2_rehearsal_c2.json 2> $O/bench2_c2.err; echo "2-rank rehearsal c2 rc=$?" >> $O/rc.log
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo PCM_ADV_GRAPH=0 timeout 600 python bench.py --gpus 2 --config c3 --steps 2 --warmup 2 --no-graph > $O/bench2_rehearsal_c3.json 2> $O/bench2_c3.err; echo "2-rank rehearsal c3 rc=$?" >> $O/rc.log
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c5 --steps 2 --warmup 1 > $O/bench2_rehearsal_c5.json 2> $O/bench2_c5.err; echo "2-rank rehearsal c5 rc=$?" >> $O/rc.log
timeout 600 python tools/fwd2t_trace.py > $O/fwd2t.txt 2>&1; echo "fwd2t rc=$?" >> $O/rc.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_z -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
python tools/prof_summary.py $(find /tmp/prof_z -name "*.db" | head -1) 70 > $O/kernel_stats_bench_bs16.txt 2>&1; echo "summary rc=$?" >> $O/rc.log
cp gpurun_out/*.json $O/ 2>/dev/null
cat $O/libs.log $O/rc.log; tail -n 22 $O/pytest_gpu.log; tail -n 5 $O/smoke.log; cut -c1-900 $O/bench_c2_default_flags.json; echo; tail -n 6 $O/fwd2t.txt
for f in c2_12_steps c2_no_prefetch c2_deterministic c2_teacher_fp16 c2_fp16 c3 c4 c5; do grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/bench_$f.json | sed "s/^/$f /"; done
cut -c1-170 $O/pmc_step_table.txt | head -12
for f in c2 c3 c5; do cut -c1-260 $O/bench2_rehearsal_$f.json; echo; tail -n 2 $O/bench2_$f.err; done
