#!/bin/bash
# round 6, call K: rehearsal of the driver's N = 2 launch line on ONE GPU (both ranks on device 0, gloo in place of RCCL) with the pipelined capture:
# split-graph capture (teacher branch joined before the cut) + bucketed exchange + the per-bucket log in the line; also c3 / c4 / c5 lines of the final tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06k; mkdir -p $O; export TMPDIR=/tmp
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench2_bf16.json 2> $O/bench2_bf16.err; echo "torchrun bf16 rc=$?" >> $O/rc.log
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-prefetch > $O/bench2_noprefetch.json 2> $O/bench2_noprefetch.err; echo "self-spawn no-prefetch rc=$?" >> $O/rc.log
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo PCM_ADV_GRAPH=0 PCM_HEAD_GRAD_EXCHANGE=bf16 timeout 600 python bench.py --gpus 2 --config c3 --steps 2 --warmup 2 --no-graph > $O/bench2_c3_bf16_heads.json 2> $O/bench2_c3.err; echo "c3 2-rank bf16 head exchange rc=$?" >> $O/rc.log
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --steps 8 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.log; done
timeout 600 python bench.py --config c4 --steps 8 --warmup 3 --no-prefetch > $O/bench_c4_noprefetch.json 2>> $O/bench_c4.err; echo "c4 nopf rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
cat $O/rc.log; cut -c1-1500 $O/bench2_bf16.json; echo; tail -n 3 $O/bench2_bf16.err; cut -c1-300 $O/bench2_noprefetch.json; echo; cut -c1-400 $O/bench2_c3_bf16_heads.json; echo; tail -n 3 $O/bench2_c3.err; for f in $O/bench_c*.json; do echo -n "$f: "; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $f; done; tail -n 4 $O/smoke.log
