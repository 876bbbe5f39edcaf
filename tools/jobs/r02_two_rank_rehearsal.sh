#!/bin/bash
# rehearsal of the N = 2 launch line of the driver on ONE GPU (both ranks on device 0, gloo instead of RCCL): split-graph capture,
# bucketed all-reduces between the replays, barrier + max-over-ranks timing, the JSON line of rank 0
cd $GRAFT_REPO_ROOT; O=gpurun_out/v; mkdir -p $O
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 12 $O/bench2.err; cat $O/bench2.json | cut -c1-600
