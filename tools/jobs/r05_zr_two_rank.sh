#!/bin/bash
# round 5, call ZR: rehearsal of the driver's N = 2 launch line on ONE GPU (both ranks on device 0, gloo in place of RCCL) on the final tree, in both
# formats: split-graph capture + bucketed exchange; the half build sums loss-scaled gradients.  A launch rehearsal, not a scaling number.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05zr; mkdir -p $O; export TMPDIR=/tmp
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench2_bf16.json 2> $O/bench2_bf16.err; echo "torchrun bf16 rc=$?" >> $O/rc.log
PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --precision fp16 > $O/bench2_fp16.json 2> $O/bench2_fp16.err; echo "self-spawn fp16 rc=$?" >> $O/rc.log
cat $O/rc.log; cut -c1-700 $O/bench2_bf16.json; echo; cut -c1-400 $O/bench2_fp16.json; tail -3 $O/bench2_fp16.err
