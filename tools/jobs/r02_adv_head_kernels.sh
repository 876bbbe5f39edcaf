#!/bin/bash
# discriminator-head kernels after the block-level reductions / tiled weight repack: kernel tests, graph-vs-eager adversarial test, 36-head probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "pack or adv_kernels or discriminator" > $O/test_k.log 2>&1; echo "kernels rc=$?" >> $O/rc.log
timeout 600 python -m pytest tests/test_gpu_adv.py -x -q -k "graph_replay" > $O/test_adv.log 2>&1; echo "adv graph rc=$?" >> $O/rc.log
timeout 500 python tools/adv_step_probe.py 8 > $O/adv_probe.txt 2>&1; echo "adv probe rc=$?" >> $O/rc.log
cat $O/rc.log; tail -n 2 $O/test_k.log; tail -n 2 $O/test_adv.log; tail -n 8 $O/adv_probe.txt
