#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/q; mkdir -p $O
PCM_GEMM_TABLE=$O/gemm_table.txt timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2> $O/bench.err > $O/bench.json; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; grep -i "timed" $O/bench.err
