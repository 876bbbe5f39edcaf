#!/bin/bash
# round 6, call F: is test_fp16_adv_steps_graph_replay_equals_eager flaky (atomics) or broken; two-stream overlap probe of the step's independent passes; rest of the suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2 3 4; do timeout 300 python -m pytest tests/test_gpu_fp16.py -q -x -k "adv_steps_graph_replay" 2>&1 | tail -n 3 >> $O/flaky.log; done
for r in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_adv.py -q -x -k "graph_replay" 2>&1 | tail -n 2 >> $O/flaky_bf16.log; done
timeout 600 python tools/two_stream_probe.py --reps 5 > $O/two_stream.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=5 --deselect tests/test_gpu_fp16.py::test_fp16_adv_steps_graph_replay_equals_eager > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/flaky.log; cat $O/flaky_bf16.log; grep -a "ms (eager" $O/two_stream.txt; tail -n 6 $O/pytest_gpu.log
