#!/bin/bash
# round 5, call H: (1) the fp16-teacher / bf16-student split at the real SD1.5 size (tests/test_gpu_step.py::test_fp16_teacher_bf16_student_split);
# (2) its cost on the whole C2 step (bench.py --teacher-fp16 against the default, interleaved twice); (3) the whole GPU suite + smoke on the
# tree whose kernel files no longer test PCM_HOST_EMU (macro isolation) -- the hardware objects must be unchanged in behaviour
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, \"phased-consistency-model_amd\"); from pcm_amd import capi; [capi.Lib(p) for p in (capi.DEFAULT_LIB, capi.F16_LIB, capi.TOOLS_LIB, capi.TOOLS_F16_LIB)]; print(\"libs load\")" || exit 7
timeout 600 python -m pytest tests/test_gpu_step.py -q -x -s -k "fp16_teacher" > $O/pytest_teacher_split.log 2>&1; echo "split rc=$?" >> $O/rc.log
for i in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_teacher_same_$i.txt
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --teacher-fp16 2>/dev/null | tail -n 1 | cut -c1-200 > $O/step_teacher_fp16_$i.txt
done
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
cp gpurun_out/*.json $O/ 2>/dev/null
cat $O/rc.log; tail -n 8 $O/pytest_teacher_split.log; for f in $O/step_teacher_*; do echo $f; cat $f; done; tail -n 22 $O/pytest_gpu.log; tail -n 3 $O/smoke.log
