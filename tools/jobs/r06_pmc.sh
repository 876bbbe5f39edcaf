#!/bin/bash
# round 6: the north star's own evidence on THIS tree -- MFMA utilisation and HBM GB/s per kernel of one eager bs-16 step, from three separate
# rocprofv3 --pmc passes (SQ counters; FETCH_SIZE; WRITE_SIZE -- each with --kernel-trace only), merged by tools/pmc_step_table.py, which stamps the
# table with the kernel sources' id (bench.py marks a table whose stamp is not its library's as STALE)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${PCM_JOB_OUT:-r06pmc}; mkdir -p $O; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-prefetch --no-cpu-baseline --no-roofline"
(cd /tmp && timeout 420 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_s -o s -- $CMD > $GRAFT_REPO_ROOT/$O/pmc_s.log 2>&1); echo "pmc S rc=$?" >> $O/rc.log
(cd /tmp && timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f -- $CMD > $GRAFT_REPO_ROOT/$O/pmc_f.log 2>&1); echo "pmc F rc=$?" >> $O/rc.log
(cd /tmp && timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w -- $CMD > $GRAFT_REPO_ROOT/$O/pmc_w.log 2>&1); echo "pmc W rc=$?" >> $O/rc.log
python tools/pmc_step_table.py $(find /tmp/pmc_s -name "*.db" | head -1) $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) 40 > $O/pmc_step_table.txt 2>&1; echo "table rc=$?" >> $O/rc.log
cat $O/rc.log; cut -c1-170 $O/pmc_step_table.txt | head -30; tail -n 2 $O/pmc_s.log
