#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/s; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 420 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_s -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1); echo "pmc rc=$?" >> $O/rc.log
python tools/pmc_table.py $(find /tmp/pmc_s -name "*.db" | head -1) 36 > $O/pmc_table.txt 2>&1; echo "table rc=$?" >> $O/rc.log
cat $O/rc.log; head -30 $O/pmc_table.txt | cut -c1-190; tail -3 $O/pmc.log
