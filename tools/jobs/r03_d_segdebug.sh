#!/bin/bash
# round 3, call D: (1) where the time of the SEGMENTED adversarial graph replay goes (PCM_SEG_TIMING: host time per graph launch / host
# action): one rank with forced segmentation, then two ranks on device 0 over gloo; (2) SQ counters of the two forward-attention kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03d; mkdir -p $O; export TMPDIR=/tmp
PCM_SEG_FORCE=1 PCM_SEG_TIMING=1 timeout 400 python bench.py --config c3 --batch 4 --steps 4 --warmup 2 > $O/seg_1rank.json 2> $O/seg_1rank.err; echo "seg 1 rank rc=$?" >> $O/rc.log
PCM_SEG_TIMING=1 PCM_FORCE_DEVICE=0 PCM_DIST_BACKEND=gloo timeout 300 python bench.py --config c3 --gpus 2 --batch 4 --steps 2 --warmup 2 > $O/seg_2rank.json 2> $O/seg_2rank.err; echo "seg 2 rank rc=$?" >> $O/rc.log
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_a -o a -- python $GRAFT_REPO_ROOT/tools/attn_fwd_pmc.py > $GRAFT_REPO_ROOT/$O/pmc_attn.log 2>&1); echo "pmc rc=$?" >> $O/rc.log
python tools/pmc_table.py $(find /tmp/pmc_a -name "*.db" | head -1) 12 > $O/pmc_attn_table.txt 2>&1
cat $O/rc.log; grep "seg replay" $O/seg_1rank.err | head -8 | cut -c1-400; cut -c1-300 $O/seg_1rank.json; grep "seg replay" $O/seg_2rank.err | head -8 | cut -c1-500; cut -c1-300 $O/seg_2rank.json; cat $O/pmc_attn_table.txt | cut -c1-200
