#!/bin/bash
# round 5, call G: dK/dV kernel of the pre-scaled-query attention at THREE waves per SIMD with the register-lean schedule (no second transpose-read
# fragment set in flight; optionally the score / softmax phase one query half at a time), against the shipped two waves per SIMD.
# The variant libraries tools/probes/libpcm_dkdvlb3{,h}.so were built in the container from csrc/ WITH tools/probes/patches/r05_dkdv_lean_schedule_3_waves.patch
# applied (python tools/probes/build_variant.py dkdvlb3 -DPCM_ATTN_DKDV_WAVES=3 [-DPCM_ATTN_DKDV_HALVES=1]); the shipped source does not carry that schedule
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O; export TMPDIR=/tmp
for v in dkdvlb3 dkdvlb3h; do
  timeout 300 python tools/attn_ps_ab.py 3 tools/probes/libpcm_$v.so > $O/attn_ps_ab_$v.txt 2>&1; echo "$v rc=$?" >> $O/rc.log
done
cat $O/rc.log; for v in dkdvlb3 dkdvlb3h; do echo == $v; grep -o "^B=.\{38\}\|bwd.*register staging[^)]*)\|alt lib.*" $O/attn_ps_ab_$v.txt | paste - - - | head -8; done
