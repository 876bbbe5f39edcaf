#!/bin/bash
# round 6: the trainer programs end to end on the GPU at their real sizes (random weights, synthetic data) -- the draw-one-batch-ahead loops with the
# teacher prefetch (default) against PCM_TEACHER_PREFETCH=0: same loss sequence expected up to atomics-order noise
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06cli; mkdir -p $O; export TMPDIR=/tmp
P=phased-consistency-model_amd
for pf in 1 0; do
  PCM_TEACHER_PREFETCH=$pf timeout 600 python $P/train_pcm_lora_sd15.py --pretrained_teacher_model random --synthetic_data --output_dir /tmp/out_sd15_$pf --max_train_steps 8 --train_batch_size 4 --multiphase 4 > $O/sd15_$pf.log 2>&1; echo "sd15 prefetch=$pf rc=$?" >> $O/rc.log
  PCM_TEACHER_PREFETCH=$pf timeout 600 python $P/train_pcm_lora_sd15_adv.py --pretrained_teacher_model random --synthetic_data --output_dir /tmp/out_adv_$pf --max_train_steps 6 --train_batch_size 2 --multiphase 2 > $O/adv_$pf.log 2>&1; echo "sd15_adv prefetch=$pf rc=$?" >> $O/rc.log
  PCM_TEACHER_PREFETCH=$pf timeout 600 python $P/train_pcm_lora_sd3.py --pretrained_teacher_model random --synthetic_data --lora_rank 32 --num_euler_timesteps 100 --multiphase 2 --train_batch_size 2 --output_dir /tmp/out_sd3_$pf --max_train_steps 4 > $O/sd3_$pf.log 2>&1; echo "sd3 prefetch=$pf rc=$?" >> $O/rc.log
  for d in sd15 adv sd3; do cat /tmp/out_${d}_$pf/logs/*.jsonl > $O/${d}_$pf.jsonl 2>/dev/null; done
done
cat $O/rc.log; for d in sd15 adv sd3; do for pf in 1 0; do echo "== $d prefetch=$pf"; cut -c1-200 $O/${d}_$pf.jsonl | head -8; done; done; tail -n 3 $O/sd3_1.log
