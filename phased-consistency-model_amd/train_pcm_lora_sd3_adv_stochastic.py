#!/usr/bin/env python
"""train_pcm_lora_sd3_adv_stochastic.py — the stochastic-sampler recipe of the SD3 adversarial trainer
(code/text_to_image_sd3/train_pcm_lora_sd3_adv_stochastic.py; run.sh's first recipe: ``--num_euler_timesteps=100 --multiphase=1``).
Its training step is the deterministic script's (the two files differ in the LoRA target list -- no ``pos_embed.proj`` here, :1008 --
the scheduler construction and the validation sampler, PCMFMStochasticScheduler: ``sample_pcm_lora_sd3.py --stochastic``)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_pcm_lora_sd3_adv as adv  # noqa: E402

adv.STOCHASTIC = True
parse_args, main = adv.parse_args, adv.main

if __name__ == "__main__":
    main(parse_args())
