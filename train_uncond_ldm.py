#!/usr/bin/env python3
"""MI355X counterpart of the reference's unconditional LATENT trainer (/root/reference/train_uncond_ldm.py).

Same command line and YAML schema (``model.first_stage`` = the frozen KL autoencoder, ``model.class_name:
ddm.ddm_const_2.LatentDiffusion``).  The reference's latent driver differs from its pixel-space one only in
  * building the first stage and handing it to the wrapper as ``auto_encoder``     [train_uncond_ldm.py:42-59]
  * ``on_train_batch_start(batch)`` on the very first micro-batch (std-rescaling)  [:234-238]
  * restoring ``scale_factor`` from the checkpoint                                 [:206-207]
  * sampling 16 images for the periodic grid                                       [:300-309]
all of which the shared Trainer in train_uncond_dpm.py does whenever the model exposes those hooks, so this file is
the entry point only.  Launch like train_uncond_dpm.py.
"""
from train_uncond_dpm import main, parse_args

if __name__ == "__main__":
    main(parse_args())
