"""GPU: the train_uncond_dpm / sample_uncond counterparts end to end on a reduced model: YAML -> construct by
dotted name -> a few optimiser steps (grad-accum 2) -> checkpoint with the reference's dict layout -> resume ->
sampler loads the EMA weights and writes PNGs named f'{i: 010d}.png'."""
import os
import subprocess
import sys

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_checkpoint_resume_and_sample(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg = yaml.load(open(os.path.join(ROOT, "configs/cifar10/ddm_uncond_const_uncond_unet.yaml")), Loader=yaml.SafeLoader)
    cfg["model"]["unet"].update(model_channels=64, num_blocks=1)
    cfg["data"]["batch_size"] = 8
    res = str(tmp_path / "run")
    cfg["trainer"].update(results_folder=res, train_num_steps=6, save_and_sample_every=3, log_freq=1, test_before=False,
                          ema_update_after_step=2, ema_update_every=1)
    cfg["sampler"].update(batch_size=4, sample_num=8, ckpt_path=os.path.join(res, "model-1.pt"), save_folder=os.path.join(res, "png"))
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_uncond_dpm.py"), "--cfg", path, "--max-steps", "3"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "images/sec=" in r.stdout
    ck = torch.load(os.path.join(res, "model-1.pt"), map_location="cpu", weights_only=True)
    assert set(ck) == {"step", "model", "opt", "lr_scheduler", "ema", "scaler"} and ck["step"] == 3
    assert "model.model.enc.16x16_block0.qkv.weight" in ck["model"] and "eps" in ck["model"]
    assert "ema_model.model.model.enc.32x32_conv.weight" in ck["ema"] and "online_model.eps" in ck["ema"]
    assert os.path.exists(os.path.join(res, "sample-1.png"))
    # resume from milestone 1 and run to step 6
    cfg["trainer"]["resume_milestone"] = 1
    yaml.safe_dump(cfg, open(path, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_uncond_dpm.py"), "--cfg", path], capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "[Train Step] 4/6" in r.stdout and "[Train Step] 6/6" in r.stdout
    assert torch.load(os.path.join(res, "model-2.pt"), map_location="cpu", weights_only=True)["step"] == 6
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sample_uncond.py"), "--cfg", path], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    names = sorted(os.listdir(os.path.join(res, "png")))
    assert names == [f"{i: 010d}.png" for i in range(8)], names


def test_latent_train_checkpoint_and_sample(tmp_path):
    """train_uncond_ldm counterpart on a reduced latent config: first stage built from model.first_stage, std-rescaling on
    the first batch (on_train_batch_start), 'scale_factor' carried by the checkpoint, sampler decodes to image space."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg = yaml.load(open(os.path.join(ROOT, "configs/celebahq/celeb_uncond_ddm_const2_unet_ldm.yaml")), Loader=yaml.SafeLoader)
    cfg["model"].update(image_size=[64, 64], default_scale=False, scale_factor=1.0)
    cfg["model"]["first_stage"]["ddconfig"].update(ch=32, resolution=[64, 64])
    cfg["model"]["unet"].update(img_resolution=16, model_channels=64, num_blocks=1, attn_resolutions=[8])
    cfg["data"].update(image_size=[64, 64], batch_size=4)
    res = str(tmp_path / "run")
    cfg["trainer"].update(results_folder=res, train_num_steps=4, save_and_sample_every=2, log_freq=1, test_before=False,
                          gradient_accumulate_every=2, ema_update_after_step=1, ema_update_every=1)
    cfg["sampler"].update(batch_size=2, sample_num=4, ckpt_path=os.path.join(res, "model-1.pt"), save_folder=os.path.join(res, "png"))
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_uncond_ldm.py"), "--cfg", path, "--max-steps", "2"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "USING STD-RESCALING" in r.stdout and "images/sec=" in r.stdout
    ck = torch.load(os.path.join(res, "model-1.pt"), map_location="cpu", weights_only=True)
    assert ck["step"] == 2 and "scale_factor" in ck["model"] and float(ck["model"]["scale_factor"]) != 1.0
    assert "first_stage_model.encoder.conv_in.weight" in ck["model"]
    assert os.path.exists(os.path.join(res, "sample-1.png"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sample_uncond.py"), "--cfg", path], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    names = sorted(os.listdir(os.path.join(res, "png")))
    assert names == [f"{i: 010d}.png" for i in range(4)], names
    from PIL import Image
    assert Image.open(os.path.join(res, "png", names[0])).size == (64, 64)
