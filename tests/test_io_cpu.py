"""Checkpoint containers and sample writers (video_llamagen_amd.io; reference: sample_t2i.py:60-69, modeling_causalvae.py:578-601,
modeling_videobase.py:42-53, sample_t2v_1f_diff.py:49-58, GETTING_STARTED.md:60)."""
import json
import os

import numpy as np
import pytest
import torch

import video_llamagen_amd  # noqa: F401
from video_llamagen_amd import io as vio


def test_gpt_checkpoint_containers(tmp_path):
    w = {"tok_embeddings.weight": torch.ones(2, 2), "freqs_cis": torch.zeros(1)}
    for key in ("model", "module", "state_dict"):
        assert vio.select_gpt_state_dict({key: w, "optimizer": {}, "steps": 3}) is w
    assert vio.select_gpt_state_dict(w, from_fsdp=True) is w
    with pytest.raises(Exception, match="from-fsdp"):
        vio.select_gpt_state_dict({"weights": w})
    path = str(tmp_path / "c.pt")
    torch.save({"model": w, "steps": 1}, path)
    sd = vio.load_gpt_checkpoint(path)
    assert list(sd) == ["tok_embeddings.weight"]          # freqs_cis dropped


def test_vae_state_dict_selection(monkeypatch):
    ema = {"module.decoder.conv_in.conv.weight": 1, "module.loss.x": 2}
    plain = {"decoder.conv_in.conv.weight": 3, "loss.x": 4}
    monkeypatch.delenv("NOT_USE_EMA_MODEL", raising=False)
    sd = vio.select_vae_state_dict({"ema_state_dict": ema, "state_dict": plain}, ignore_keys=("loss",))
    assert sd == {"decoder.conv_in.conv.weight": 1}                                   # EMA preferred, "module." stripped, loss.* dropped
    monkeypatch.setenv("NOT_USE_EMA_MODEL", "1")
    assert vio.select_vae_state_dict({"ema_state_dict": ema, "state_dict": plain})["decoder.conv_in.conv.weight"] == 3
    monkeypatch.delenv("NOT_USE_EMA_MODEL")
    assert vio.select_vae_state_dict({"ema_state_dict": {}, "state_dict": {"gen_model": plain}}) == plain   # empty EMA; GAN layout
    assert vio.select_vae_state_dict(plain) == plain                                  # bare state dict


def test_find_vae_checkpoint(tmp_path):
    d = tmp_path / "vae"
    d.mkdir()
    with pytest.raises(FileNotFoundError):
        vio.find_vae_checkpoint(str(d))
    (d / "config.json").write_text(json.dumps({"_class_name": "CausalVAEModel", "hidden_size": 32, "embed_dim": 8}))
    for n in ("a.ckpt", "b.ckpt"):
        torch.save({}, str(d / n))
    cfg, ck = vio.find_vae_checkpoint(str(d))
    assert cfg == {"hidden_size": 32, "embed_dim": 8}
    import glob
    assert ck == glob.glob(os.path.join(str(d), "*.ckpt"))[-1]


def test_find_vae_checkpoint_diffusers_layout(tmp_path):
    """modeling_videobase.py:52-53: without a *.ckpt the directory is read the way diffusers' ModelMixin.from_pretrained does -
    diffusion_pytorch_model[.variant].safetensors first, then .bin, optionally under a subfolder; tensors round-trip bit for bit."""
    from safetensors.torch import save_file
    d = tmp_path / "hub"
    (d / "vae").mkdir(parents=True)
    cfgj = json.dumps({"_class_name": "CausalVAEModel", "_diffusers_version": "0.24.0", "hidden_size": 32, "embed_dim": 8})
    (d / "config.json").write_text(cfgj)
    (d / "vae" / "config.json").write_text(cfgj)
    with pytest.raises(FileNotFoundError):
        vio.find_vae_checkpoint(str(d))
    sd = {"decoder.conv_in.conv.weight": torch.randn(4, 3, 3, 3, 3), "decoder.conv_in.conv.bias": torch.randn(4).to(torch.bfloat16)}
    torch.save(sd, str(d / "diffusion_pytorch_model.bin"))
    cfg, path = vio.find_vae_checkpoint(str(d))
    assert cfg == {"hidden_size": 32, "embed_dim": 8} and path.endswith("diffusion_pytorch_model.bin")
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    save_file(sd, str(d / "vae" / "diffusion_pytorch_model.fp16.safetensors"))
    cfg, path = vio.find_vae_checkpoint(str(d))
    assert path.endswith("diffusion_pytorch_model.safetensors")           # safetensors preferred over .bin
    got = vio.load_vae_weight_file(path)
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) and got[k].dtype == sd[k].dtype for k in sd)
    assert vio.select_vae_state_dict(got).keys() == sd.keys()              # a plain state dict passes through unchanged
    cfg, path = vio.find_vae_checkpoint(str(d), subfolder="vae", variant="fp16")
    assert path == os.path.join(str(d), "vae", "diffusion_pytorch_model.fp16.safetensors")
    torch.save({}, str(d / "zz.ckpt"))                                     # a *.ckpt wins, as in the reference
    assert vio.find_vae_checkpoint(str(d))[1].endswith("zz.ckpt")


def test_video_and_npz_writers(tmp_path):
    x = torch.linspace(-1.5, 1.5, 3 * 2 * 4 * 4).view(3, 2, 4, 4)
    u8 = vio.video_to_uint8(x)
    ref = (255 * ((torch.clamp(x, -1, 1) + 1) / 2).permute(1, 2, 3, 0).numpy()).astype(np.uint8)
    assert u8.shape == (2, 4, 4, 3) and u8.dtype == np.uint8 and np.array_equal(u8, ref) and u8.min() == 0 and u8.max() == 255
    files = vio.custom_to_video(x, fps=2.0, output_file=str(tmp_path / "clip.mp4"))
    assert files and all(os.path.exists(f) for f in files)
    if files[0].endswith(".npy"):
        assert np.array_equal(np.load(files[0]), u8)
    p = vio.save_samples_npz(np.zeros((5, 8, 8, 3), np.uint8), str(tmp_path / "s.npz"))
    assert np.load(p)["arr_0"].shape == (5, 8, 8, 3)
    with pytest.raises(ValueError):
        vio.save_samples_npz(np.zeros((5, 8, 8, 3), np.float32), str(tmp_path / "bad.npz"))
