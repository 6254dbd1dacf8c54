"""End-to-end sample entry points (counterparts of the reference's sample_*.py) on the GPU, random-init weights."""
import argparse
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ns(**kw):
    return argparse.Namespace(**kw)


def test_sample_c2i(tmp_path):
    import video_llamagen_amd  # noqa: F401
    from video_llamagen_amd import sample_c2i
    out = str(tmp_path / "c2i")
    sample_c2i.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="c2i", cls_token_num=1, precision="bf16", vq_model="VQ-16", vq_ckpt=None,
                        codebook_size=16384, codebook_embed_dim=8, image_size=256, downsample_size=16, num_classes=1000, cfg_scale=4.0,
                        cfg_interval=-1, seed=0, top_k=2000, temperature=1.0, top_p=1.0, num_samples=2, out=out))
    img = np.load(out + ".npy")
    assert img.shape == (2, 256, 256, 3) and img.dtype == np.uint8 and img.std() > 0


def test_sample_t2i_and_t2v(tmp_path):
    import video_llamagen_amd  # noqa: F401
    from video_llamagen_amd import sample_t2i, sample_t2v
    out = str(tmp_path / "t2i")
    sample_t2i.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="t2i", cls_token_num=120, precision="bf16", vq_model="VQ-16", vq_ckpt=None,
                        codebook_size=16384, codebook_embed_dim=8, image_size=256, downsample_size=16, cfg_scale=7.5, seed=0, top_k=1000,
                        temperature=1.0, top_p=1.0, num_samples=2, out=out))
    assert np.load(out + ".npy").shape == (2, 256, 256, 3)
    out = str(tmp_path / "t2v")
    sample_t2v.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="t2v", cls_token_num=120, precision="bf16", vae_model="VAE-16", vae_ckpt=None,
                        vae_embed_dim=8, tile_overlap_factor=0.125, image_size=64, downsample_size=8, num_frames=5, t_downsample_size=4,
                        cfg_scale=1.0, seed=0, num_samples=2, out=out))
    vid = np.load(out + ".npy")
    assert vid.shape == (2, 5, 64, 64, 3) and vid.dtype == np.uint8      # 2 latent frames -> 2T-1 = 3 -> 5 frames
