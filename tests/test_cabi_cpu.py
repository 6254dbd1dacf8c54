"""CPU-only: the C-ABI library builds/loads and exports every symbol include/vlg.h declares; host-only entry points
and argument validation behave (no GPU compute is launched here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import video_llamagen_amd  # noqa: F401
    from video_llamagen_amd import _lib, build
    build.build(verbose=False)
    _lib.lib()
    return _lib


def test_header_symbols_exported(L):
    hdr = open(os.path.join(ROOT, "include", "vlg.h")).read()
    names = set(re.findall(r"\b(vlg_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 28
    lib = L.lib()
    for n in sorted(names):
        assert hasattr(lib, n), n
    assert names == set(L.SYMBOLS), names ^ set(L.SYMBOLS)


def test_version_and_rope_host(L):
    from oracle import vlg_oracle as O
    v = (C.c_int * 3)()
    L.lib().vlg_version(v)
    assert v[2] == 950
    buf = np.zeros((1 + 256, 32, 2), np.float32)
    assert L.lib().vlg_rope_table(16, 1, 64, C.c_float(10000.0), 1, buf.ctypes.data_as(C.c_void_p)) == 0
    np.testing.assert_allclose(buf, O.rope_table_2d(16, 64, 10000.0, 1), atol=3e-6)
    assert L.lib().vlg_rope_table(16, 1, 64, C.c_float(10000.0), 1, None) == -1
    assert b"vlg_rope_table" in L.lib().vlg_last_error()


def test_config_validation_without_gpu(L):
    h = C.c_void_p()
    bad = L.GptConfig(dim=100, n_layer=1, n_head=3, vocab_size=8, block_size=16, cls_token_num=1, model_type=0, dtype=0)
    assert L.lib().vlg_gpt_create(C.byref(bad), C.byref(h)) == -2            # dim % n_head
    bad = L.GptConfig(dim=128, n_layer=1, n_head=2, vocab_size=8, block_size=15, cls_token_num=1, model_type=0, dtype=0)
    assert L.lib().vlg_gpt_create(C.byref(bad), C.byref(h)) == -2            # block_size not a square (gpt.py:293)
    bad = L.GptConfig(dim=128, n_layer=1, n_head=2, vocab_size=8, block_size=16, cls_token_num=1, model_type=7, dtype=0)
    assert L.lib().vlg_gpt_create(C.byref(bad), C.byref(h)) == -3
    assert b"please check model type" in L.lib().vlg_last_error()           # gpt.py:277
    bad = L.GptConfig(dim=144, n_layer=1, n_head=2, vocab_size=8, block_size=16, cls_token_num=1, model_type=0, dtype=0)
    assert L.lib().vlg_gpt_create(C.byref(bad), C.byref(h)) == -3            # head_dim 72
    assert L.lib().vlg_gpt_create(None, C.byref(h)) == -1
    assert L.lib().vlg_vq_create(None, None) == -1
    assert L.lib().vlg_vae_create(None, None) == -1


def test_host_mirror_registry():
    import video_llamagen_amd as V
    assert set(V.GPT_models) == {"GPT-B", "GPT-L", "GPT-XL", "GPT-XXL", "GPT-XXXL", "GPT-1B", "GPT-3B", "GPT-7B"}   # gpt.py:467-470
    m = V.GPT_models["GPT-XL"](block_size=1024, cls_token_num=120, model_type="t2i")
    assert (m.config.n_layer, m.config.n_head, m.config.dim) == (36, 20, 1280) and m.model_type == "t2i"
    assert V.GPT_models["GPT-3B"]().config.dim // 32 == 100                  # hd = 100
    with pytest.raises(Exception, match="please check model type"):
        V.GPT_models["GPT-B"](model_type="nope")
    with pytest.raises(Exception, match="please check model type"):
        V.generate(V.GPT_models["GPT-B"](model_type="t2v", cls_token_num=120), None, 4)        # generate.py:144
    with pytest.raises(Exception, match="please check model type"):
        V.generate_t2v(V.GPT_models["GPT-B"](), None, 4)
    assert set(V.VQ_models) == {"VQ-16", "VQ-8"} and "VAE-16" in V.VAE_models
    vae = V.VAE_models["VAE-16"](embed_dim=8)
    assert vae.tile_latent_min_size == 64 and vae.tile_latent_min_size_t == 5 and vae.config.embed_dim == 8   # Q15
    shapes = vae.decoder_param_shapes()
    assert shapes["decoder.conv_in.conv.weight"] == (512, 4, 3, 3, 3)
    assert shapes["decoder.up.1.upsample.conv.conv.weight"] == (256, 256, 1, 3, 3)
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert 100e6 < n < 140e6                                                 # decoder side of the 135.4 M-param VAE
