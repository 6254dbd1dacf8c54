"""T5 text encoder (the conditioning step of t2i / t2v, language/t5.py:60-81) through the C-ABI vs transformers' T5EncoderModel goldens
and the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import cases, detweights
from oracle import vlg_oracle as O
from vlg_testutil import to_np

pytestmark = pytest.mark.gpu


def _model(cfg, dtype, sd):
    import video_llamagen_amd as V
    m = V.T5EncoderModel(cfg).to("cuda", dtype).eval()
    _, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert unexpected == []
    return m


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_t5_vs_transformers_golden(golden, dt):
    g = golden("t5")
    cfg = cases.TINY_T5
    m = _model(cfg, torch.float32 if dt == "fp32" else torch.bfloat16, detweights.t5_weights(cfg))
    y = to_np(m(input_ids=torch.from_numpy(g["t5_ids"]), attention_mask=torch.from_numpy(g["t5_mask"]))["last_hidden_state"])
    ref = g[f"t5_{dt}"]
    assert y.shape == ref.shape
    valid = g["t5_mask"].astype(bool)          # padded positions are masked out downstream (emb_masks, generate.py:156-165)
    tol = 3e-4 if dt == "fp32" else 6e-2
    assert np.abs(y - ref)[valid].max() < tol * max(1.0, np.abs(ref).max())


def test_t5_real_widths_vs_oracle():
    """flan-t5-xl's layer shape (d_model 2048, 32 heads x 64, d_ff 5120; 2 of its 24 layers), 120 tokens as language/t5.py:20 pads to,
    ragged attention masks; fp32 handle against the oracle, bf16 handle within bf16 tolerance; T5Embedder call surface."""
    import video_llamagen_amd as V
    cfg = dict(V.t5_model.FLAN_T5_XL, num_layers=2, vocab_size=512)
    sd = detweights.t5_weights(cfg)
    ids = cases.rng(72).integers(1, cfg["vocab_size"], size=(3, 120)).astype(np.int64)
    mask = np.zeros((3, 120), np.int64)
    for b, n in enumerate((120, 7, 33)):
        mask[b, :n] = 1
    ids = ids * mask
    ref = O.T5Oracle(cfg, sd, "fp32").encode(ids, mask)
    valid = mask.astype(bool)
    scale = max(1.0, np.abs(ref[valid]).max())
    for dtype, tol in ((torch.float32, 1e-3), (torch.bfloat16, 8e-2)):
        m = _model(cfg, dtype, sd)
        emb = V.T5Embedder("cuda", m)
        y, mk = emb.get_text_embeddings_from_ids(torch.from_numpy(ids), torch.from_numpy(mask))
        assert tuple(y.shape) == (3, 120, 2048) and y.dtype == dtype and torch.equal(mk.cpu(), torch.from_numpy(mask))
        assert np.abs(to_np(y) - ref)[valid].max() < tol * scale, dtype
        with pytest.raises(V._lib.VlgError):
            emb.get_text_embeddings(["a cat"])                # no tokenizer in this package (sentencepiece model not shipped)
        del m
