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


def test_t5_embedder_from_strings_with_a_local_tokenizer_directory(tmp_path):
    """Text -> ids -> features from STRINGS (language/t5.py:60-81): a tokenizer directory built offline with the `tokenizers` package stands in
    for the reference's spiece.model directory (not available offline); captions are cleaned the reference's way first (caption.py), then
    tokenised to 120 padded positions, then encoded - equal to encoding the ids by hand."""
    import json
    from tokenizers import Tokenizer, models, pre_tokenizers
    import video_llamagen_amd as V
    words = ["<pad>", "</s>", "<unk>", "a", "cat", "dog", "photo", "of", "the", "sitting", "on", "mat", "cute"]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(str(tmp_path / "tokenizer.json"))
    json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "pad_token": "<pad>", "eos_token": "</s>", "unk_token": "<unk>", "model_max_length": 120},
              open(tmp_path / "tokenizer_config.json", "w"))
    cfg = dict(cases.TINY_T5, vocab_size=len(words))
    sd = detweights.t5_weights(cfg)
    m = _model(cfg, torch.float32, sd)
    emb = V.T5Embedder("cuda", m, tokenizer_path=str(tmp_path))
    texts = ["A  CUTE-Cat sitting on the mat!!! https://example.com/x.png", "<b>Dog</b> photo #12"]
    # the url rule eats "https://" and "example.com/x"; ".png" alone is no file name ([\S]+ needs a character in front), its "png" goes with the
    # extension rule, the dot stays
    assert [emb.text_preprocessing(t) for t in texts] == ["a cute-cat sitting on the mat!!! .", "dog photo"]
    y, mk = emb.get_text_embeddings(texts)
    assert tuple(y.shape) == (2, 120, cfg["d_model"]) and mk.sum(1).tolist() == [10, 2]     # "a cute - cat sitting on the mat !!! ." / "dog photo"
    ids = emb.tokenizer([emb.text_preprocessing(t) for t in texts], max_length=120, padding="max_length", truncation=True, return_tensors="pt")
    y2, _ = emb.get_text_embeddings_from_ids(ids["input_ids"], ids["attention_mask"])
    assert torch.equal(y, y2)
    ref = O.T5Oracle(cfg, sd, "fp32").encode(ids["input_ids"].numpy(), ids["attention_mask"].numpy())
    valid = ids["attention_mask"].numpy().astype(bool)
    assert np.abs(to_np(y) - ref)[valid].max() < 1e-3 * max(1.0, np.abs(ref[valid]).max())
    with pytest.raises(V._lib.VlgError):
        V.T5Embedder("cuda", m, tokenizer_path=str(tmp_path / "missing"))
