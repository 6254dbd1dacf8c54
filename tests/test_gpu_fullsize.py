"""BASELINE configs 2, 3 and 5 at their REAL sizes on the GPU against the numpy oracle on the same deterministic weights:
prefill + 3 decode steps (4 tokens) with the sampling settings of the reference's scripts, under shared Exp(1) noise.

  C2  GPT-L  (24 layers, D 1024)  c2i 24x24, 8 classes, CFG 4.0, top-k 2000           serve/sample_c2i.py:88-95
  C3  GPT-XL (36 layers, D 1280)  t2i 32x32, 120 text tokens with ragged masks, CFG 7.5, top-k 1000   sample_t2i.py:105-129,163-167
  C5  GPT-3B (24 layers, D 3200, head_dim 100)  c2i 24x24, 32 classes -> 64 rows, CFG 1.65   gpt.py:445, GETTING_STARTED.md:53

Bars.  fp32 handle: combined (CFG) logits within 2e-3 of the logit range on every step a sample still follows the oracle's
trajectory; sampled ids token for token equal to the oracle's until a draw the oracle itself decided by less than 2e-2 in
log(best / runner-up) of p/q (logit error 2e-3 * range moves that log-ratio by at most ~4e-3 * range) or whose winner sits on the
top-k boundary.  bf16 handle (the benchmark kernels): first-step logits within 8e-2 of the range (bf16 weights/activations through
24-36 layers), first token equal wherever decided by more than that; then all 4 steps again teacher-forced on the oracle's ids (round 4).
"""
import numpy as np
import pytest
import torch

from oracle import cases, detweights
from oracle import vlg_oracle as O
from vlg_testutil import product_gpt, to_np

pytestmark = pytest.mark.gpu

N_NEW = 4


def _full_cfg(name, model_type, block, cls):
    return dict(cases.GPT_SIZES[name], vocab_size=16384, block_size=block, cls_token_num=cls, model_type=model_type, num_classes=1000,
                caption_dim=2048, norm_eps=1e-5, rope_base=10000.0, multiple_of=256, head="logits")


def _draw_stats(logits, q, temperature, top_k, top_p):
    """Per row: log(best / runner-up) of p/q and the number of kept tokens ranked below the winner (0 = winner on the filter edge)."""
    _, probs = O.sample(logits, temperature, top_k, top_p, True, q)
    sc = probs / q
    order = np.argsort(-sc, axis=-1)[:, :2]
    best = np.take_along_axis(sc, order[:, :1], -1)[:, 0]
    second = np.maximum(np.take_along_axis(sc, order[:, 1:2], -1)[:, 0], 1e-38)
    pw = np.take_along_axis(probs, order[:, :1], -1)
    slack = ((probs > 0) & (probs < pw)).sum(-1)
    return np.log(best / second), slack


def _check(cfg, cond, masks, B, sampling, cfg_scale):
    import video_llamagen_amd as V
    sd = detweights.gpt_weights(cfg)
    V_ = cfg["vocab_size"]
    noise = cases.exp_noise((N_NEW, B, V_), seed=7)
    tr = {}
    om = O.GPTOracle(cfg, sd, "fp32")
    ref_ids = O.generate(om, cond, N_NEW, masks, cfg_scale=cfg_scale, sample_logits=True, noise=noise, trace=tr, **sampling)
    ref_lg = np.stack(tr["logits"], 0)                              # [N, B, V] combined logits
    del om
    stats = [_draw_stats(ref_lg[i], noise[i], sampling["temperature"], sampling["top_k"], sampling["top_p"]) for i in range(N_NEW)]
    rng_ = float(ref_lg.max() - ref_lg.min())
    tc = torch.from_numpy(cond)
    tm = torch.from_numpy(masks) if masks is not None else None
    for dt, tol, eps in ((torch.float32, 2e-3, 2e-2), (torch.bfloat16, 8e-2, 0.5)):
        m, _ = product_gpt(cfg, dt, sd=sd)
        ids, trace = V.generate(m, tc, N_NEW, tm, cfg_scale=cfg_scale, sample_logits=True, noise=torch.from_numpy(noise), return_trace=True,
                                **sampling)
        ids, lg = ids.cpu().numpy(), to_np(trace)
        del m
        torch.cuda.empty_cache()
        assert ids.shape == (B, N_NEW) and np.isfinite(lg).all()
        forks = 0
        for b in range(B):
            last = N_NEW if dt == torch.float32 else 1              # bf16: later steps see a different KV history
            for i in range(last):
                err = np.abs(lg[i, b] - ref_lg[i, b]).max()
                assert err < tol * rng_, (str(dt), b, i, err, rng_)
                if ids[b, i] != ref_ids[b, i]:
                    margin, slack = stats[i][0][b], stats[i][1][b]
                    assert margin < eps or slack == 0, (str(dt), b, i, margin, slack)
                    forks += 1
                    break
        assert forks <= max(1, B // 4), (str(dt), forks)
        if dt == torch.bfloat16:
            # round 4: the later steps too - teacher-forced on the oracle's ids (vlg_gpt_set_teacher), so the bf16 handle's KV history holds the
            # oracle's tokens and every step's logits are comparable (same 8e-2 of the range)
            m, _ = product_gpt(cfg, dt, sd=sd)
            _, trace = V.generate(m, tc, N_NEW, tm, cfg_scale=cfg_scale, sample_logits=True, noise=torch.from_numpy(noise), return_trace=True,
                                  teacher=torch.from_numpy(ref_ids.astype(np.int32)), **sampling)
            lg = to_np(trace)
            del m
            torch.cuda.empty_cache()
            for i in range(N_NEW):
                err = np.abs(lg[i] - ref_lg[i]).max()
                assert err < tol * rng_, ("bf16 teacher-forced", i, err, rng_)


def test_c2_gpt_l_c2i_full_size():
    cfg = _full_cfg("GPT-L", "c2i", 576, 1)
    cond = cases.class_ids(8, 1000, seed=0)
    _check(cfg, cond, None, 8, dict(temperature=1.0, top_k=2000, top_p=1.0), 4.0)


def test_c3_gpt_xl_t2i_full_size():
    cfg = _full_cfg("GPT-XL", "t2i", 1024, 120)
    c, mk = cases.text_cond(4, 120, 2048, lens=[120, 8, 57, 33])
    _check(cfg, c, mk, 4, dict(temperature=1.0, top_k=1000, top_p=1.0), 7.5)


def test_c5_gpt_3b_c2i_64_rows_full_size():
    cfg = _full_cfg("GPT-3B", "c2i", 576, 1)
    cond = cases.class_ids(32, 1000, seed=3)
    _check(cfg, cond, None, 32, dict(temperature=1.0, top_k=0, top_p=1.0), 1.65)
