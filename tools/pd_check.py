#!/usr/bin/env python3
"""Quick A/B of the persistent decode step (csrc/pdecode.hip) against the launch chain: tiny fp32 / bf16 models vs the reference goldens,
then full-width layers (GPU box only).  python tools/pd_check.py [quick]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import video_llamagen_amd as V  # noqa: E402
from oracle import cases  # noqa: E402
from vlg_testutil import product_gpt, to_np  # noqa: E402

SPIN = int(os.environ.get("PD_SPIN", "200000"))


def tiny(tag, cfg, dt):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "gpt.npz"))
    m, _ = product_gpt(cfg, torch.float32 if dt == "fp32" else torch.bfloat16)
    m.debug_spin_max = SPIN
    cond = torch.from_numpy(cases.class_ids(3, cfg["num_classes"])) if cfg["model_type"] == "c2i" else None
    masks = None
    if cond is None:
        c, mk = cases.text_cond(3, cfg["cls_token_num"], cfg["caption_dim"])
        cond, masks = torch.from_numpy(c), torch.from_numpy(mk)
    out = {}
    for pd in (False, True):
        m.pdecode = pd
        t = time.time()
        ids, tr = V.generate(m, cond, cfg["block_size"], masks, cfg_scale=2.5, cfg_interval=6, sample_logits=False, return_trace=True)
        torch.cuda.synchronize()
        out[pd] = (ids.cpu().numpy(), to_np(tr))
        print(tag, dt, "pdecode", pd, "%.3f s" % (time.time() - t), flush=True)
    d = np.abs(out[True][1] - out[False][1]).max()
    same = (out[True][0] == out[False][0]).all()
    ref_ok = (out[True][0] == g[f"{tag}_fp32_cfg_ids"]).all() if dt == "fp32" else None
    print(tag, dt, "max |dlogits| pd vs chain", d, "ids equal", same, "ids == reference", ref_ok, flush=True)
    return same and (ref_ok is not False)


def wide(name, rows, L=2, N=24, t2v=False):
    dev = "cuda"
    if t2v:
        m = V.Transformer(V.ModelArgs(dim=1280, n_layer=L, n_head=20, block_size=1024, cls_token_num=120, model_type="t2v", vae_embed_dim=8,
                                      num_frames=17, t_downsample_size=4)).to(dev, torch.bfloat16)
        m.init_random_weights(seed=3)
        g = torch.Generator().manual_seed(0)
        cond = torch.randn(rows, 120, 2048, generator=g) * 0.1
        mask = torch.zeros(rows, 120)
        for b in range(rows):
            mask[b, 120 - (8 + 3 * b):] = 1
        cond = cond * mask[:, :, None]
        run = lambda: V.generate_t2v(m, cond, N, mask)
    else:
        dim, nh = {"GPT-L": (1024, 16), "GPT-XL": (1280, 20), "GPT-B": (768, 12)}[name]
        m = V.Transformer(V.ModelArgs(dim=dim, n_layer=L, n_head=nh, block_size=576, cls_token_num=1, model_type="c2i")).to(dev, torch.bfloat16)
        m.init_random_weights(seed=1)
        cond = torch.randint(0, 1000, (rows // 2,), generator=torch.Generator().manual_seed(0)).to(dev)
        run = lambda: V.generate(m, cond, N, cfg_scale=4.0, sample_logits=False, return_trace=True)
    m.debug_spin_max = SPIN
    res = {}
    for pd in (False, True):
        m.pdecode = pd
        r = run()
        torch.cuda.synchronize()
        t = time.time()
        r = run()
        torch.cuda.synchronize()
        print(name, rows, "rows pdecode", pd, "%.4f s" % (time.time() - t), flush=True)
        res[pd] = r
    if t2v:
        a, b = res[True], res[False]
        sc = max(1.0, b.abs().max().item())
        print("  first token equal", torch.equal(a[:, 0], b[:, 0]), "max diff first 3", (a[:, :3] - b[:, :3]).abs().max().item() / sc, flush=True)
    else:
        (ia, ta), (ib, tb) = res[True], res[False]
        sc = tb.abs().max().item()
        print("  step-1 logits rel diff", ((ta[1] - tb[1]).abs().max() / sc).item(), "ids equal frac", (ia == ib).float().mean().item(), flush=True)


if __name__ == "__main__":
    ok = True
    ok &= tiny("c2i", cases.TINY_C2I, "fp32")
    ok &= tiny("t2i", cases.TINY_T2I, "fp32")
    ok &= tiny("c2i", cases.TINY_C2I, "bf16")
    if "quick" not in sys.argv:
        wide("GPT-L", 16)
        wide("GPT-XL", 8)
        wide("GPT-XL-t2v", 4, t2v=True, N=40)
    print("PD_CHECK", "OK" if ok else "MISMATCH")
