#!/usr/bin/env python3
"""Do an HBM-bound decode-attention stream and a latency-bound skinny-GEMM chain share the chip when they come from two HIP streams?
Graph A = `layers` split-KV attention launches (16 rows, rotating KV buffers), graph B = `layers` x 4 skinny GEMMs (M rows, GPT-XL shapes,
cold weights).  Times A alone, B alone, and both launched back to back on two streams.  GPU box only:  python tools/overlap_lab.py [rows] [pos]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd  # noqa: E402,F401
from video_llamagen_amd import _lib as L  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 2679
layers, H, hd, D, F = 36, 20, 64, 1280, 3584
S = ((pos + 8) // 8) * 8
dt = torch.bfloat16
dev = "cuda"

q = torch.randn(rows, H, hd, device=dev, dtype=dt)
nbuf = 8
ks = [torch.randn(rows, H, S, hd, device=dev, dtype=dt) for _ in range(nbuf)]
vs = [torch.randn(rows, H, S, hd, device=dev, dtype=dt) for _ in range(nbuf)]
ao = torch.empty(rows, D, device=dev, dtype=dt)
W = [[torch.randn(n, k, device=dev, dtype=dt) * 0.02 for (n, k) in ((3 * D, D), (D, D), (2 * F, D), (D, F))] for _ in range(layers)]
x = torch.randn(rows, D, device=dev, dtype=dt)
g = torch.randn(rows, F, device=dev, dtype=dt)

# optional CU partition: argv[3] = number of CUs (of 256, every k-th one) given to the GEMM stream; the attention stream gets the others
gemm_cus = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if gemm_cus:
    hip = C.CDLL("libamdhip64.so")
    every = 256 // gemm_cus
    bits_b = [1 if (i % every) == 0 else 0 for i in range(256)]
    bits_a = [1 - b for b in bits_b]

    def masked(bits):
        words = (C.c_uint32 * 8)(*[sum(bits[32 * w + j] << j for j in range(32)) for w in range(8)])
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(8), words)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(st.value)
    sa, sb = masked(bits_a), masked(bits_b)
else:
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def attn_chain(st):
    for l in range(layers):
        L.check(L.lib().vlg_attn_decode(L.ptr(q), L.ptr(ks[l % nbuf]), L.ptr(vs[l % nbuf]), L.ptr(ao), rows, H, S, hd, pos, None, 0, 0, L.VLG_BF16,
                                        C.c_void_p(st.cuda_stream)))


def gemm_chain():
    for l in range(layers):
        a = x @ W[l][0].t()
        b = x @ W[l][1].t()
        c = x @ W[l][2].t()
        d = g @ W[l][3].t()
    return a, b, c, d


# warm both paths (scratch growth, hipBLASLt heuristics) before capture
with torch.cuda.stream(sa):
    attn_chain(sa)
with torch.cuda.stream(sb):
    gemm_chain()
torch.cuda.synchronize()

ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(ga, stream=sa):
    attn_chain(sa)
with torch.cuda.graph(gb, stream=sb):
    keep = gemm_chain()
torch.cuda.synchronize()


def wall(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def run_a():
    with torch.cuda.stream(sa):
        ga.replay()


def run_b():
    with torch.cuda.stream(sb):
        gb.replay()


def run_ab():
    run_a()
    run_b()


ta, tb, tab = wall(run_a), wall(run_b), wall(run_ab)
kvb = 2.0 * rows * H * hd * (pos + 1) * 2
print(f"rows {rows} pos {pos} gemm CUs {gemm_cus or 256}: attention graph {ta / layers:.1f} us/layer ({kvb / (ta / layers) / 1e3:.0f} GB/s), gemm graph {tb / layers:.1f} us/layer, "
      f"both on two streams {tab / layers:.1f} us/layer (serial sum {(ta + tb) / layers:.1f}, ideal max {max(ta, tb) / layers:.1f})")
