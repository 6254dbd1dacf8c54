#!/usr/bin/env python3
"""Micro-benchmarks of the decode-step kernels through the C-ABI unit entry points (GPU box only).
    python tools/bench_kernels.py attn|linear|all"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd  # noqa: E402,F401
from video_llamagen_amd import _lib as L  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def bench_attn(Bp=32, H=20, S=5240, hd=64):
    dt = torch.bfloat16
    q = torch.randn(Bp, H, hd, device="cuda", dtype=dt)
    k = torch.randn(Bp, H, S, hd, device="cuda", dtype=dt)
    v = torch.randn(Bp, H, S, hd, device="cuda", dtype=dt)
    out = torch.empty(Bp, H * hd, device="cuda", dtype=dt)
    st = L.stream_ptr()
    for pos in (255, 1023, 2679, 5239):
        us = timeit(lambda: L.check(L.lib().vlg_attn_decode(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), Bp, H, S, hd, pos, None, 0, 0, L.VLG_BF16, st)))
        by = 2.0 * Bp * H * hd * (pos + 1) * 2
        print(f"attn Bp={Bp} H={H} hd={hd} pos={pos}: {us:8.1f} us  {by / us / 1e3:8.1f} GB/s (incl. combine + host sync in entry point)")


def bench_attn_rotating(Bp=32, H=20, S=5240, hd=64, nbuf=8):
    """the same launch cycling through `nbuf` distinct KV buffers (as the 36 layers of a decode step do): TLB / page-locality effects"""
    dt = torch.bfloat16
    q = torch.randn(Bp, H, hd, device="cuda", dtype=dt)
    ks = [torch.randn(Bp, H, S, hd, device="cuda", dtype=dt) for _ in range(nbuf)]
    vs = [torch.randn(Bp, H, S, hd, device="cuda", dtype=dt) for _ in range(nbuf)]
    out = torch.empty(Bp, H * hd, device="cuda", dtype=dt)
    st = L.stream_ptr()
    for pos in (1023, 2679, 5239):
        it = [0]

        def call():
            i = it[0] % nbuf
            it[0] += 1
            L.check(L.lib().vlg_attn_decode(L.ptr(q), L.ptr(ks[i]), L.ptr(vs[i]), L.ptr(out), Bp, H, S, hd, pos, None, 0, 0, L.VLG_BF16, st))
        us = timeit(call, iters=4 * nbuf, warm=nbuf)
        by = 2.0 * Bp * H * hd * (pos + 1) * 2
        print(f"attn rotating x{nbuf} pos={pos}: {us:8.1f} us  {by / us / 1e3:8.1f} GB/s")


def bench_linear(M=32):
    dt = torch.bfloat16
    st = L.stream_ptr()
    for name, N, K in (("wqkv", 3840, 1280), ("wo", 1280, 1280), ("w13", 7168, 1280), ("w2", 1280, 3584)):
        x = torch.randn(M, K, device="cuda", dtype=dt)
        w = torch.randn(N, K, device="cuda", dtype=dt)
        o = torch.empty(M, N, device="cuda", dtype=dt)
        us = timeit(lambda: L.check(L.lib().vlg_linear(L.ptr(x), L.ptr(w), L.ptr(o), M, N, K, L.VLG_BF16, st)))
        print(f"linear {name} M={M} N={N} K={K}: {us:8.1f} us  {N * K * 2 / us / 1e3:8.1f} GB/s (gemm + reduce + host sync)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("attn", "all"):
        bench_attn()
    if what in ("attn_rot", "all"):
        bench_attn_rotating()
    if what in ("linear", "all"):
        bench_linear()
