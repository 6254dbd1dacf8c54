#!/usr/bin/env python3
"""Headline benchmark: GPT-XL text-to-video token sampling (+ CausalVideoVAE decode), 17 frames @ 256x256.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch of synthetic conditions on every rank:
prefill (120 text tokens) + 5119 KV-cached decode steps for `--batch` videos (5 x 32 x 32 latent tokens each,
BASELINE config 4 / SURVEY.md §8d "C4 ds 8"), the CausalVideoVAE decode of those latents to 17 x 256 x 256
frames, and (N > 1) the one RCCL all-gather of the results.  Batch shards are independent: per-GPU work is
fixed as N grows ("weak").  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_gpt(V, a, device):
    m = V.GPT_models[a.gpt_model](block_size=a.latent ** 2, cls_token_num=120, model_type="t2v",
                                  vae_embed_dim=a.vae_embed_dim, num_frames=a.num_frames, t_downsample_size=4,
                                  caption_dim=2048, head=a.head)
    m.to(device=device, dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32).eval()
    m.init_random_weights(seed=1234)
    return m


def synth_cond(B, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    emb = torch.randn(B, 120, 2048, generator=g) * 0.1
    lens = torch.randint(8, 121, (B,), generator=g)
    mask = torch.zeros(B, 120)
    for b in range(B):
        mask[b, 120 - int(lens[b]):] = 1.0              # left padding (sample_t2i.py:105-119)
    return (emb * mask[:, :, None]).to(device), mask.to(device)


def cpu_baseline(a):
    """The numpy oracle (a port, not the reference) on the host cores, on a bounded sample of the same workload:
    GPT-XL t2v fp32, same shapes, batch `cb`, prefill + a few decode steps."""
    import threadpoolctl  # noqa: F401  (numpy BLAS thread count is reported)
    from oracle import cases, detweights
    from oracle import vlg_oracle as O
    cb, nsteps = 4, 6
    cfg = dict(cases.GPT_SIZES[a.gpt_model], vocab_size=16384, block_size=a.latent ** 2, cls_token_num=120, model_type="t2v",
               num_classes=1000, caption_dim=2048, norm_eps=1e-5, rope_base=10000.0, multiple_of=256,
               vae_embed_dim=a.vae_embed_dim, num_frames=a.num_frames, t_downsample_size=4, head="adapter2",
               adapter_in_std=0.3, adapter_out_std=0.3)
    sd = detweights.gpt_weights(cfg)
    m = O.GPTOracle(cfg, sd, "fp32")
    c, mk = cases.text_cond(cb, 120, 2048)
    t0 = time.time()
    O.generate_t2v(m, c, nsteps, mk)
    dt = time.time() - t0
    try:
        cores = threadpoolctl.threadpool_info()[0]["num_threads"]
    except Exception:
        cores = os.cpu_count()
    return {"value": cb * nsteps / dt, "unit": "video tokens/s", "cores": int(cores), "kind": "port",
            "sample": f"numpy oracle, {a.gpt_model} t2v fp32, batch {cb}, prefill(120)+{nsteps - 1} decode steps "
                      f"({cb * nsteps} tokens at positions 120..{120 + nsteps - 1}) in {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="videos per GPU")
    ap.add_argument("--gpt-model", default="GPT-XL")
    ap.add_argument("--latent", type=int, default=32, help="latent grid (256 px / downsample 8)")
    ap.add_argument("--num-frames", type=int, default=17)
    ap.add_argument("--vae-embed-dim", type=int, default=8)
    ap.add_argument("--new-tokens", type=int, default=0, help="debug: generate fewer than vae_t*latent^2 tokens")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--cfg-scale", type=float, default=1.0)
    ap.add_argument("--head", default="adapter2", choices=["adapter2", "hidden"],
                    help="t2v head: adapter2 (gpt_video.py MSE head, default) or hidden (gpt_video_diff.py DiffLoss, 100 DDPM steps/token)")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--vae-chunk", type=int, default=4, help="videos per vae.decode call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="batch lanes inside generate (0 = auto)")
    ap.add_argument("--no-graph", action="store_true", help="eager decode loop instead of HIP-graph replay")
    ap.add_argument("--attn-inlaunch", action="store_true", help="merge the split-KV partials inside the attention launch (slower)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VLG_BENCH_ONE_DEVICE"):      # rehearsal of the N > 1 path on a one-GPU box: every rank on cuda:0
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VLG_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; "gloo" only for the one-GPU rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import video_llamagen_amd as V
    vae_t = (a.num_frames - 1) // 4 + 1
    N = a.new_tokens or vae_t * a.latent ** 2
    full = N == vae_t * a.latent ** 2
    B = a.batch
    gpt = build_gpt(V, a, device)
    gpt.lanes = a.lanes
    gpt.use_graph = not a.no_graph
    gpt.attn_inlaunch = a.attn_inlaunch
    cond, mask = synth_cond(B, device, seed=1 + rank)
    vae = None
    if not a.no_vae and full:
        # CausalVAEModel constructor defaults with embed_dim = vae_embed_dim (the only decoder topology
        # fully defined in the reference), random-initialised, bf16 as in sample_t2v_1f_diff.py:180
        vae = V.VAE_models["VAE-16"](embed_dim=a.vae_embed_dim).to(device, torch.bfloat16).init_random_weights(seed=3)
        vae.enable_tiling()

    def step():
        lat = V.generate_t2v(gpt, cond, N, mask, cfg_scale=a.cfg_scale)
        out = lat
        if vae is not None:
            z = lat.view(B, vae_t, a.latent, a.latent, a.vae_embed_dim).permute(0, 4, 1, 2, 3).contiguous()
            frames = []
            for i in range(0, B, a.vae_chunk):           # bounded activation footprint; videos are independent
                v = vae.decode(z[i:i + a.vae_chunk])     # sample_t2v_1f_diff.py:175-181
                frames.append(((v.clamp(-1, 1) + 1) * 127.5).to(torch.uint8))   # custom_to_video, :49-58
            out = torch.cat(frames, 0)
        if world > 1:
            # concatenated form [world * B, ...]: accepted by RCCL and by gloo (the stacked form [world, B, ...] is RCCL-only)
            gathered = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype, device=device)
            dist.all_gather_into_tensor(gathered, out.contiguous())
            out = gathered
        return out

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_gen = 0.0
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    tokens = world * B * N * a.steps
    res = {
        "metric": "video tokens/sec (whole job) for GPT-XL t2v 17f@256 sampling + VAE decode",
        "value": tokens / dt, "unit": "video tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"{a.gpt_model} t2v ({a.head} head), 120 text tokens + {N} latent tokens "
                               f"({vae_t}x{a.latent}x{a.latent}, vae_embed_dim {a.vae_embed_dim}), cfg {a.cfg_scale}, "
                               f"{B} videos per GPU, {'CausalVideoVAE decode to 17x256x256 included' if vae is not None else 'VAE decode NOT included'}",
                   "global_batch": world * B, "seq_len": 120 + N, "parallelism": f"batch-shard x{world}"},
        "frames_per_s": (world * B * a.num_frames * a.steps / dt) if vae is not None else None,
        "tokens_per_s_per_gpu": tokens / dt / world,
    }

    if rank == 0 and not a.no_roofline:
        # dominant kernel = split-KV decode attention (reads K,V rows 0..p of one layer).  Extra eager pass with HIP
        # events on the library's launch stream around layer 0's attention kernel of every decode step.
        gpt.time_attn = True
        V.generate_t2v(gpt, cond, N, mask, cfg_scale=a.cfg_scale)
        torch.cuda.synchronize()
        gpt.time_attn = False
        ms, by, n = gpt.attn_timing()
        # An EMPTY event pair on the same stream already measures a few us, so the bracketed time over-states the kernel (rocprofv3's
        # per-dispatch duration is ~3 us shorter).  `achieved` keeps the bracketed (conservative) time; the overhead is reported beside it.
        ovh = gpt.attn_event_overhead_ms()
        if n > 0 and ms > 0:
            ach = by / (ms * 1e-3) / 1e9
            traffic, tsrc = None, None
            try:   # HBM bytes per algorithmic byte measured with rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction), see file
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_attn_fetch.json")))
                ratio = float(np.mean([r["traffic_over_algorithmic"] for r in pmc["rows"] if r["pos"] >= 1000]))
                traffic, tsrc = ratio * by / n, "profiles/r01_pmc_attn_fetch.json (FETCH_SIZE x2, ratio %.4f to algorithmic)" % ratio
            except Exception:
                pass
            res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": traffic, "traffic_source": tsrc, "kernel": "attn_partial_kernel", "launches_timed": n,
                               "avg_launch_us": 1e3 * ms / n, "avg_algorithmic_bytes_per_launch": by / n,
                               "empty_event_pair_us": 1e3 * ovh}
        wb, kb, ob = gpt.algorithmic_bytes()
        res["algorithmic_bytes_per_step"] = {"weights": wb, "kv": kb, "other": ob}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(a)
        except Exception as e:  # reported, never fatal for the GPU numbers
            res["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
