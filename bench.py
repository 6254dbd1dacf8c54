#!/usr/bin/env python3
"""Headline benchmark: GPT-XL text-to-video token sampling (+ CausalVideoVAE decode), 17 frames @ 256x256.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch of synthetic conditions on every rank:
prefill (120 text tokens) + 5119 KV-cached decode steps for the rank's videos (5 x 32 x 32 latent tokens each,
BASELINE config 4 / SURVEY.md 8d "C4 ds 8"), the CausalVideoVAE decode of those latents to 17 x 256 x 256
frames, and (N > 1) the one RCCL all-gather of the results.

Scaling ("B 32 total, sharded 32/16/8/4 per GPU"): with N > 1 the default is `--scaling strong` - the GLOBAL batch
stays `--batch` (32) and every rank takes batch / N videos; `--scaling weak` keeps `--batch` videos per GPU.  At N = 1 they coincide.

Prints ONE JSON line on rank 0.  Besides the contract keys it carries
  roofline        the dominant kernel (split-KV decode attention, HBM-bound), HIP-event timed inside one warm-up step
  roofline_vae    the dominant decoder kernel (halo-tile implicit-GEMM conv, MFMA-bound), HIP-event timed in one decode
  cpu_baseline    the numpy oracle on the host cores on a bounded sample of the same workload (sampling leg + VAE leg)
  extra_configs   short driver-observed runs of BASELINE configs 2, 3, 5 and of the DiffLoss-head variant of config 4
The extras (one call each, ~8 s in all) run before the CPU baseline and only while the process is inside its time budget
(`--budget-s`, default 538 s from start: the driver stops the default run at 600 s); what was skipped is listed.
The CPU baseline runs in a child process during the untimed warm-up steps (see main), so it costs no wall time.
"""
import argparse
import json
import os
import sys
import time

T_START = time.perf_counter()

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def elapsed():
    return time.perf_counter() - T_START


def build_gpt(V, a, device, head=None):
    m = V.GPT_models[a.gpt_model](block_size=a.latent ** 2, cls_token_num=120, model_type="t2v",
                                  vae_embed_dim=a.vae_embed_dim, num_frames=a.num_frames, t_downsample_size=4,
                                  caption_dim=2048, head=head or a.head)
    m.to(device=device, dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32).eval()
    m.init_random_weights(seed=1234)
    if os.environ.get("VLG_BENCH_ONE_DEVICE"):
        # Rehearsal: several ranks share ONE card.  The persistent kernels (decode step, DiffLoss sampler) need their whole grid resident and
        # two of them from two processes can split the compute units between them - each then waits for workgroups that cannot start, the
        # bounded waits run out and the call fails with VLG_ERR_STATE (observed; include/vlg.h "Device-side faults").  One process per GPU,
        # as the driver launches it, never shares a card.
        m.pdecode = False
        m.dl_persist = False
    return m


def projected_strong():
    """One-GPU timings of the per-GPU shards of the 32-video job and the speed-ups they project, from the tracked file tools/bench_shards.py
    writes (measured, not typed in; the driver computes the real efficiency from its own per-N runs)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r04_shard_timings.json")))
    except Exception:
        return None


def synth_cond(B, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    emb = torch.randn(B, 120, 2048, generator=g) * 0.1
    lens = torch.randint(8, 121, (B,), generator=g)
    mask = torch.zeros(B, 120)
    for b in range(B):
        mask[b, 120 - int(lens[b]):] = 1.0              # left padding (sample_t2i.py:105-119)
    return (emb * mask[:, :, None]).to(device), mask.to(device)


def cpu_baseline(a):
    """The numpy oracle (a port, not the reference) on the host cores, on a bounded sample of the same workload: GPT-XL t2v fp32,
    same shapes, batch 4, prefill + 240 decode steps; then one latent frame (16 x 16 cells) of the CausalVideoVAE decoder at full width; then BASELINE
    config 1 exactly (about 20-25 s in all)."""
    import threadpoolctl  # noqa: F401  (numpy BLAS thread count is reported)
    threadpoolctl.threadpool_limits(limits=16)   # a one-GPU box's share of the host (more BLAS threads than that oversubscribe it: 2x slower)
    from oracle import cases, detweights
    from oracle import vlg_oracle as O
    cb, nsteps = 4, 241    # ~15-20 s on 16 host threads since the oracle stopped copying weights per call and attending over unwritten cache rows
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
    del m, sd
    try:
        cores = threadpoolctl.threadpool_info()[0]["num_threads"]
    except Exception:
        cores = os.cpu_count()
    res = {"value": cb * nsteps / dt, "unit": "video tokens/s", "cores": int(cores), "kind": "port",
           "sample": f"numpy oracle, {a.gpt_model} t2v fp32, batch {cb}, prefill(120)+{nsteps - 1} decode steps "
                     f"({cb * nsteps} tokens at positions 120..{120 + nsteps - 1}) in {dt:.1f}s"}
    # VAE leg: the decoder at its real channel widths (512/256/128) on ONE latent frame of 16 x 16 cells -> 1 x 128 x 128 pixels
    # (1/4 of a 256-px frame: 0.37 TFLOP), so the whole baseline stays within ~20-25 s of CPU work
    vcfg = dict(hidden_size=128, z_channels=4, embed_dim=a.vae_embed_dim, hidden_size_mult=(1, 2, 4, 4), num_res_blocks=2)
    vsd = detweights.vae_weights(vcfg)
    vo = O.VAEOracle(vsd, hidden_size=128, hidden_size_mult=(1, 2, 4, 4), num_res_blocks=2)
    z = cases.rng(9).standard_normal((1, a.vae_embed_dim, 1, 16, 16), dtype=np.float32)
    t0 = time.time()
    y = vo.decode(z)
    dv = time.time() - t0
    res["vae"] = {"value": 0.25 / dv, "unit": "256x256-frame equivalents/s", "sample":
                  f"numpy oracle, CausalVideoVAE decoder (constructor defaults, fp32), 1 latent frame 16x16 -> {tuple(y.shape)} in {dv:.1f}s "
                  f"(0.37 TFLOP; a 17-frame 256x256 video is 19.95 TFLOP)"}
    # BASELINE config 1 exactly as BASELINE.md section 3 planned it: LlamaGen-B c2i 256x256 (256 tokens), one class, greedy, fp32, batch 1
    # (the reference's own CPU-runnable case; the same call tests/test_oracle_golden.py pins to the reference's ids)
    b_sd = detweights.gpt_weights(cases.GPT_B)
    bm = O.GPTOracle(cases.GPT_B, b_sd, "fp32")
    t0 = time.time()
    ids = O.generate(bm, np.array([207]), 256, None, cfg_scale=1.0, cfg_interval=-1, sample_logits=False)
    d1 = time.time() - t0
    res["config1"] = {"value": 256 / d1, "unit": "image tokens/s", "sample": f"numpy oracle, GPT-B c2i 16x16 tokens, class 207, greedy, fp32, batch 1: "
                      f"{ids.shape[1]} tokens in {d1:.1f}s"}
    return res


def run_extra_configs(V, device, budget_s):
    """BASELINE configs 2, 3, 5 (one GPU's share) - sampling wall time of one generate() after one warm-up, random weights, bf16.
    Each entry is skipped (and says so) once the process has used its time budget."""
    from video_llamagen_amd.sample_common import synthetic_text
    out, skipped = {}, []

    def timed(fn):
        # ONE call, first use of the handle: includes KV-cache allocation and the instantiation of the decode graph (a few ms)
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t

    def c2i(tag, name, B, grid, cfg_scale, top_k, est_s):
        if elapsed() + est_s > budget_s:
            skipped.append(tag)
            return
        m = V.GPT_models[name](block_size=grid * grid, cls_token_num=1, model_type="c2i").to(device, torch.bfloat16).init_random_weights(seed=1)
        cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(0)).to(device)
        dt = timed(lambda: V.generate(m, cond, grid * grid, cfg_scale=cfg_scale, temperature=1.0, top_k=top_k, top_p=1.0, sample_logits=True, seed=7))
        wb, kb, ob = m.algorithmic_bytes()
        out[tag] = {"workload": f"{name} c2i {16 * grid}x{16 * grid} ({grid * grid} tokens), {B} images, cfg {cfg_scale}, top-k {top_k}, bf16, sampling only",
                    "sampling_s": dt, "tokens_per_s": B * grid * grid / dt, "hbm_floor_s_at_8TBs": (wb + kb + ob) / 8e12}
        del m

    c2i("C2", "GPT-L", 8, 24, 4.0, 2000, 3)          # serve/sample_c2i.py:88-95, the README workload
    if elapsed() + 4 <= budget_s:
        # the same workload through the request front-end (serve/sample_c2i.py:49-64: 8 class prompts + 8 null-class prompts = 16
        # sequences x 576 tokens, vLLM SamplingParams): iteration-level engine on a block-granular KV cache
        m = V.GPT_models["GPT-L"](block_size=576, cls_token_num=1, model_type="c2i").to(device, torch.bfloat16).init_random_weights(seed=1)
        labels = [int(c) for c in torch.randint(0, 1000, (8,), generator=torch.Generator().manual_seed(0))]
        sp = V.SamplingParams(temperature=1.0, top_k=2000, top_p=1.0, max_tokens=576, seed=7)

        def serve():
            eng = V.ContinuousLLMEngine(m, cfg_scale=4.0, max_num_seqs=16, max_tokens=576, kv_block_size=64)
            for i, c in enumerate(labels + [m.num_classes] * 8):
                eng.add_request(str(i), None, sp, [c])
            n = 0
            while eng.has_unfinished_requests():
                n += sum(len(o.outputs[0].token_ids) for o in eng.step())
            assert n == 16 * 576, n
        dt = timed(serve)
        out["C2_serving"] = {"workload": "GPT-L c2i 384x384 through ContinuousLLMEngine: 8 class + 8 null-class prompts = 16 sequences x 576 tokens, cfg 4.0, "
                                         "top-k 2000, bf16, KV blocks of 64 positions, iterations enqueued asynchronously (pinned staging ring)",
                             "sampling_s": dt, "tokens_per_s": 8 * 576 / dt, "sequences_tokens_per_s": 16 * 576 / dt}
        del m
    else:
        skipped.append("C2_serving")
    if elapsed() + 5 <= budget_s:
        m = V.GPT_models["GPT-XL"](block_size=1024, cls_token_num=120, model_type="t2i").to(device, torch.bfloat16).init_random_weights(seed=1)
        cond, mask = synthetic_text(4, 120, 2048, 1, device)
        dt = timed(lambda: V.generate(m, cond, 1024, mask, cfg_scale=7.5, temperature=1.0, top_k=1000, top_p=1.0, sample_logits=True, seed=7))
        wb, kb, ob = m.algorithmic_bytes()
        # the condition prefill alone (8 x 120 rows through all layers + the first token): generate with ONE new token, second call (warm)
        V.generate(m, cond, 1, mask, cfg_scale=7.5, temperature=1.0, top_k=1000, top_p=1.0, sample_logits=True, seed=7)
        dpf = timed(lambda: V.generate(m, cond, 1, mask, cfg_scale=7.5, temperature=1.0, top_k=1000, top_p=1.0, sample_logits=True, seed=7))
        out["C3"] = {"workload": "GPT-XL t2i 512x512 (1024 tokens), 120 text tokens, 4 images, cfg 7.5, top-k 1000, bf16, sampling only (T5 features given)",
                     "sampling_s": dt, "tokens_per_s": 4 * 1024 / dt, "hbm_floor_s_at_8TBs": (wb + kb + ob) / 8e12, "prefill_s": dpf}
        del m
    else:
        skipped.append("C3")
    c2i("C5", "GPT-3B", 32, 24, 1.65, 0, 8)          # one GPU's 32 of the 256 images (gpt.py:445, GETTING_STARTED.md:53): 64 rows, head_dim 100
    torch.cuda.empty_cache()
    return out, skipped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="videos: the global batch under --scaling strong, per GPU under --scaling weak")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="auto = strong when N > 1 (SURVEY.md 8d: 32 videos in total, sharded over the GPUs)")
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
    ap.add_argument("--vae-overlap", action="store_true",
                    help="decode step i's latents on a second stream beside the sampling of step i+1 (measured r02: 20.62 vs 20.59 s/step, "
                         "the two do not share the chip; off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short runs of configs 2, 3, 5 and of the DiffLoss head")
    ap.add_argument("--budget-s", type=float, default=538.0,
                    help="an extra run starts only if the process would still be younger than this when it ends (the driver stops the default run at 600 s)")
    ap.add_argument("--hidden-tokens", type=int, default=256, help="tokens of the short DiffLoss-head run")
    ap.add_argument("--no-graph", action="store_true", help="eager decode loop instead of HIP-graph replay")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()

    if a.cpu_baseline_only:
        # child of the default run (see below): waits for "go" on stdin, times the numpy oracle on the host cores, prints one JSON line.
        # Never touches the GPU.
        if sys.stdin.readline().strip() != "go":
            return
        try:
            print(json.dumps(cpu_baseline(a)), flush=True)
        except Exception as e:
            print(json.dumps({"error": repr(e)}), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VLG_BENCH_ONE_DEVICE"):      # rehearsal of the N > 1 path on a one-GPU box: every rank on cuda:0
        local = 0
    # The CPU baseline runs in a child process on the host cores WHILE the GPU does its untimed warm-up steps 2..W, so that it costs the
    # run no wall time (25 steps of 20 s leave little of the driver's 600 s).  The child is started here, before this process touches
    # the GPU, and sleeps on its stdin until the first warm-up step (the roofline measurement) is over; it is joined before the timed
    # region starts.  With fewer than 2 warm-up steps the baseline runs inline at the end instead.
    cpu_child = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.warmup >= 2:
        import subprocess
        argv = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--gpt-model", a.gpt_model, "--latent", str(a.latent),
                "--num-frames", str(a.num_frames), "--vae-embed-dim", str(a.vae_embed_dim)]
        cpu_child = subprocess.Popen(argv, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT)
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
    from video_llamagen_amd import _lib as L
    from video_llamagen_amd.dist import shard_range
    scaling = a.scaling if a.scaling != "auto" else ("strong" if world > 1 else "weak")
    if scaling == "strong":
        lo, hi = shard_range(a.batch, rank, world)          # contiguous shard of the global batch (dist.py; ragged when N does not divide it)
        B, global_batch = hi - lo, a.batch
        cond_all, mask_all = synth_cond(a.batch, device, seed=1)
        cond, mask = cond_all[lo:hi].contiguous(), mask_all[lo:hi].contiguous()
        del cond_all, mask_all
        if B == 0:
            raise SystemExit("--batch %d leaves rank %d without work at %d GPUs" % (a.batch, rank, world))
    else:
        B, global_batch = a.batch, a.batch * world
        cond, mask = synth_cond(B, device, seed=1 + rank)
    vae_t = (a.num_frames - 1) // 4 + 1
    N = a.new_tokens or vae_t * a.latent ** 2
    full = N == vae_t * a.latent ** 2
    gpt = build_gpt(V, a, device)
    gpt.use_graph = not a.no_graph
    vae = None
    if not a.no_vae and full:
        # CausalVAEModel constructor defaults with embed_dim = vae_embed_dim (the only decoder topology
        # fully defined in the reference), random-initialised, bf16 as in sample_t2v_1f_diff.py:180
        vae = V.VAE_models["VAE-16"](embed_dim=a.vae_embed_dim).to(device, torch.bfloat16).init_random_weights(seed=3)
        vae.enable_tiling()
    if world > 1 and scaling == "strong":
        # ragged shards (N does not divide the batch) are padded to the largest shard for the gather
        Bmax = shard_range(a.batch, 0, world)[1]
    else:
        Bmax = B

    # The library calls are stream-ordered and do not block the host (include/vlg.h), so the decode of one step's latents can be put
    # on a second stream beside the NEXT step's sampling (--vae-overlap).  Every step's frames are complete when the closing
    # torch.cuda.synchronize() returns.
    main_stream = torch.cuda.current_stream(device)
    vae_stream = torch.cuda.Stream(device) if (vae is not None and a.vae_overlap) else main_stream

    def step():
        lat = V.generate_t2v(gpt, cond, N, mask, cfg_scale=a.cfg_scale)
        out = lat
        if vae is not None:
            ready = torch.cuda.Event()
            ready.record(main_stream)
            with torch.cuda.stream(vae_stream):
                vae_stream.wait_event(ready)
                lat.record_stream(vae_stream)
                z = lat.view(B, vae_t, a.latent, a.latent, a.vae_embed_dim).permute(0, 4, 1, 2, 3).contiguous()
                frames = []
                for i in range(0, B, a.vae_chunk):           # bounded activation footprint; videos are independent
                    v = vae.decode(z[i:i + a.vae_chunk])     # sample_t2v_1f_diff.py:175-181
                    frames.append(((v.clamp(-1, 1) + 1) * 127.5).to(torch.uint8))   # custom_to_video, :49-58
                out = torch.cat(frames, 0)
                if world > 1:
                    out = gather(out)
        elif world > 1:
            out = gather(out)
        return out

    def gather(out):
        if out.shape[0] < Bmax:
            out = torch.cat([out, out.new_zeros((Bmax - out.shape[0],) + tuple(out.shape[1:]))], 0)
        # concatenated form [world * B, ...]: accepted by RCCL and by gloo (the stacked form [world, B, ...] is RCCL-only)
        gathered = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype, device=device)
        dist.all_gather_into_tensor(gathered, out.contiguous())
        return gathered

    # ---- warm-up: W untimed steps.  On rank 0 the first of them is also the roofline measurement: the same step with HIP events
    # (on the library's launch stream) around layer 0's attention kernel of every decode step, and around every halo conv launch
    # of the first vae.decode call.  No extra generate is spent on it.
    roof_done = False
    for w in range(a.warmup):
        if w == 0 and rank == 0 and not a.no_roofline:
            gpt.time_attn = True
            L.check(L.lib().vlg_conv_timing(1))
            step()
            torch.cuda.synchronize()
            gpt.time_attn = False
            L.check(L.lib().vlg_conv_timing(0))
            roof_done = True
        else:
            step()
        if rank == 0:   # a progress line per step on stderr (a silent 9-minute run looks hung to a watchdog); no synchronisation added
            log(f"[bench] warm-up step {w + 1}/{a.warmup} enqueued at {elapsed():.0f} s")
        if w == 0 and cpu_child is not None:
            cpu_child.stdin.write("go\n")
            cpu_child.stdin.flush()
    torch.cuda.synchronize()
    cpu_res = None
    if cpu_child is not None:          # joined BEFORE the timed region: nothing but the GPU steps runs inside it
        try:
            out, _ = cpu_child.communicate(timeout=600)
            cpu_res = json.loads(out.strip().splitlines()[-1])
            if "sample" in cpu_res:
                cpu_res["sample"] += "; measured in a child process on the host cores during the GPU's untimed warm-up steps"
        except Exception as e:
            cpu_child.kill()
            cpu_res = {"error": repr(e)}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(a.steps):
        last = step()
        if rank == 0:
            log(f"[bench] step {i + 1}/{a.steps} enqueued at {elapsed():.0f} s")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpt.status(sync=False)      # generate() is asynchronous; a device-side time-out in any timed step voids the number (raises)
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    tokens = global_batch * N * a.steps
    res = {
        "metric": "video tokens/sec (whole job) for GPT-XL t2v 17f@256 sampling + VAE decode",
        "value": tokens / dt, "unit": "video tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"{a.gpt_model} t2v ({a.head} head), 120 text tokens + {N} latent tokens "
                               f"({vae_t}x{a.latent}x{a.latent}, vae_embed_dim {a.vae_embed_dim}), cfg {a.cfg_scale}, "
                               f"{global_batch} videos in total = {B} on this GPU, "
                               f"{'CausalVideoVAE decode to 17x256x256 included' if vae is not None else 'VAE decode NOT included'}",
                   "global_batch": global_batch, "seq_len": 120 + N, "parallelism": f"batch-shard x{world} ({scaling})"},
        "frames_per_s": (global_batch * a.num_frames * a.steps / dt) if vae is not None else None,
        "tokens_per_s_per_gpu": tokens / dt / world,
    }
    if world > 1:
        # what rank 0 holds after the step's all_gather (ragged shards padded to the largest one), and what one GPU's shard timings project
        # for this strong-scaling run (measured on ONE GPU with --batch 32/N --no-vae, the VAE share added; the driver
        # computes the real efficiency from the per-N lines)
        res["gathered_shape"] = list(last.shape) if last is not None else None
        res["config"]["shard_sizes"] = [shard_range(a.batch, r, world)[1] - shard_range(a.batch, r, world)[0] for r in range(world)] \
            if scaling == "strong" else [B] * world
        res["config"]["projected_strong_speedup_from_one_gpu_shards"] = projected_strong()

    if rank == 0 and not a.no_roofline:
        if not roof_done:           # --warmup 0: one extra instrumented step after the timed region
            gpt.time_attn = True
            L.check(L.lib().vlg_conv_timing(1))
            step()
            torch.cuda.synchronize()
            gpt.time_attn = False
            L.check(L.lib().vlg_conv_timing(0))
        ms, by, n = gpt.attn_timing()
        # An EMPTY event pair on the same stream already measures a few us, so the bracketed time over-states the kernel (rocprofv3's
        # per-dispatch duration is ~3 us shorter).  `achieved` keeps the bracketed (conservative) time; the overhead is reported beside it.
        ovh = gpt.attn_event_overhead_ms()
        if n > 0 and ms > 0:
            ach = by / (ms * 1e-3) / 1e9
            traffic, tsrc = None, None
            try:   # HBM bytes per algorithmic byte measured with rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction), see file
                pmc_file = "r04_pmc_attn_fetch.json"   # re-collected in round 4 with tools/pmc_attn_fetch.py (rounds 2 and 3: 1.0005-1.010, the same)
                pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
                ratio = float(np.mean([r["traffic_over_algorithmic"] for r in pmc["rows"] if r["pos"] >= 1000]))
                traffic, tsrc = ratio * by / n, "profiles/%s (FETCH_SIZE x2, ratio %.4f to algorithmic)" % (pmc_file, ratio)
            except Exception:
                pass
            res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": traffic, "traffic_source": tsrc, "kernel": "attn_partial_kernel", "launches_timed": n,
                               "avg_launch_us": 1e3 * ms / n, "avg_algorithmic_bytes_per_launch": by / n,
                               "empty_event_pair_us": 1e3 * ovh, "measured_in": "warm-up step 1 (eager launches, same kernels)"}
        cms, cfl, cn = C_double(), C_double(), C_int64()
        L.check(L.lib().vlg_conv_timing_read(byref(cms), byref(cfl), byref(cn)))
        if cn.value > 0 and cms.value > 0:
            tf = cfl.value / (cms.value * 1e-3) / 1e12
            res["roofline_vae"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
                                   "traffic": None, "kernel": "conv_halo_kernel", "launches_timed": cn.value,
                                   "avg_launch_us": 1e3 * cms.value / cn.value, "avg_flop_per_launch": cfl.value / cn.value,
                                   "measured_in": f"the vae.decode calls of warm-up step 1 ({a.vae_chunk} videos per call), bf16 MFMA"}
        wb, kb, ob = gpt.algorithmic_bytes()
        res["algorithmic_bytes_per_step"] = {"weights": wb, "kv": kb, "other": ob}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_extras:
        # ---- driver-observed short runs of the other configurations (bounded by the time budget; most important first) ----
        try:
            del vae
            del gpt
            torch.cuda.empty_cache()
            extras, skipped = run_extra_configs(V, device, a.budget_s)

            def timed_t2v(g, n):
                torch.cuda.synchronize()
                t = time.perf_counter()
                V.generate_t2v(g, cond, n, mask)
                torch.cuda.synchronize()
                g.status(sync=False)
                return time.perf_counter() - t

            # the secondary shape of config 4 (gpt_video.py:381-401,704-714): spatial downsample 16 -> 2048-wide latent tokens,
            # 5 x 16 x 16 = 1280 of them (S 1400); the latent adapters run as generic GEMMs (C > 16)
            if elapsed() + 9 <= a.budget_s and a.head != "hidden":
                a16 = argparse.Namespace(**vars(a))
                a16.latent, a16.vae_embed_dim = 16, 2048
                g16 = build_gpt(V, a16, device)
                n16 = vae_t * 16 * 16
                V.generate_t2v(g16, cond, 2, mask)
                d16 = timed_t2v(g16, n16)
                extras["C4_ds16"] = {"workload": f"{a.gpt_model} t2v (adapter2 head), 120 text tokens + {n16} latent tokens ({vae_t}x16x16, vae_embed_dim 2048), "
                                                 f"{B} videos, bf16, sampling only", "sampling_s": d16, "tokens_per_s": B * n16 / d16}
                del g16
                torch.cuda.empty_cache()
            else:
                skipped.append("C4_ds16")
            # the headline model behind the request front-end (round 4: sessions for the continuous-latent models): the B videos as B requests through
            # ContinuousLLMEngine, 256 latent tokens each - one host round trip per iteration where generate_t2v replays a graph
            if elapsed() + 8 <= a.budget_s and a.head != "hidden":
                gs = build_gpt(V, a, device)
                ns = 256
                sps = V.SamplingParams(temperature=1.0, max_tokens=ns)

                def serve_t2v():
                    eng = V.ContinuousLLMEngine(gs, max_num_seqs=B, max_tokens=ns)
                    for i in range(B):
                        eng.add_request(str(i), None, sps, prompt_embeds=cond[i], emb_mask=mask[i])
                    n = 0
                    while eng.has_unfinished_requests():
                        n += sum(int(o.outputs[0].latents.shape[0]) for o in eng.step())
                    assert n == B * ns, n
                serve_t2v()
                torch.cuda.synchronize()
                t = time.perf_counter()
                serve_t2v()
                torch.cuda.synchronize()
                dsv = time.perf_counter() - t
                V.generate_t2v(gs, cond, 2, mask)
                dgen = timed_t2v(gs, ns)
                extras["C4_serving"] = {"workload": f"{a.gpt_model} t2v (adapter2 head) through ContinuousLLMEngine: {B} requests x {ns} latent tokens, bf16, one batched "
                                                    f"prefill; generate_t2v of the same {B} x {ns} tokens beside it", "sampling_s": dsv,
                                        "tokens_per_s": B * ns / dsv, "generate_t2v_s": dgen}
                del gs
                torch.cuda.empty_cache()
            else:
                skipped.append("C4_serving")
            # the DiffLoss head (gpt_video_diff.py: 100 DDPM steps per token) in three 256-token windows of the 5120-token sequence - start,
            # middle, end (option debug_pos_offset: decode starts `offset` positions in, over a zero-filled cache prefix) - and the full-length
            # time they interpolate to (trapezoid over the windows' mid positions; the token step grows linearly with the context it attends to)
            if elapsed() + 5 <= a.budget_s and a.head != "hidden":
                gh = build_gpt(V, a, device, head="hidden")
                nh = a.hidden_tokens
                ntot = a.latent ** 2 * vae_t
                win = {}
                dth = timed_t2v(gh, nh)
                win[0] = dth
                extras["C4_diffloss_head"] = {"workload": f"{a.gpt_model} t2v, hidden head + DiffLoss sampler (100 DDPM steps per token, gpt_video_diff.py), "
                                                          f"{B} videos, first {nh} of {ntot} latent tokens (positions 120..{119 + nh}), bf16, sampling only",
                                              "sampling_s": dth, "tokens_per_s": B * nh / dth, "ms_per_token_step": 1e3 * dth / nh}
                for tag, off in (("late", ntot - nh), ("mid", (ntot - nh) // 2)):
                    if elapsed() + 8 > a.budget_s:
                        skipped.append("C4_diffloss_head_" + tag)
                        continue
                    gh.debug_pos_offset = off
                    V.generate_t2v(gh, cond, 2, mask)            # allocation of the full-length cache outside the timed calls
                    dtl = min(timed_t2v(gh, nh) for _ in range(2))   # best of two: the first call after a 31 GB allocation has measured 40 % high once
                    win[off] = dtl
                    extras["C4_diffloss_head_" + tag] = {"workload": f"as C4_diffloss_head, {nh} tokens from position {120 + off} (earlier cache rows zero-filled)",
                                                         "sampling_s": dtl, "tokens_per_s": B * nh / dtl, "ms_per_token_step": 1e3 * dtl / nh}
                if len(win) >= 2:
                    pts = sorted((off + nh / 2.0, d / nh) for off, d in win.items())   # (window mid position, seconds per token step)
                    full = pts[0][1] * pts[0][0] + pts[-1][1] * (ntot - pts[-1][0])      # flat extension to both ends
                    for (p0, t0_), (p1, t1_) in zip(pts, pts[1:]):
                        full += 0.5 * (t0_ + t1_) * (p1 - p0)
                    extras["C4_diffloss_head_full_length_interpolated"] = {
                        "workload": f"{ntot}-token generate of {B} videos with the DiffLoss head, interpolated from the {len(win)} measured windows",
                        "sampling_s": full, "tokens_per_s": B * ntot / full, "windows_ms_per_token_step": {str(int(p)): 1e3 * t for p, t in pts}}
                gh.debug_pos_offset = 0
                del gh
                torch.cuda.empty_cache()
                # the reference's only shipped t2v launch line runs fp32 (scripts/sample/sample_t2v_diff.bash:26, --precision none)
                if elapsed() + 9 <= a.budget_s:
                    a32 = argparse.Namespace(**vars(a))
                    a32.dtype = "fp32"
                    g32 = build_gpt(V, a32, device, head="hidden")
                    V.generate_t2v(g32, cond, 2, mask)
                    d32 = timed_t2v(g32, nh)
                    extras["C4_diffloss_head_fp32"] = {"workload": f"as C4_diffloss_head in fp32 (--precision none), first {nh} tokens", "sampling_s": d32,
                                                       "tokens_per_s": B * nh / d32, "ms_per_token_step": 1e3 * d32 / nh}
                    del g32
                    torch.cuda.empty_cache()
                else:
                    skipped.append("C4_diffloss_head_fp32")
                # DiffLoss.sample's own guidance (cfg_iter != 1, diffloss.py:37-41,240-248; generate_video_diff.py:81-133 passes it through): the B videos
                # as 2B network rows - (conditional, unconditional) pairs (b, b + B) - i.e. 64 rows: the persistent sampler on groups of eight rows
                if elapsed() + 8 <= a.budget_s and 2 * B <= 64:
                    gg = build_gpt(V, a, device, head="hidden")
                    cond2, mask2 = torch.cat([cond, torch.zeros_like(cond)]), torch.cat([mask, mask])

                    def timed_guided(n):
                        torch.cuda.synchronize()
                        t = time.perf_counter()
                        V.generate_t2v(gg, cond2, n, mask2, cfg_iter=2.0)
                        torch.cuda.synchronize()
                        gg.status(sync=False)
                        return time.perf_counter() - t

                    timed_guided(2)
                    dg = timed_guided(nh)
                    extras["C4_diffloss_head_guided"] = {"workload": f"as C4_diffloss_head with DiffLoss.sample's guidance (cfg_iter 2.0): {B} videos = {2 * B} network rows "
                                                                     f"(pairs b, b + {B}), first {nh} tokens, bf16", "sampling_s": dg, "tokens_per_s": B * nh / dg,
                                                         "ms_per_token_step": 1e3 * dg / nh}
                    del gg
                else:
                    skipped.append("C4_diffloss_head_guided")
            else:
                skipped.append("C4_diffloss_head")
            torch.cuda.empty_cache()
            res["extra_configs"] = extras
            res["extra_configs_skipped"] = skipped
        except Exception as e:
            res["extra_configs"] = {"error": repr(e)}
    if cpu_res is not None and rank == 0:
        res["cpu_baseline"] = cpu_res
    elif rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            if elapsed() > a.budget_s + 25:
                raise RuntimeError("skipped: %.0f s into the run, past the time budget" % elapsed())
            res["cpu_baseline"] = cpu_baseline(a)
        except Exception as e:  # reported, never fatal for the GPU numbers
            res["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        res["wall_s_total"] = elapsed()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


from ctypes import byref, c_double as C_double, c_int64 as C_int64  # noqa: E402

if __name__ == "__main__":
    main()
