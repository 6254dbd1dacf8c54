"""`generate` with the reference signature (autoregressive/models/generate.py:127-180) over vlg_gpt_generate.

The whole prefill + decode loop (CFG batch doubling :130-142, cache setup :154, mask fix-up :156-165, prefill
:173, N-1 decode steps :177, sampling :57-66) runs inside libvlg on the GPU; this file only validates and
marshals arguments the way the reference does.
"""
import ctypes as C

import torch

from . import _lib as L


def _params(cfg_scale, cfg_interval, temperature=1.0, top_k=0, top_p=1.0, sample_logits=True, seed=None, **_):
    if seed is None:
        seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
    return L.SamplingParams(cfg_scale=float(cfg_scale), cfg_interval=int(cfg_interval), temperature=float(temperature),
                            top_k=int(top_k or 0), top_p=float(top_p), sample_logits=1 if sample_logits else 0,
                            seed=int(seed))


def _run(model, cond, max_new_tokens, emb_masks, cfg_scale, cfg_interval, noise, trace, sampling_kwargs, cfg_iter=1.0, teacher=None):
    model._ensure_handle()
    dev = model._device
    B = cond.shape[0]
    if model.model_type == 'c2i':
        T = 1
        cond_d = cond.to(device=dev, dtype=torch.int64).contiguous()
    else:
        T = cond.shape[1]
        if T != model.cls_token_num:
            raise L.VlgError(-2, "cond has %d tokens, model expects cls_token_num=%d" % (T, model.cls_token_num))
        if cond.shape[2] != model.config.caption_dim:
            raise L.VlgError(-2, "cond feature dim %d != caption_dim %d" % (cond.shape[2], model.config.caption_dim))
        cond_d = cond.to(device=dev, dtype=torch.float32).contiguous()
    max_batch_size_cfg = B * 2 if cfg_scale > 1.0 else B
    model.setup_caches(max_batch_size=max_batch_size_cfg, max_seq_length=T + max_new_tokens,
                       dtype=model.tok_embeddings.weight.dtype)
    mask_d = None
    if emb_masks is not None:
        assert emb_masks.shape[0] == B            # generate.py:157-158
        assert emb_masks.shape[-1] == T
        mask_d = emb_masks.to(device=dev, dtype=torch.float32).contiguous()
    sp = _params(cfg_scale, cfg_interval, **sampling_kwargs)
    latent = model._head_code() != L.VLG_HEAD_LOGITS
    width = model.config.vae_embed_dim if latent else model.config.vocab_size
    noise_d = None
    if noise is not None:
        noise_d = noise.to(device=dev, dtype=torch.float32).contiguous()
        want = (max_new_tokens, B, model.config.vocab_size) if not latent else \
            (max_new_tokens, int(model.config.num_sampling_steps) + 1, B, model.config.vae_embed_dim)
        assert tuple(noise_d.shape) == want, (tuple(noise_d.shape), want)
    trace_d = None
    if trace:
        trace_d = torch.empty((max_new_tokens, B, width), dtype=torch.float32, device=dev)
    teach_d = None
    if teacher is not None:   # teacher forcing (include/vlg.h vlg_gpt_set_teacher): step i + 1 is fed teacher[:, i]; outputs stay the model's own
        want = (B, max_new_tokens, width) if latent else (B, max_new_tokens)
        assert tuple(teacher.shape) == want, (tuple(teacher.shape), want)
        teach_d = teacher.to(device=dev, dtype=torch.float32 if latent else torch.int32).contiguous()
    out_ids = out_lat = None
    if latent:
        out_lat = torch.empty((B, max_new_tokens, width), dtype=torch.float32, device=dev)
    else:
        out_ids = torch.empty((B, max_new_tokens), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"graph", C.c_int64(1 if model.use_graph else 0)))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"time_attn", C.c_int64(1 if model.time_attn else 0)))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"fuse_swiglu", C.c_int64(1 if model.fuse_swiglu else 0)))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"fuse_gemm", C.c_int64(1 if model.fuse_gemm else 0)))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"dl_persist", C.c_int64(int(getattr(model, "dl_persist", True)))))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"pdecode", C.c_int64(1 if getattr(model, "pdecode", True) else 0)))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"debug_pos_offset", C.c_int64(int(getattr(model, "debug_pos_offset", 0)))))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"weights_fm", C.c_int64(1 if getattr(model, "weights_fm", True) else 0)))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"act_fm", C.c_int64(1 if getattr(model, "act_fm", True) else 0)))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"pd_rows", C.c_int64(int(getattr(model, "pd_rows", 0)))))
        L.check(L.lib().vlg_gpt_set_option(model._handle, b"debug_spin_max", C.c_int64(int(getattr(model, "debug_spin_max", 0)))))
        if latent and model._head_code() == L.VLG_HEAD_HIDDEN:
            L.check(L.lib().vlg_gpt_set_option_f64(model._handle, b"cfg_iter", C.c_double(float(cfg_iter))))
        L.check(L.lib().vlg_gpt_set_teacher(model._handle, L.ptr(None if latent else teach_d), L.ptr(teach_d if latent else None)))
        try:
            L.check(L.lib().vlg_gpt_generate(model._handle, L.ptr(cond_d), L.ptr(mask_d), C.c_int32(B), C.c_int32(max_new_tokens),
                                             C.byref(sp), L.ptr(noise_d), L.ptr(out_ids), L.ptr(out_lat), L.ptr(trace_d),
                                             L.stream_ptr(dev)))
        finally:
            if teach_d is not None:
                L.check(L.lib().vlg_gpt_set_teacher(model._handle, None, None))
                teach_d.record_stream(torch.cuda.current_stream(dev))   # the enqueued steps still read it
        # The C call returns with the work enqueued (include/vlg.h), and so does this mirror: like any torch op the result tensor is
        # ordered on the current stream.  The persistent kernels bound their in-launch waits and report a wait that ran out through the
        # handle's fault word; that surfaces as VlgError(VLG_ERR_STATE) on the handle's next call or through model.status() (which the
        # sample scripts call before they write files).  model.check_faults = True waits for this call and raises at once instead.
        if getattr(model, "check_faults", False):
            L.check(L.lib().vlg_gpt_status(model._handle, C.c_int32(1)))
    return (out_lat if latent else out_ids), trace_d


@torch.no_grad()
def generate(model, cond, max_new_tokens, emb_masks=None, cfg_scale=1.0, cfg_interval=-1, noise=None, return_trace=False,
             teacher=None, **sampling_kwargs):
    """Returns int32 [B, max_new_tokens] (generate.py:168,180).  Extra (non-reference) keywords: `noise`
    ([N,B,V] Exp(1) draws so that results can be compared with the CPU oracle), `seed`, `return_trace`, `teacher` (int [B, N]: teacher
    forcing - step i + 1 is fed teacher[:, i], the returned ids / trace are still the model's own)."""
    if model.model_type not in ('c2i', 't2i'):
        raise Exception("please check model type")          # generate.py:144
    out, tr = _run(model, cond, max_new_tokens, emb_masks, cfg_scale, cfg_interval, noise, return_trace, sampling_kwargs, teacher=teacher)
    return (out, tr) if return_trace else out


@torch.no_grad()
def generate_t2v(model, cond, max_new_tokens, emb_masks=None, cfg_scale=1.0, cfg_interval=-1, return_trace=False, noise=None,
                 cfg_iter=1.0, teacher=None, **sampling_kwargs):
    """Continuous-latent generate (generate_video_diff.py:185-228): returns float [B, N, vae_embed_dim].
    head 'adapter2': token = vae_latent_adapter2(h) (gpt_video.py:431); head 'hidden': token = DiffLoss.sample(h, temperature,
    cfg_iter) (generate_video_diff.py:89-91,132-134) - `noise` [N, steps+1, B, C] N(0,1) draws make it reproducible.  cfg_iter != 1 is
    DiffLoss.sample's own guidance (diffloss.py:37-41): rows [0, B/2) conditional, [B/2, B) unconditional, B even.  `teacher` (float
    [B, N, C]): teacher forcing - step i + 1 is fed teacher[:, i]; the returned latents are still the model's own outputs."""
    if model.model_type != 't2v':
        raise Exception("please check model type")          # generate_video_diff.py:196
    if cfg_iter != 1.0 and model._head_code() != L.VLG_HEAD_HIDDEN:
        raise L.VlgError(-3, "cfg_iter is DiffLoss.sample's guidance: hidden head only")
    out, tr = _run(model, cond, max_new_tokens, emb_masks, cfg_scale, cfg_interval, noise, return_trace, sampling_kwargs, cfg_iter, teacher=teacher)
    return (out, tr) if return_trace else out
