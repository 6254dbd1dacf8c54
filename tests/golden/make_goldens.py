#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (/root/reference) on CPU.

    python tests/golden/make_goldens.py [--only gpt,sampler,vq,vae,t2v,gptb]

Only runs in the build container (needs /root/reference); outputs are small .npz
fixtures committed under tests/golden/.  The inputs/weights are regenerated from
seeds (oracle/cases.py, oracle/detweights.py), so fixtures hold expected outputs only.
torch 2.10.0 CPU kernels, default matmul precision (true fp32).
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import cases, detweights, ref_harness  # noqa: E402

torch.set_grad_enabled(False)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def build_ref_gpt(gptmod, cfg, dtype):
    args = gptmod.ModelArgs(dim=cfg["dim"], n_layer=cfg["n_layer"], n_head=cfg["n_head"],
                            vocab_size=cfg["vocab_size"], block_size=cfg["block_size"],
                            cls_token_num=cfg["cls_token_num"], model_type=cfg["model_type"],
                            num_classes=cfg["num_classes"], caption_dim=cfg["caption_dim"],
                            **({k: cfg[k] for k in ("vae_embed_dim", "num_frames", "t_downsample_size") if k in cfg}
                               if cfg["model_type"] == "t2v" else {}),
                            **({"shuffle_video_tokens": False} if cfg["model_type"] == "t2v" else {}))
    m = gptmod.Transformer(args)
    sd = detweights.gpt_weights(cfg)
    missing, unexpected = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("freqs") for k in missing), missing
    return m.to(dtype).eval()


def ref_generate_trace(genmod, model, cond, N, emb_masks, cfg_scale, cfg_interval, **kw):
    """generate() + capture of the combined logits fed to sample() at each step."""
    captured = []
    orig = genmod.sample

    def spy(logits, **k):
        captured.append(logits[:, -1, :].float().clone().numpy())
        return orig(logits, **k)

    genmod.sample = spy
    try:
        ids = genmod.generate(model, cond, N, emb_masks, cfg_scale=cfg_scale, cfg_interval=cfg_interval, **kw)
    finally:
        genmod.sample = orig
    return ids.numpy(), np.stack(captured, 0)


def gold_rope(out):
    gptmod, _ = ref_harness.load_gpt()
    gv = ref_harness.load_gpt_video()
    for g, hd, cls in ((16, 64, 1), (24, 100, 1), (4, 64, 8), (32, 64, 120)):
        tab = gptmod.precompute_freqs_cis_2d(g, hd, 10000, cls).numpy()
        out[f"rope2d_g{g}_hd{hd}_c{cls}_sum"] = np.array([tab.astype(np.float64).sum(), np.abs(tab).astype(np.float64).sum()])
        out[f"rope2d_g{g}_hd{hd}_c{cls}_rows"] = tab[[0, cls, cls + 1, cls + g, cls + g * g - 1]]
    tab = gv.precompute_freqs_cis_3d_video(4, 3, 64, 10000, 8).numpy()
    out["rope3d_g4_t3_hd64_c8"] = tab


def gold_gpt(out):
    gptmod, genmod = ref_harness.load_gpt()
    for tag, cfg in (("c2i", cases.TINY_C2I), ("t2i", cases.TINY_T2I), ("hd100", cases.TINY_HD100)):
        N = cfg["block_size"]
        for dt_name, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            m = build_ref_gpt(gptmod, cfg, dtype)
            B = 3
            if cfg["model_type"] == "c2i":
                cond, masks = t(cases.class_ids(B, cfg["num_classes"])), None
            else:
                c, mk = cases.text_cond(B, cfg["cls_token_num"], cfg["caption_dim"], lens=[120, 3, 57])
                cond, masks = t(c).to(dtype), t(mk)
            # greedy, no cfg
            ids, lg = ref_generate_trace(genmod, m, cond, N, masks, 1.0, -1, temperature=1.0, top_k=0, top_p=1.0, sample_logits=False)
            out[f"{tag}_{dt_name}_greedy_ids"] = ids
            out[f"{tag}_{dt_name}_greedy_logits"] = lg.astype(np.float32)
            # greedy + cfg + cfg_interval
            ids, lg = ref_generate_trace(genmod, m, cond, N, masks, 2.5, 6, temperature=1.0, top_k=0, top_p=1.0, sample_logits=False)
            out[f"{tag}_{dt_name}_cfg_ids"] = ids
            out[f"{tag}_{dt_name}_cfg_logits"] = lg.astype(np.float32)
            # stochastic: shared exponential noise through a patched torch.multinomial
            noise = cases.exp_noise((N, B, cfg["vocab_size"]), seed=7)
            step = [0]
            orig_mn = torch.multinomial

            margins, slacks = [], []

            def mn(probs, num_samples=1, **k):
                q = t(noise[step[0]])
                step[0] += 1
                sc = (probs / q)
                top2 = torch.topk(sc, 2, dim=-1).values
                # how decided the draw was: log(best / runner-up) of p/q, and how many kept tokens rank below the drawn one in p
                # (0 = it is the last token the top-k / top-p filter kept)
                margins.append(torch.log(top2[:, 0] / top2[:, 1].clamp_min(1e-38)).numpy())
                win = torch.argmax(sc, dim=-1, keepdim=True)
                slacks.append(((probs > 0) & (probs < probs.gather(1, win))).sum(-1).numpy())
                return win

            torch.multinomial = mn
            try:
                ids, lg = ref_generate_trace(genmod, m, cond, N, masks, 3.0, -1, temperature=0.9, top_k=50, top_p=0.95, sample_logits=True)
            finally:
                torch.multinomial = orig_mn
            out[f"{tag}_{dt_name}_sample_ids"] = ids
            out[f"{tag}_{dt_name}_sample_margin"] = np.stack(margins, 0).astype(np.float32)      # [N, B]
            out[f"{tag}_{dt_name}_sample_slack"] = np.stack(slacks, 0).astype(np.int32)
            out[f"{tag}_{dt_name}_sample_logits0"] = lg[0].astype(np.float32)


def gold_sampler(out):
    _, genmod = ref_harness.load_gpt()
    logits = cases.sampler_logits()
    q = cases.exp_noise(logits.shape, seed=13)
    grid = [(0, 1.0, 1.0), (1000, 1.0, 1.0), (2000, 1.0, 0.7), (0, 0.9, 1.0), (100, 0.8, 1.3), (1, 1.0, 1.0), (16384, 0.5, 1.0), (300, 0.3, 0.5)]
    out["sampler_grid"] = np.array(grid, np.float64)
    for gi, (k, p, temp) in enumerate(grid):
        lg = t(logits.copy())[:, None, :]
        idx, probs = genmod.sample(lg, temperature=temp, top_k=int(k), top_p=p, sample_logits=False)
        probs = probs.numpy()
        out[f"sampler_{gi}_greedy"] = idx.numpy().reshape(-1)
        out[f"sampler_{gi}_nnz"] = (probs > 0).sum(-1)
        out[f"sampler_{gi}_pmax"] = probs.max(-1)
        out[f"sampler_{gi}_psum_top"] = np.sort(probs, -1)[:, -16:].astype(np.float64).sum(-1)
        out[f"sampler_{gi}_noise_idx"] = np.argmax(probs / q, -1)
        # torch.multinomial itself under a seeded CPU generator == argmax(p / exponential_)
        g = torch.Generator().manual_seed(1234)
        e = torch.empty(probs.shape).exponential_(1, generator=g)
        g2 = torch.Generator().manual_seed(1234)
        mn = torch.multinomial(t(probs), 1, generator=g2).reshape(-1).numpy()
        assert (np.argmax(probs / e.numpy(), -1) == mn).all()
    # engineered ties for top-k (ties kept, Q5)
    tie = np.zeros((1, 64), np.float32)
    tie[0, :10] = 5.0
    tie[0, 10:20] = 4.0
    f = genmod.top_k_top_p_filtering(t(tie.copy()), top_k=12, top_p=1.0).numpy()
    out["sampler_tie_kept"] = np.isfinite(f).sum(-1)


def build_ref_vq(vqmod, sd):
    m = vqmod.VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8)
    full = m.state_dict()
    for k, v in sd.items():
        assert k in full and tuple(full[k].shape) == v.shape, k
        full[k] = t(v)
    m.load_state_dict(full)
    return m.eval()


def vq_gap(quantize, z):
    """Runner-up minus best distance of VectorQuantizer.forward's argmin (vq_model.py:215-233), from the module's own tensors with the
    module's own expression: how decided each nearest-neighbour choice of the reference was."""
    F = torch.nn.functional
    zf = F.normalize(torch.einsum('b c h w -> b h w c', z).contiguous().view(-1, quantize.e_dim), p=2, dim=-1)
    emb = F.normalize(quantize.embedding.weight, p=2, dim=-1)
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1) - 2 * torch.einsum('bd,dn->bn', zf, torch.einsum('n d -> d n', emb))
    two = torch.topk(d, 2, dim=1, largest=False).values
    return (two[:, 1] - two[:, 0]).numpy().astype(np.float32)


def gold_vq(out):
    vqmod = ref_harness.load_vq()
    sd = detweights.vq_weights()
    m = build_ref_vq(vqmod, sd)
    g = 4
    code = cases.rng(21).integers(0, 16384, size=(2, g * g)).astype(np.int64)
    q = m.quantize.get_codebook_entry(t(code), [2, 8, g, g], True)
    out["vq_entry"] = q.numpy()
    img = m.decode_code(t(code), [2, 8, g, g])
    out["vq_decode_g4"] = img.numpy()
    # unit I/O of the post_quant+conv_in+mid stage and one upsample level, for localisation
    h = m.post_quant_conv(q)
    h = m.decoder.conv_in(h)
    out["vq_conv_in"] = h.numpy()
    for blk in m.decoder.mid:
        h = blk(h)
    out["vq_mid"] = h.numpy()
    # encode side (vq_model.py:41-45): encoder -> quant_conv -> argmin, 64 px image -> 4x4 codes
    esd = detweights.vq_encoder_weights()
    full = m.state_dict()
    for k, v in esd.items():
        assert k in full and tuple(full[k].shape) == v.shape, k
        full[k] = t(v)
    m.load_state_dict(full)
    ximg = cases.rng(24).standard_normal((2, 3, 64, 64), dtype=np.float32)
    hq = m.quant_conv(m.encoder(t(ximg)))
    out["vq_encode_z"] = hq.numpy()
    _, _, (_, _, eidx) = m.quantize(hq)
    out["vq_encode_idx"] = eidx.numpy()
    out["vq_encode_gap"] = vq_gap(m.quantize, hq)
    # argmin incl. engineered ties (two identical codebook rows -> first index wins)
    z = cases.rng(22).standard_normal((2, 8, 6, 6), dtype=np.float32)
    _, _, (_, _, idx) = m.quantize(t(z))
    out["vq_argmin"] = idx.numpy()
    out["vq_argmin_gap"] = vq_gap(m.quantize, t(z))
    m2 = build_ref_vq(vqmod, sd)
    w = m2.quantize.embedding.weight.data
    w[777] = w[5]
    zz = torch.nn.functional.normalize(w[[5, 5, 9]], dim=-1).reshape(1, 3, 1, 8).permute(0, 3, 1, 2).contiguous()
    _, _, (_, _, idx2) = m2.quantize(zz)
    out["vq_argmin_tie"] = idx2.numpy()


def gold_vae(out):
    Decoder, mods = ref_harness.load_vae_decoder_cls()
    cfg = cases.TINY_VAE
    sd = detweights.vae_weights(cfg)
    dec = Decoder(z_channels=cfg["z_channels"], hidden_size=cfg["hidden_size"], hidden_size_mult=cfg["hidden_size_mult"],
                  attn_resolutions=[], conv_in="CausalConv3d", conv_out="CausalConv3d", attention="AttnBlock3D",
                  resnet_blocks=("ResnetBlock3D",) * 4, spatial_upsample=("", "SpatialUpsample2x", "SpatialUpsample2x", "SpatialUpsample2x"),
                  temporal_upsample=("", "", "TimeUpsample2x", "TimeUpsample2x"), mid_resnet="ResnetBlock3D",
                  dropout=0.0, resolution=256, num_res_blocks=cfg["num_res_blocks"]).eval()
    dsd = {k[len("decoder."):]: t(v) for k, v in sd.items() if k.startswith("decoder.")}
    dec.load_state_dict(dsd)
    pq = mods.CausalConv3d(cfg["embed_dim"], cfg["z_channels"], 1)
    pq.load_state_dict({"conv.weight": t(sd["post_quant_conv.conv.weight"]), "conv.bias": t(sd["post_quant_conv.conv.bias"])})
    z = cases.rng(31).standard_normal((1, cfg["embed_dim"], 3, 4, 4), dtype=np.float32)
    y = dec(pq(t(z)))                                   # CausalVAEModel.decode, modeling_causalvae.py:401-403
    out["vae_decode"] = y.numpy()
    z1 = cases.rng(32).standard_normal((2, cfg["embed_dim"], 1, 4, 4), dtype=np.float32)
    out["vae_decode_1f"] = dec(pq(t(z1))).numpy()
    # unit I/O: AttnBlock3D at t=3 (Q12), TimeUpsample2x, CausalConv3d k3
    x = cases.rng(33).standard_normal((1, 128, 3, 4, 4), dtype=np.float32)
    out["vae_attn_t3"] = dec.mid.attn_1(t(x)).numpy()
    x2 = cases.rng(34).standard_normal((1, 4, 5, 2, 2), dtype=np.float32)
    out["vae_timeup"] = mods.TimeUpsample2x(4, 4)(t(x2)).numpy()
    out["vae_conv_in"] = dec.conv_in(pq(t(z))).numpy()
    # encode side: Encoder + quant_conv -> posterior parameters (modeling_causalvae.py:382-392), 9 frames x 32 x 32 -> [.,16,3,4,4]
    esd = detweights.vae_encoder_weights(cfg)
    enc = mods.RefEncoder(z_channels=cfg["z_channels"], hidden_size=cfg["hidden_size"], hidden_size_mult=cfg["hidden_size_mult"],
                          attn_resolutions=[], conv_in="CausalConv3d", conv_out="CausalConv3d", attention="AttnBlock3D",
                          resnet_blocks=("ResnetBlock3D",) * 4, spatial_downsample=("SpatialDownsample2x",) * 3 + ("",),
                          temporal_downsample=("", "TimeDownsample2x", "TimeDownsample2x", ""), mid_resnet="ResnetBlock3D", dropout=0.0,
                          resolution=256, num_res_blocks=cfg["num_res_blocks"], double_z=True).eval()
    enc.load_state_dict({k[len("encoder."):]: t(v) for k, v in esd.items() if k.startswith("encoder.")})
    qc = mods.CausalConv3d(2 * cfg["z_channels"], 2 * cfg["embed_dim"], 1)
    qc.load_state_dict({"conv.weight": t(esd["quant_conv.conv.weight"]), "conv.bias": t(esd["quant_conv.conv.bias"])})
    xv = cases.rng(36).standard_normal((1, 3, 9, 32, 32), dtype=np.float32)
    out["vae_moments"] = qc(enc(t(xv))).numpy()
    # tiled decode (modeling_causalvae.py:468-570) with toy tile sizes so that a [1,C,5,6,6] latent tiles in t, h and w
    Stub = ref_harness.load_vae_tiling_methods()
    st = Stub()
    st.decoder, st.post_quant_conv, st.use_quant_layer = dec, pq, True
    st.tile_sample_min_size, st.tile_latent_min_size, st.tile_latent_min_size_t, st.tile_overlap_factor = 32, 4, 3, 0.25
    zt = cases.rng(35).standard_normal((1, cfg["embed_dim"], 5, 6, 6), dtype=np.float32)
    out["vae_tiled"] = st.tiled_decode(t(zt)).numpy()
    # tiled encode (modeling_causalvae.py:444-466,491-530): [1,3,9,48,48] tiles in t (chunks of 5 frames, one shared) and in h, w (32-px
    # tiles at stride 24, 1 latent row/column blended)
    st.encoder, st.quant_conv = enc, qc
    st.tile_sample_min_size, st.tile_latent_min_size, st.tile_sample_min_size_t, st.tile_overlap_factor = 32, 4, 5, 0.25
    xt = cases.rng(37).standard_normal((1, 3, 9, 48, 48), dtype=np.float32)
    out["vae_tiled_moments"] = st.tiled_encode(t(xt)).numpy()


def gold_vaeunits(out):
    """Per-op I/O of the CausalVideoVAE building blocks from the reference's own modules, plus one full-width
    single-frame decode (constructor defaults, 512/256/128 channels, [1,8,1,32,32] -> [1,3,1,256,256]) as statistics + crops."""
    Decoder, mods = ref_harness.load_vae_decoder_cls()
    import importlib
    rb = importlib.import_module("causalvideovae.model.modules.resnet_block")
    r = cases.rng

    def w(seed, shape, std):
        return (r(seed).standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    # CausalConv3d k3 (conv.py:76-130)
    cc = mods.CausalConv3d(32, 64, 3, padding=1)
    cc.load_state_dict({"conv.weight": t(w(82, (64, 32, 3, 3, 3), 0.05)), "conv.bias": t(w(83, (64,), 0.1))})
    out["unit_conv_k3"] = cc(t(w(81, (1, 32, 3, 6, 6), 1.0))).numpy()
    # SpatialDownsample2x: zero pad (0,1) + CausalConv3d (1,3,3) stride (1,2,2) (updownsample.py:63-93)
    dn = mods.SpatialDownsample2x(32, 64)
    dn.load_state_dict({"conv.conv.weight": t(w(85, (64, 32, 1, 3, 3), 0.08)), "conv.conv.bias": t(w(86, (64,), 0.1))})
    out["unit_down"] = dn(t(w(84, (1, 32, 3, 8, 8), 1.0))).numpy()
    # SpatialUpsample2x: nearest x2 + CausalConv3d (1,3,3) (updownsample.py:124-153)
    up = mods.SpatialUpsample2x(32, 64)
    up.load_state_dict({"conv.conv.weight": t(w(88, (64, 32, 1, 3, 3), 0.08)), "conv.conv.bias": t(w(89, (64,), 0.1))})
    out["unit_up"] = up(t(w(87, (1, 32, 2, 4, 4), 1.0))).numpy()
    # Normalize + swish (normalize.py:14-17, ops.py:14-15)
    gn = mods.Normalize(64)
    gn.load_state_dict({"weight": t(1 + w(91, (64,), 0.1)), "bias": t(w(92, (64,), 0.05))})
    xg = t(w(90, (2, 64, 3, 4, 4), 1.5))
    out["unit_gn"] = gn(xg).numpy()
    out["unit_gn_swish"] = mods.nonlinearity(gn(xg)).numpy() if hasattr(mods, "nonlinearity") else (gn(xg) * torch.sigmoid(gn(xg))).numpy()
    # ResnetBlock3D 32 -> 64 with the 1x1x1 shortcut (resnet_block.py:140-172)
    res = rb.ResnetBlock3D(in_channels=32, out_channels=64, dropout=0.0).eval()
    res.load_state_dict({"norm1.weight": t(1 + w(94, (32,), 0.1)), "norm1.bias": t(w(95, (32,), 0.05)),
                         "conv1.conv.weight": t(w(96, (64, 32, 3, 3, 3), 0.04)), "conv1.conv.bias": t(w(97, (64,), 0.05)),
                         "norm2.weight": t(1 + w(98, (64,), 0.1)), "norm2.bias": t(w(99, (64,), 0.05)),
                         "conv2.conv.weight": t(w(100, (64, 64, 3, 3, 3), 0.03)), "conv2.conv.bias": t(w(101, (64,), 0.05)),
                         "nin_shortcut.conv.weight": t(w(102, (64, 32, 1, 1, 1), 0.2)), "nin_shortcut.conv.bias": t(w(103, (64,), 0.05))})
    out["unit_res"] = res(t(w(93, (1, 32, 3, 4, 4), 1.0))).numpy()
    # full-width decoder, one frame
    cfg = dict(hidden_size=128, z_channels=4, embed_dim=8, hidden_size_mult=(1, 2, 4, 4), num_res_blocks=2)
    sd = detweights.vae_weights(cfg)
    dec = Decoder(z_channels=4, hidden_size=128, hidden_size_mult=(1, 2, 4, 4), attn_resolutions=[], conv_in="CausalConv3d",
                  conv_out="CausalConv3d", attention="AttnBlock3D", resnet_blocks=("ResnetBlock3D",) * 4,
                  spatial_upsample=("", "SpatialUpsample2x", "SpatialUpsample2x", "SpatialUpsample2x"),
                  temporal_upsample=("", "", "TimeUpsample2x", "TimeUpsample2x"), mid_resnet="ResnetBlock3D", dropout=0.0, resolution=256,
                  num_res_blocks=2).eval()
    dec.load_state_dict({k[len("decoder."):]: t(v) for k, v in sd.items() if k.startswith("decoder.")})
    pq = mods.CausalConv3d(8, 4, 1)
    pq.load_state_dict({"conv.weight": t(sd["post_quant_conv.conv.weight"]), "conv.bias": t(sd["post_quant_conv.conv.bias"])})
    z = cases.rng(38).standard_normal((1, 8, 1, 32, 32), dtype=np.float32)
    y = dec(pq(t(z))).numpy()
    assert y.shape == (1, 3, 1, 256, 256)
    out["full1f_stats"] = np.array([y.astype(np.float64).sum(), np.abs(y).astype(np.float64).sum(), y.min(), y.max()], np.float64)
    out["full1f_crop"] = y[0, :, 0, 100:132, 100:132].copy()
    out["full1f_grid"] = y[0, :, 0, ::8, ::8].copy()


def gold_codebook(out):
    """Codebook.forward in eval mode from the reference's own class (CausalVideoVAE/causalvideovae/model/modules/quant.py:8-99, the
    byte-identical sibling of tokenizer_video/vqvae.py:130-213): default size 2048 x 256 on a [2,256,2,4,4] latent."""
    ref_harness.load_vae_modules()
    import importlib
    quant = importlib.import_module("causalvideovae.model.modules.quant")
    r = cases.rng(41)
    E = r.standard_normal((2048, 256), dtype=np.float32)
    z = r.standard_normal((2, 256, 2, 4, 4), dtype=np.float32)
    cb = quant.Codebook(2048, 256).eval()
    cb.embeddings.data.copy_(t(E))
    cb._need_init = False
    res = cb(t(z))
    out["cb_encodings"] = res["encodings"].numpy()
    out["cb_embeddings"] = res["embeddings"].numpy()
    out["cb_commitment_loss"] = np.float32(res["commitment_loss"].item())
    out["cb_perplexity"] = np.float32(res["perplexity"].item())
    flat = t(z).permute(0, 2, 3, 4, 1).flatten(end_dim=-2)
    d = (flat ** 2).sum(dim=1, keepdim=True) - 2 * flat @ cb.embeddings.t() + (cb.embeddings.t() ** 2).sum(dim=0, keepdim=True)
    two = torch.topk(d, 2, dim=1, largest=False).values
    out["cb_gap"] = (two[:, 1] - two[:, 0]).numpy().astype(np.float32)
    # a latent made of codebook rows plus small noise: every position has a clear winner, many codes unused (perplexity << n_codes)
    ids = r.integers(0, 64, size=(1, 3, 4, 4))
    z2 = (np.moveaxis(E[ids], -1, 1) + 0.01 * r.standard_normal((1, 256, 3, 4, 4), dtype=np.float32)).astype(np.float32)
    res2 = cb(t(z2))
    assert (res2["encodings"].numpy() == ids).all()
    out["cb2_commitment_loss"] = np.float32(res2["commitment_loss"].item())
    out["cb2_perplexity"] = np.float32(res2["perplexity"].item())


def gold_t2v(out):
    gv = ref_harness.load_gpt_video()
    cfg = cases.TINY_T2V
    vae_t = (cfg["num_frames"] - 1) // cfg["t_downsample_size"] + 1
    N = vae_t * cfg["block_size"]
    for dt_name, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = build_ref_gpt(gv, cfg, dtype)
        B = 2
        c, mk = cases.text_cond(B, cfg["cls_token_num"], cfg["caption_dim"], lens=[8, 4])
        # batched prefill + decode with the reference Transformer (cfg=1), mask fix-up as generate.py:156-165
        T = cfg["cls_token_num"]
        m.setup_caches(B, T + N, dtype)
        m.causal_mask[:, :, :T] = m.causal_mask[:, :, :T] * t(mk).bool().unsqueeze(1)
        eye = torch.eye(m.causal_mask.size(1), m.causal_mask.size(2))
        m.causal_mask[:] = (m.causal_mask * (1 - eye) + eye).bool()
        outs = []
        h, _ = m(cond_embed=t(c).to(dtype), video_latent=None, input_pos=torch.arange(T))
        e = h[:, -1:, :]
        outs.append(e.float().numpy())
        for i in range(N - 1):
            h, _ = m(cond_embed=None, video_latent=e, input_pos=torch.tensor([T + i]))
            e = h[:, -1:, :]
            outs.append(e.float().numpy())
        out[f"t2v_{dt_name}_latents"] = np.concatenate(outs, 1)


def gold_t2vdiff(out):
    """gpt_video_diff + DiffLoss head through the reference's own generate_video_diff.generate (B = 1, cfg 1: the only
    mode the shipped code supports), 10 sampling steps, fixed noise fed through patched torch.randn / randn_like."""
    gvd, gen = ref_harness.load_gpt_video_diff()
    cfg = cases.TINY_T2V_DIFF
    vae_t = (cfg["num_frames"] - 1) // cfg["t_downsample_size"] + 1
    N = 6
    S = cfg["num_sampling_steps"]
    sd = detweights.gpt_weights(cfg)
    args = gvd.ModelArgs(dim=cfg["dim"], n_layer=cfg["n_layer"], n_head=cfg["n_head"], vocab_size=cfg["vocab_size"],
                         block_size=cfg["block_size"], cls_token_num=cfg["cls_token_num"], model_type="t2v", caption_dim=cfg["caption_dim"],
                         vae_embed_dim=cfg["vae_embed_dim"], num_frames=cfg["num_frames"], t_downsample_size=cfg["t_downsample_size"],
                         diffloss_d=cfg["diffloss_d"], diffloss_w=cfg["diffloss_w"], num_sampling_steps=str(S))
    m = gvd.Transformer(args)
    missing, unexpected = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("freqs") or k in ("mask_token", "output.weight", "tok_embeddings.weight") for k in missing), missing
    m = m.eval()
    C = cfg["vae_embed_dim"]
    noise = cases.rng(51).standard_normal((N, S + 1, 1, C), dtype=np.float32)
    c, mk = cases.text_cond(1, cfg["cls_token_num"], cfg["caption_dim"], lens=[5])
    calls = [0]
    orig_randn, orig_like, orig_cuda = torch.randn, torch.randn_like, torch.Tensor.cuda

    def nxt(shape):
        tok, k = divmod(calls[0], S + 1)
        calls[0] += 1
        a = t(noise[tok, k])
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
        return a.clone()

    torch.randn = lambda *shape, **kw: nxt(shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape)
    torch.randn_like = lambda x, **kw: nxt(x.shape)
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        lat = gen.generate(m, t(c), N, t(mk), cfg_scale=1.0, temperature=0.9, cfg_iter=1.0)
    finally:
        torch.randn, torch.randn_like, torch.Tensor.cuda = orig_randn, orig_like, orig_cuda
    assert calls[0] == N * (S + 1)
    out["t2vdiff_latents"] = lat.float().numpy()
    # schedule + one network evaluation for localisation
    gd = m.diffloss.gen_diffusion
    out["sched_timestep_map"] = np.array(gd.timestep_map, np.int64)
    out["sched_sqrt_recip"] = gd.sqrt_recip_alphas_cumprod
    out["sched_coef1"] = gd.posterior_mean_coef1
    out["sched_min_log"] = gd.posterior_log_variance_clipped
    out["sched_max_log"] = np.log(gd.betas)
    x = t(cases.rng(52).standard_normal((3, C), dtype=np.float32))
    z = t(cases.rng(53).standard_normal((3, cfg["dim"]), dtype=np.float32))
    out["net_out"] = m.diffloss.net(x, torch.tensor([999, 444, 0]), z).numpy()
    # DiffLoss.sample with guidance inside the sampler (diffloss.py:37-41 -> forward_with_cfg :240-248) in isolation: z rows [cond | uncond],
    # one x_T draw per pair (torch.randn(n, C)), then one torch.randn_like(x) per reverse step for all 2n rows
    n_pair = 2
    zc = cases.rng(54).standard_normal((2 * n_pair, cfg["dim"]), dtype=np.float32)
    nz = cases.rng(56).standard_normal((S + 1, 2 * n_pair, C), dtype=np.float32)
    calls2 = [0]

    def nxt2(shape):
        k = calls2[0]
        calls2[0] += 1
        a = t(nz[k][:n_pair] if k == 0 else nz[k])
        assert tuple(a.shape) == tuple(shape), (a.shape, shape, k)
        return a.clone()

    torch.randn = lambda *shape, **kw: nxt2(shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape)
    torch.randn_like = lambda x, **kw: nxt2(x.shape)
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            out["dlcfg_latents"] = m.diffloss.sample(t(zc), 0.9, 2.5).float().numpy()
    finally:
        torch.randn, torch.randn_like, torch.Tensor.cuda = orig_randn, orig_like, orig_cuda
    assert calls2[0] == S + 1
    gd100 = gvd.DiffLoss(target_channels=8, z_channels=16, depth=1, width=16, num_sampling_steps="100").gen_diffusion
    out["sched100_timestep_map"] = np.array(gd100.timestep_map, np.int64)
    out["sched100_coef2"] = gd100.posterior_mean_coef2


def gold_videovq(out):
    """tokenizer_video VQVAE.decode (vqvae.py:48-51): embedding -> post_vq_conv -> Decoder, from the reference's own classes."""
    ns = ref_harness.load_videovq_classes()
    cfg = cases.TINY_VIDEOVQ
    sd = detweights.videovq_weights(cfg)
    nh = cfg["n_hiddens"]
    dec = ns["Decoder"](nh, cfg["n_res_layers"], (4, 4, 4)).eval()
    dsd = {k[len("decoder."):]: t(v) for k, v in sd.items() if k.startswith("decoder.")}
    full = dec.state_dict()
    for k, v in dsd.items():
        assert k in full and tuple(full[k].shape) == tuple(v.shape), (k, full.get(k, torch.zeros(0)).shape, v.shape)
        full[k] = v
    dec.load_state_dict(full)
    pq = ns["SamePadConv3d"](cfg["embedding_dim"], nh, 1)
    pq.load_state_dict({"conv.weight": t(sd["post_vq_conv.conv.weight"]), "conv.bias": t(sd["post_vq_conv.conv.bias"])})
    enc = cases.rng(61).integers(0, cfg["n_codes"], size=(2, 2, 3, 4)).astype(np.int64)
    h = torch.nn.functional.embedding(t(enc), t(sd["codebook.embeddings"]))
    h = pq(ns["shift_dim"](h, -1, 1))
    out["videovq_post"] = h.numpy()
    out["videovq_res0"] = dec.res_stack[0](h).numpy()
    out["videovq_decode"] = dec(h).numpy()


def gold_gptb(out):
    """BASELINE config 1: GPT-B c2i 16x16 greedy fp32, B=1: ids + top1-top2 margins."""
    gptmod, genmod = ref_harness.load_gpt()
    m = build_ref_gpt(gptmod, cases.GPT_B, torch.float32)
    cond = t(cases.class_ids(1, 1000, seed=0))
    ids, lg = ref_generate_trace(genmod, m, cond, 256, None, 1.0, -1, temperature=1.0, top_k=0, top_p=1.0, sample_logits=False)
    srt = np.sort(lg[:, 0, :], -1)
    out["gptb_ids"] = ids
    out["gptb_margin"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
    out["gptb_logits_step0"] = lg[0, 0].astype(np.float32)
    out["gptb_top1"] = srt[:, -1].astype(np.float32)


PARTS = dict(rope=gold_rope, gpt=gold_gpt, sampler=gold_sampler, vq=gold_vq, vae=gold_vae, t2v=gold_t2v, gptb=gold_gptb, t2vdiff=gold_t2vdiff, videovq=gold_videovq,
             vaeunits=gold_vaeunits, codebook=gold_codebook)

def gold_t5(out):
    """transformers.T5EncoderModel (the class language/t5.py:60 loads) on CPU with the build's deterministic weights: fp32 and bf16,
    12 tokens, one fully valid row and one padded to 5 valid tokens (padding='max_length' convention, language/t5.py:66-74)."""
    from transformers import T5Config, T5EncoderModel
    cfg = cases.TINY_T5
    sd = detweights.t5_weights(cfg)
    ids = cases.rng(71).integers(0, cfg["vocab_size"], size=(2, 12)).astype(np.int64)
    mask = np.ones((2, 12), np.int64)
    mask[1, 5:] = 0
    ids[1, 5:] = 0
    out["t5_ids"], out["t5_mask"] = ids, mask
    for dt_name, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = T5EncoderModel(T5Config(**cfg)).eval()
        missing, unexpected = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.endswith("embed_tokens.weight") for k in missing), (missing, unexpected)
        m = m.to(dtype)
        with torch.no_grad():
            y = m(input_ids=t(ids), attention_mask=t(mask))["last_hidden_state"]
        out[f"t5_{dt_name}"] = y.float().numpy()


PARTS["t5"] = gold_t5


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(PARTS))
    a = ap.parse_args()
    for name in a.only.split(","):
        out = {}
        PARTS[name](out)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, "->", path, os.path.getsize(path) // 1024, "KB", flush=True)
