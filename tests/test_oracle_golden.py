"""Pins the numpy oracle (oracle/vlg_oracle.py) against golden vectors produced by the real
reference on CPU (tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle import cases, detweights
from oracle import vlg_oracle as O


def test_rope_tables(golden):
    g = golden("rope")
    for gr, hd, cls in ((16, 64, 1), (24, 100, 1), (4, 64, 8), (32, 64, 120)):
        tab = O.rope_table_2d(gr, hd, 10000.0, cls)
        assert tab.shape == (cls + gr * gr, hd // 2, 2)
        assert not tab[:cls].any()                                  # Q1: zero rows for condition positions
        rows = tab[[0, cls, cls + 1, cls + gr, cls + gr * gr - 1]]
        np.testing.assert_allclose(rows, g[f"rope2d_g{gr}_hd{hd}_c{cls}_rows"], atol=2e-6)
        s = g[f"rope2d_g{gr}_hd{hd}_c{cls}_sum"]
        assert abs(tab.astype(np.float64).sum() - s[0]) < 1e-2
    np.testing.assert_allclose(O.rope_table_3d(4, 3, 64, 10000.0, 8), g["rope3d_g4_t3_hd64_c8"], atol=2e-6)


def _inputs(cfg, B=3):
    if cfg["model_type"] == "c2i":
        return cases.class_ids(B, cfg["num_classes"]), None
    return cases.text_cond(B, cfg["cls_token_num"], cfg["caption_dim"], lens=[120, 3, 57])


@pytest.mark.parametrize("tag,cfg", [("c2i", cases.TINY_C2I), ("t2i", cases.TINY_T2I), ("hd100", cases.TINY_HD100)])
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_gpt_generate(golden, tag, cfg, dt):
    g = golden("gpt")
    sd = detweights.gpt_weights(cfg)
    m = O.GPTOracle(cfg, sd, dt)
    cond, masks = _inputs(cfg)
    N = cfg["block_size"]
    tol = 2e-4 if dt == "fp32" else 6e-2
    for name, kw in (("greedy", dict(cfg_scale=1.0, cfg_interval=-1)), ("cfg", dict(cfg_scale=2.5, cfg_interval=6))):
        tr = {}
        ids = O.generate(m, cond, N, masks, sample_logits=False, trace=tr, **kw)
        ref_ids = g[f"{tag}_{dt}_{name}_ids"]
        ref_lg = g[f"{tag}_{dt}_{name}_logits"]
        lg = np.stack(tr["logits"], 0)
        if dt == "fp32":
            assert (ids == ref_ids).all()
            np.testing.assert_allclose(lg, ref_lg, atol=tol, rtol=1e-4)
        else:
            # bf16: trajectories may fork at a near-tie; logits compared up to the first fork
            same = (ids == ref_ids).all(axis=0)
            upto = N if same.all() else int(np.argmin(same)) + 1
            err = np.abs(lg[:upto] - ref_lg[:upto]).max()
            assert err < tol * max(1.0, np.abs(ref_lg).max()), err
            assert upto >= 2
    # stochastic with shared exponential noise, CFG 3.0, top-k 50, top-p 0.95, T 0.9
    noise = cases.exp_noise((N, 3, cfg["vocab_size"]), seed=7)
    ids = O.generate(m, cond, N, masks, cfg_scale=3.0, temperature=0.9, top_k=50, top_p=0.95, sample_logits=True, noise=noise)
    ref_ids = g[f"{tag}_{dt}_sample_ids"]
    if dt == "fp32":
        assert (ids == ref_ids).mean() > 0.98                       # near-tie flips in filtered sets are legal
    else:
        assert (ids[:, 0] == ref_ids[:, 0]).all()


def test_sampler_grid(golden):
    g = golden("sampler")
    logits = cases.sampler_logits()
    q = cases.exp_noise(logits.shape, seed=13)
    for gi, (k, p, temp) in enumerate(g["sampler_grid"]):
        idx, probs = O.sample(logits, temperature=float(temp), top_k=int(k), top_p=float(p), sample_logits=False)
        assert (idx == g[f"sampler_{gi}_greedy"]).all()
        nnz = (probs > 0).sum(-1)
        assert (np.abs(nnz - g[f"sampler_{gi}_nnz"]) <= 1).all(), (gi, nnz, g[f"sampler_{gi}_nnz"])
        np.testing.assert_allclose(probs.max(-1), g[f"sampler_{gi}_pmax"], rtol=1e-5)
        idx2, _ = O.sample(logits, temperature=float(temp), top_k=int(k), top_p=float(p), sample_logits=True, q=q)
        assert (idx2 == g[f"sampler_{gi}_noise_idx"]).all()
    tie = np.zeros((1, 64), np.float32)
    tie[0, :10] = 5.0
    tie[0, 10:20] = 4.0
    f = O.top_k_top_p_filtering(tie, top_k=12)
    assert np.isfinite(f).sum() == g["sampler_tie_kept"][0] == 20   # ties kept (Q5)


def test_gpt_b_config1(golden):
    """BASELINE config 1: GPT-B c2i 256 tokens greedy fp32 B=1."""
    g = golden("gptb")
    m = O.GPTOracle(cases.GPT_B, detweights.gpt_weights(cases.GPT_B), "fp32")
    tr = {}
    ids = O.generate(m, cases.class_ids(1, 1000, seed=0), 256, None, sample_logits=False, trace=tr)
    lg = np.stack(tr["logits"], 0)[:, 0]
    np.testing.assert_allclose(lg[0], g["gptb_logits_step0"], atol=2e-4)
    safe = g["gptb_margin"] > 1e-3
    same = ids[0] == g["gptb_ids"][0]
    assert same.all() or same[: int(np.argmin(same))].all()
    assert same[safe].all() or not same.all()
    assert same.mean() > 0.99


def test_vq_decode_and_argmin(golden):
    g = golden("vq")
    sd = detweights.vq_weights()
    vq = O.VQOracle(sd)
    code = cases.rng(21).integers(0, 16384, size=(2, 16)).astype(np.int64)
    q = vq.get_codebook_entry(code, [2, 8, 4, 4])
    np.testing.assert_allclose(q, g["vq_entry"], atol=1e-6)
    img = vq.decode_code(code, [2, 8, 4, 4])
    assert img.shape == (2, 3, 64, 64)
    np.testing.assert_allclose(img, g["vq_decode_g4"], atol=2e-3 * np.abs(g["vq_decode_g4"]).max())
    z = cases.rng(22).standard_normal((2, 8, 6, 6), dtype=np.float32)
    idx, _ = vq.argmin(z)
    assert (idx == g["vq_argmin"]).mean() > 0.98
    sd2 = dict(sd)
    E = sd["quantize.embedding.weight"].copy()
    E[777] = E[5]
    sd2["quantize.embedding.weight"] = E
    vq2 = O.VQOracle(sd2)
    zz = O.l2norm_rows(E[[5, 5, 9]]).reshape(1, 3, 1, 8).transpose(0, 3, 1, 2)
    idx2, _ = vq2.argmin(zz)
    assert (idx2 == g["vq_argmin_tie"]).all() and idx2[0] == 5     # first minimum (Q11)


def test_vae_decode(golden):
    g = golden("vae")
    cfg = cases.TINY_VAE
    sd = detweights.vae_weights(cfg)
    vae = O.VAEOracle(sd, hidden_size=cfg["hidden_size"], hidden_size_mult=cfg["hidden_size_mult"], num_res_blocks=cfg["num_res_blocks"])
    x2 = cases.rng(34).standard_normal((1, 4, 5, 2, 2), dtype=np.float32)
    np.testing.assert_allclose(O.time_upsample2x(x2), g["vae_timeup"], atol=1e-6)    # Q13: 5 -> 9 frames
    x = cases.rng(33).standard_normal((1, 128, 3, 4, 4), dtype=np.float32)
    np.testing.assert_allclose(vae._attn("decoder.mid.attn_1", x), g["vae_attn_t3"], atol=2e-4)   # Q12
    z = cases.rng(31).standard_normal((1, cfg["embed_dim"], 3, 4, 4), dtype=np.float32)
    y = vae.decode(z)
    assert y.shape == (1, 3, 9, 32, 32)
    np.testing.assert_allclose(y, g["vae_decode"], atol=2e-3 * np.abs(g["vae_decode"]).max())
    z1 = cases.rng(32).standard_normal((2, cfg["embed_dim"], 1, 4, 4), dtype=np.float32)
    np.testing.assert_allclose(vae.decode(z1), g["vae_decode_1f"], atol=2e-3 * np.abs(g["vae_decode_1f"]).max())


def test_vae_tiled_decode(golden):
    """CausalVAEModel.tiled_decode (modeling_causalvae.py:468-570) with toy tile sizes: temporal chunks + blended spatial tiles."""
    g = golden("vae")
    cfg = cases.TINY_VAE
    vae = O.VAEOracle(detweights.vae_weights(cfg), hidden_size=cfg["hidden_size"], hidden_size_mult=cfg["hidden_size_mult"],
                      num_res_blocks=cfg["num_res_blocks"])
    zt = cases.rng(35).standard_normal((1, cfg["embed_dim"], 5, 6, 6), dtype=np.float32)
    y = O.vae_tiled_decode(vae.decode, zt, 4, 3, 32, 0.25)
    ref = g["vae_tiled"]
    assert y.shape == ref.shape == (1, 3, 17, 48, 48)
    np.testing.assert_allclose(y, ref, atol=2e-3 * np.abs(ref).max())


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_t2v_adapter2(golden, dt):
    g = golden("t2v")
    cfg = cases.TINY_T2V
    m = O.GPTOracle(cfg, detweights.gpt_weights(cfg), dt)
    c, mk = cases.text_cond(2, cfg["cls_token_num"], cfg["caption_dim"], lens=[8, 4])
    N = 3 * cfg["block_size"]
    lat = O.generate_t2v(m, c, N, mk)
    ref = g[f"t2v_{dt}_latents"]
    assert lat.shape == ref.shape == (2, N, cfg["vae_embed_dim"])
    tol = 2e-4 if dt == "fp32" else 8e-2
    assert np.abs(lat - ref).max() < tol * max(1.0, np.abs(ref).max())


def test_diffloss_head(golden):
    """gpt_video_diff + DiffLoss.sample through the reference's generate_video_diff.generate (B = 1, 10 sampling steps)."""
    g = golden("t2vdiff")
    sc = O.diffusion_schedule(10)
    assert (sc["timestep_map"] == g["sched_timestep_map"]).all()
    np.testing.assert_allclose(sc["sqrt_recip"], g["sched_sqrt_recip"], rtol=1e-12)
    np.testing.assert_allclose(sc["coef1"], g["sched_coef1"], rtol=1e-12)
    np.testing.assert_allclose(sc["min_log"], g["sched_min_log"], rtol=1e-12)
    np.testing.assert_allclose(sc["max_log"], g["sched_max_log"], rtol=1e-12)
    s100 = O.diffusion_schedule(100)
    assert (s100["timestep_map"] == g["sched100_timestep_map"]).all()
    np.testing.assert_allclose(s100["coef2"], g["sched100_coef2"], rtol=1e-12)
    cfg = cases.TINY_T2V_DIFF
    sd = detweights.gpt_weights(cfg)
    head = O.DiffLossOracle(sd, num_sampling_steps=10)
    C = cfg["vae_embed_dim"]
    x = cases.rng(52).standard_normal((3, C), dtype=np.float32)
    z = cases.rng(53).standard_normal((3, cfg["dim"]), dtype=np.float32)
    np.testing.assert_allclose(head.net(x, np.array([999, 444, 0]), z), g["net_out"], atol=2e-5, rtol=1e-4)
    m = O.GPTOracle(cfg, sd, "fp32")
    N, S = 6, 10
    noise = cases.rng(51).standard_normal((N, S + 1, 1, C), dtype=np.float32)
    c, mk = cases.text_cond(1, cfg["cls_token_num"], cfg["caption_dim"], lens=[5])
    lat = O.generate_t2v_diff(m, head, c, N, mk, noise, temperature=0.9)
    ref = g["t2vdiff_latents"]
    assert lat.shape == ref.shape == (1, N, C)
    # guidance inside the sampler: the reference's DiffLoss.sample(z, 0.9, cfg=2.5) in isolation on 2 (cond, uncond) pairs
    zc = cases.rng(54).standard_normal((4, cfg["dim"]), dtype=np.float32)
    nz = cases.rng(56).standard_normal((S + 1, 4, C), dtype=np.float32)
    refc = g["dlcfg_latents"]
    outc = head.sample(zc, nz, 0.9, 2.5)
    assert np.abs(outc - refc).max() < 1e-5 * np.abs(refc).max()
    assert np.abs(head.sample(zc, nz, 0.9, 1.0) - refc).max() > 1e-2 * np.abs(refc).max()      # the guidance is not a no-op on this case
    assert np.abs(lat - ref).max() < 5e-4 * max(1.0, np.abs(ref).max())


def test_videovq_decode(golden):
    """tokenizer_video VQVAE.decode topology (BatchNorm eval, same-pad convs, axial attention, transposed 4^3 convs)."""
    g = golden("videovq")
    cfg = cases.TINY_VIDEOVQ
    sd = detweights.videovq_weights(cfg)
    vq = O.VideoVQVAEOracle(sd, n_res_layers=cfg["n_res_layers"])
    enc = cases.rng(61).integers(0, cfg["n_codes"], size=(2, 2, 3, 4)).astype(np.int64)
    h = np.moveaxis(sd["codebook.embeddings"][enc], -1, 1)
    h = O.same_pad_conv3d(h, sd["post_vq_conv.conv.weight"], sd["post_vq_conv.conv.bias"])
    np.testing.assert_allclose(h, g["videovq_post"], atol=1e-5)
    np.testing.assert_allclose(vq._res("decoder.res_stack.0", h), g["videovq_res0"], atol=2e-4)
    y = vq.decode(enc)
    assert y.shape == g["videovq_decode"].shape == (2, 3, 8, 12, 16)
    np.testing.assert_allclose(y, g["videovq_decode"], atol=2e-3 * np.abs(g["videovq_decode"]).max())


def test_encoders(golden):
    """encode side: VQ Encoder -> quant_conv -> argmin; CausalVAE Encoder -> quant_conv -> posterior moments."""
    g = golden("vq")
    sd = dict(detweights.vq_weights())
    sd.update(detweights.vq_encoder_weights())
    vq = O.VQOracle(sd)
    ximg = cases.rng(24).standard_normal((2, 3, 64, 64), dtype=np.float32)
    idx, z = vq.encode(ximg)
    np.testing.assert_allclose(z, g["vq_encode_z"], atol=2e-3 * np.abs(g["vq_encode_z"]).max())
    assert (idx == g["vq_encode_idx"]).mean() > 0.9
    gv = golden("vae")
    cfg = cases.TINY_VAE
    vsd = dict(detweights.vae_weights(cfg))
    vsd.update(detweights.vae_encoder_weights(cfg))
    vae = O.VAEOracle(vsd, hidden_size=cfg["hidden_size"], hidden_size_mult=cfg["hidden_size_mult"], num_res_blocks=cfg["num_res_blocks"])
    xv = cases.rng(36).standard_normal((1, 3, 9, 32, 32), dtype=np.float32)
    mo = vae.encode_moments(xv)
    assert mo.shape == gv["vae_moments"].shape == (1, 16, 3, 4, 4)
    np.testing.assert_allclose(mo, gv["vae_moments"], atol=2e-3 * np.abs(gv["vae_moments"]).max())
    # tiled_encode (modeling_causalvae.py:444-466,491-530): [1,3,9,48,48] with 32-px tiles, 5-frame chunks, overlap 0.25
    xt = cases.rng(37).standard_normal((1, 3, 9, 48, 48), dtype=np.float32)
    mt = O.vae_tiled_encode(vae.encode_moments, xt, 32, 5, 4, 0.25)
    ref = gv["vae_tiled_moments"]
    assert mt.shape == ref.shape == (1, 16, 3, 6, 6)
    np.testing.assert_allclose(mt, ref, atol=2e-3 * np.abs(ref).max())


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_t5_encoder(golden, dt):
    """Text-conditioning step (language/t5.py:60-81): oracle vs transformers.T5EncoderModel on the same weights / ids / padding mask."""
    g = golden("t5")
    cfg = cases.TINY_T5
    y = O.T5Oracle(cfg, detweights.t5_weights(cfg), dt).encode(g["t5_ids"], g["t5_mask"])
    ref = g[f"t5_{dt}"]
    assert y.shape == ref.shape == (2, 12, cfg["d_model"])
    valid = g["t5_mask"].astype(bool)                      # padded query rows are never consumed downstream (emb_masks zero them)
    tol = 2e-4 if dt == "fp32" else 6e-2
    assert np.abs(y - ref)[valid].max() < tol * max(1.0, np.abs(ref).max())
    # bucket function against hand-checked values (bidirectional, 32 buckets, max distance 128)
    assert O.t5_relative_bucket(np.array([0, 1, -1, 7, 8, -8, 127, 128, -500])).tolist() == [0, 17, 1, 23, 24, 8, 31, 31, 15]


def test_vae_unit_ops_and_full_frame(golden):
    """Per-op goldens from the reference's own CausalConv3d / SpatialDownsample2x / SpatialUpsample2x / Normalize / ResnetBlock3D
    modules (tests/golden/vaeunits.npz) against the oracle's primitives; full-width one-frame decode statistics against the
    oracle's decoder."""
    g = golden("vaeunits")
    u = cases.vae_unit_cases()

    def close(a, ref, tol=2e-5):
        assert a.shape == ref.shape and np.abs(a - ref).max() <= tol * max(1.0, np.abs(ref).max())

    c = u["conv_k3"]
    close(O.causal_conv3d(c["x"], c["w"], c["b"], 1), g["unit_conv_k3"])
    c = u["down"]
    close(O.conv_nd_general(c["x"], c["w"], c["b"], (1, 2, 2), ((0, 0), (0, 1), (0, 1)), causal_t=True), g["unit_down"])
    c = u["up"]
    close(O.causal_conv3d(O.nearest_up2(c["x"]), c["w"], c["b"], 1), g["unit_up"])
    c = u["gn"]
    close(O.group_norm(c["x"], c["g"], c["b"]), g["unit_gn"])
    close(O.swish(O.group_norm(c["x"], c["g"], c["b"])), g["unit_gn_swish"])
    c = u["res"]
    h = O.causal_conv3d(O.swish(O.group_norm(c["x"], c["g1"], c["b1"])), c["w1"], c["c1"], 1)
    h = O.causal_conv3d(O.swish(O.group_norm(h, c["g2"], c["b2"])), c["w2"], c["c2"], 1)
    close(O.causal_conv3d(c["x"], c["ws"], c["cs"], 0) + h, g["unit_res"])


def test_codebook_forward(golden):
    """Codebook.forward (eval) of the reference's own class (tests/golden/codebook.npz): indices, straight-through embeddings,
    commitment loss, perplexity."""
    g = golden("codebook")
    r = cases.rng(41)
    E = r.standard_normal((2048, 256), dtype=np.float32)
    z = r.standard_normal((2, 256, 2, 4, 4), dtype=np.float32)
    res = O.video_codebook_forward(z, E)
    assert g["cb_gap"].min() > 1e-2                       # every choice of the reference is decided far beyond fp32 summation noise
    assert (res["encodings"] == g["cb_encodings"]).all()
    np.testing.assert_array_equal(res["embeddings"], g["cb_embeddings"])
    np.testing.assert_allclose(res["commitment_loss"], g["cb_commitment_loss"], rtol=2e-6)
    np.testing.assert_allclose(res["perplexity"], g["cb_perplexity"], rtol=2e-5)
    ids = r.integers(0, 64, size=(1, 3, 4, 4))
    z2 = (np.moveaxis(E[ids], -1, 1) + 0.01 * r.standard_normal((1, 256, 3, 4, 4), dtype=np.float32)).astype(np.float32)
    res2 = O.video_codebook_forward(z2, E)
    assert (res2["encodings"] == ids).all()
    np.testing.assert_allclose(res2["commitment_loss"], g["cb2_commitment_loss"], rtol=2e-5)
    np.testing.assert_allclose(res2["perplexity"], g["cb2_perplexity"], rtol=2e-5)
