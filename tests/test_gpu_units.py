"""GPU parity of the decoder building blocks against per-op goldens from the reference's own modules (tests/golden/vaeunits.npz),
the full Codebook.forward (tests/golden/codebook.npz) and the full-width single-frame CausalVideoVAE decode."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import cases, detweights
from oracle import vlg_oracle as O
from vlg_testutil import to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from video_llamagen_amd import _lib
    _lib.lib()
    return _lib


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _conv(L, x, w, b, stride=1, up=0, dtype=0):
    B, Cin, T, H, W = x.shape
    Cout, _, kt, kh, kw = w.shape
    out = torch.empty((B, Cout, T, (H << up) // stride, (W << up) // stride), dtype=torch.float32, device="cuda")
    xd, wd, bd = _dev(x), _dev(w), (_dev(b) if b is not None else None)
    L.check(L.lib().vlg_causal_conv3d(L.ptr(xd), L.ptr(wd), L.ptr(bd), B, Cin, T, H, W, Cout, kt, kh, kw, stride, up, dtype, L.ptr(out),
                                      L.stream_ptr()))
    return to_np(out)


def _gn(L, x, g, b, swish, dtype=0):
    B, Cc = x.shape[:2]
    P = int(np.prod(x.shape[2:]))
    out = torch.empty(x.shape, dtype=torch.float32, device="cuda")
    xd, gd, bd = _dev(x), _dev(g), _dev(b)
    L.check(L.lib().vlg_group_norm(L.ptr(xd), L.ptr(gd), L.ptr(bd), B, Cc, C.c_int64(P), C.c_float(1e-6), 1 if swish else 0, dtype, L.ptr(out),
                                   L.stream_ptr()))
    return to_np(out)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_vae_unit_ops(L, golden, dt):
    """CausalConv3d k3, SpatialDownsample2x, SpatialUpsample2x, Normalize (+swish), TimeUpsample2x and a whole ResnetBlock3D composed
    from those kernels, against the reference modules' outputs.  fp32: accumulation-order noise only; bf16: inputs, weights and
    outputs rounded to bf16 around an fp32 accumulation (2e-2 of the output range)."""
    g = golden("vaeunits")
    u = cases.vae_unit_cases()
    code = 0 if dt == "fp32" else 1
    tol = 2e-5 if dt == "fp32" else 2e-2

    def close(a, ref, scale_tol=tol):
        assert a.shape == ref.shape
        assert np.abs(a - ref).max() <= scale_tol * max(1.0, np.abs(ref).max()), np.abs(a - ref).max()

    c = u["conv_k3"]
    close(_conv(L, c["x"], c["w"], c["b"], dtype=code), g["unit_conv_k3"])
    c = u["down"]
    close(_conv(L, c["x"], c["w"], c["b"], stride=2, dtype=code), g["unit_down"])
    c = u["up"]
    close(_conv(L, c["x"], c["w"], c["b"], up=1, dtype=code), g["unit_up"])
    c = u["gn"]
    close(_gn(L, c["x"], c["g"], c["b"], False, code), g["unit_gn"])
    close(_gn(L, c["x"], c["g"], c["b"], True, code), g["unit_gn_swish"])
    c = u["res"]                                             # resnet_block.py:158-172 out of the unit kernels
    h = _conv(L, _gn(L, c["x"], c["g1"], c["b1"], True, code), c["w1"], c["c1"], dtype=code)
    h = _conv(L, _gn(L, h, c["g2"], c["b2"], True, code), c["w2"], c["c2"], dtype=code)
    close(_conv(L, c["x"], c["ws"], c["cs"], dtype=code) + h, g["unit_res"], tol * 2)
    gv = golden("vae")                                       # TimeUpsample2x golden of round 1 (updownsample.py:182-194)
    x2 = cases.rng(34).standard_normal((1, 4, 5, 2, 2), dtype=np.float32)
    out = torch.empty((1, 4, 9, 2, 2), dtype=torch.float32, device="cuda")
    xd = _dev(x2)
    L.check(L.lib().vlg_time_upsample2x(L.ptr(xd), 1, 4, 5, C.c_int64(4), code, L.ptr(out), L.stream_ptr()))
    close(to_np(out), gv["vae_timeup"])


def test_unit_entry_point_errors(L):
    x = torch.zeros(1, 8, 1, 4, 4, device="cuda")
    w = torch.zeros(8, 8, 3, 5, 5, device="cuda")
    out = torch.zeros(1, 8, 1, 4, 4, device="cuda")
    with pytest.raises(L.VlgError):
        L.check(L.lib().vlg_causal_conv3d(L.ptr(x), L.ptr(w), None, 1, 8, 1, 4, 4, 8, 3, 5, 5, 1, 0, 0, L.ptr(out), L.stream_ptr()))
    with pytest.raises(L.VlgError):      # 24 channels do not divide into 32 groups
        L.check(L.lib().vlg_group_norm(L.ptr(x), L.ptr(x), L.ptr(x), 1, 24, C.c_int64(16), C.c_float(1e-6), 0, 0, L.ptr(out), L.stream_ptr()))


def test_vae_full_width_single_frame_vs_reference(golden):
    """The reference's own Decoder at its constructor defaults (512/256/128 channels) on one latent frame [1,8,1,32,32] ->
    [1,3,1,256,256] (modeling_causalvae.py:151-262): crops, a strided grid and sums of the reference's fp32 output."""
    import video_llamagen_amd as V
    g = golden("vaeunits")
    cfg = dict(hidden_size=128, z_channels=4, embed_dim=8, hidden_size_mult=(1, 2, 4, 4), num_res_blocks=2)
    sd = detweights.vae_weights(cfg)
    z = cases.rng(38).standard_normal((1, 8, 1, 32, 32), dtype=np.float32)
    scale = max(abs(g["full1f_stats"][2]), abs(g["full1f_stats"][3]))
    for dt, tol in ((torch.float32, 2e-3), (torch.bfloat16, 6e-2)):
        m = V.VAE_models["VAE-16"](embed_dim=8).to("cuda", dt)
        _, skipped = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        assert skipped == []
        y = to_np(m.decode(torch.from_numpy(z)))
        assert y.shape == (1, 3, 1, 256, 256)
        assert np.abs(y[0, :, 0, 100:132, 100:132] - g["full1f_crop"]).max() < tol * scale
        assert np.abs(y[0, :, 0, ::8, ::8] - g["full1f_grid"]).max() < tol * scale
        if dt == torch.float32:
            assert abs(y.astype(np.float64).sum() - g["full1f_stats"][0]) < 2e-3 * g["full1f_stats"][1]
            assert abs(np.abs(y).astype(np.float64).sum() - g["full1f_stats"][1]) < 2e-3 * g["full1f_stats"][1]


def test_codebook_forward_vs_reference(golden):
    """Codebook.forward (eval) through vlg_codebook_forward - nearest neighbour on the exact-fp32 matrix cores - against the
    reference class's outputs: indices bit-exact (every gap of the reference exceeds 0.16, fp32 summation noise is ~1e-4 here),
    straight-through embeddings bit-exact, loss / perplexity to fp32 summation order."""
    import video_llamagen_amd as V
    g = golden("codebook")
    r = cases.rng(41)
    E = r.standard_normal((2048, 256), dtype=np.float32)
    z = r.standard_normal((2, 256, 2, 4, 4), dtype=np.float32)
    cb = V.Codebook(2048, 256)
    cb.load_state_dict({"embeddings": torch.from_numpy(E), "N": torch.zeros(2048), "z_avg": torch.from_numpy(E)})
    res = cb(torch.from_numpy(z))
    assert res["encodings"].dtype == torch.int64 and tuple(res["encodings"].shape) == (2, 2, 4, 4)
    assert (res["encodings"].cpu().numpy() == g["cb_encodings"]).all()
    np.testing.assert_array_equal(to_np(res["embeddings"]), g["cb_embeddings"])
    np.testing.assert_allclose(float(res["commitment_loss"]), g["cb_commitment_loss"], rtol=2e-6)
    np.testing.assert_allclose(float(res["perplexity"]), g["cb_perplexity"], rtol=2e-5)
    ids = r.integers(0, 64, size=(1, 3, 4, 4))
    z2 = (np.moveaxis(E[ids], -1, 1) + 0.01 * r.standard_normal((1, 256, 3, 4, 4), dtype=np.float32)).astype(np.float32)
    res2 = cb(torch.from_numpy(z2))
    assert (res2["encodings"].cpu().numpy() == ids).all()
    np.testing.assert_allclose(float(res2["commitment_loss"]), g["cb2_commitment_loss"], rtol=2e-5)
    np.testing.assert_allclose(float(res2["perplexity"]), g["cb2_perplexity"], rtol=2e-5)
    assert torch.equal(cb.dictionary_lookup(res2["encodings"]).cpu(), torch.from_numpy(E[ids]))
    # ragged sizes: rows not a multiple of the 64-row tile, codes not a multiple of 32 / 128, small dims (fallback kernel for dim % 8 != 0)
    for n_codes, dim, shape in ((100, 64, (1, 64, 1, 5, 7)), (2048, 256, (3, 256, 1, 9, 9)), (33, 24, (2, 24, 2, 3, 3)), (17, 12, (1, 12, 1, 2, 5))):
        Er = r.standard_normal((n_codes, dim), dtype=np.float32)
        zr = r.standard_normal(shape, dtype=np.float32)
        c2 = V.Codebook(n_codes, dim)
        c2.load_state_dict({"embeddings": torch.from_numpy(Er)})
        got = c2(torch.from_numpy(zr))
        want = O.video_codebook_forward(zr, Er)
        _, d = O.video_codebook_argmin(zr, Er)
        srt = np.sort(d, -1)
        decided = (srt[:, 1] - srt[:, 0]) > 1e-3
        same = got["encodings"].cpu().numpy().reshape(-1) == want["encodings"].reshape(-1)
        assert (same | ~decided).all() and decided.mean() > 0.9, (n_codes, dim)
        if same.all():
            np.testing.assert_array_equal(to_np(got["embeddings"]), want["embeddings"])
            np.testing.assert_allclose(float(got["perplexity"]), want["perplexity"], rtol=2e-5)
    # full-size property: 4 videos x 5 x 32 x 32 positions made of codebook rows -> every index recovered
    big = r.integers(0, 2048, size=(4, 5, 32, 32))
    zb = torch.from_numpy(E).cuda()[torch.from_numpy(big).cuda()].permute(0, 4, 1, 2, 3).contiguous()
    rb = cb(zb)
    assert torch.equal(rb["encodings"].cpu(), torch.from_numpy(big)) and float(rb["commitment_loss"]) == 0.0


def test_codebook_forward_on_two_streams():
    """Two Codebook objects with different sizes driven from two streams at once (scratch buffers are per stream): the same results as
    one after the other."""
    import video_llamagen_amd as V
    r = cases.rng(43)
    specs = ((2048, 256, (2, 256, 2, 8, 8)), (1000, 64, (3, 64, 1, 9, 7)))
    cbs, zs, want = [], [], []
    for n_codes, dim, shape in specs:
        E = r.standard_normal((n_codes, dim), dtype=np.float32)
        cb = V.Codebook(n_codes, dim)
        cb.load_state_dict({"embeddings": torch.from_numpy(E)})
        z = torch.from_numpy(r.standard_normal(shape, dtype=np.float32)).cuda()
        cbs.append(cb)
        zs.append(z)
        res = cb(z)
        want.append({k: v.clone() for k, v in res.items()})
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[], []]
    for _ in range(20):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                got[i].append(cbs[i](zs[i]))
    torch.cuda.synchronize()
    for i in (0, 1):
        for res in got[i]:
            for k in ("encodings", "embeddings", "commitment_loss", "perplexity"):
                assert torch.equal(res[k], want[i][k]), (i, k)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_causal_conv3d_halo_tile_planar_output(L, dt):
    """vlg_causal_conv3d at shapes the halo-tile kernel takes (3x3x3 and 1x3x3, Cin a multiple of 32, Cout 128 / 256) with the entry
    point's planar fp32 output: ragged tile edges (sizes that are not multiples of the 2 x 4 x 32 tile), nearest-2x upsampling in the
    gather, several channel chunks - against the oracle's CausalConv3d.  bf16 runs the 12-wave form with loader waves."""
    code = 0 if dt == "fp32" else 1
    tol = 3e-5 if dt == "fp32" else 2e-2
    r = cases.rng(47)
    for (B, Cin, T, H, W, Cout, kt, up) in ((1, 64, 3, 5, 33, 128, 3, 0), (2, 32, 1, 9, 40, 256, 1, 0), (1, 96, 2, 3, 17, 128, 3, 1)):
        x = r.standard_normal((B, Cin, T, H, W), dtype=np.float32)
        w = (r.standard_normal((Cout, Cin, kt, 3, 3), dtype=np.float32) / np.sqrt(Cin * kt * 9)).astype(np.float32)
        b = r.standard_normal((Cout,), dtype=np.float32)
        got = _conv(L, x, w, b, up=up, dtype=code)
        xu = x.repeat(2, axis=3).repeat(2, axis=4) if up else x
        ref = O.causal_conv3d(xu, w, b, 1)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (B, Cin, T, H, W, Cout, kt, up, np.abs(got - ref).max())
