"""Host mirror of CausalVAEModel's decode side (CausalVideoVAE/causalvideovae/model/causal_vae/modeling_causalvae.py:265-404).

`VAE_models['VAE-16']` (tokenizer/tokenizer_image/vae_model.py:8) -> object with `.config.embed_dim`, `.enable_tiling()`,
`.tile_overlap_factor`, `.to(dtype)`, `.load_state_dict`, `.decode(z[B,C,t,h,w]) -> [B,3,T,H,W]`.  Tiling never triggers below
the reference's own thresholds; above them `tiled_decode` mirrors modeling_causalvae.py:468-570 (host-level
orchestration over plain decodes + linear blending).
"""
import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib as L


class DiagonalGaussianDistribution:
    """utils/distrib_utils.py:4-42 (mean | logvar along dim 1, logvar clamped to [-30, 20])."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, device=self.mean.device)
        return self.mean + self.std * noise.to(self.mean.device)

    def mode(self):
        return self.mean


class CausalVAEModel:
    def __init__(self, hidden_size=128, z_channels=4, hidden_size_mult=(1, 2, 4, 4), attn_resolutions=(), dropout=0.0,
                 resolution=256, double_z=True, embed_dim=4, num_res_blocks=2,
                 decoder_spatial_upsample=("", "SpatialUpsample2x", "SpatialUpsample2x", "SpatialUpsample2x"),
                 decoder_temporal_upsample=("", "", "TimeUpsample2x", "TimeUpsample2x"), use_quant_layer=True, **_unused):
        if attn_resolutions:
            raise L.VlgError(-3, "attn_resolutions != [] is not used by the reference configuration")
        if not use_quant_layer:
            raise L.VlgError(-3, "use_quant_layer=False is not supported")
        self.config = SimpleNamespace(hidden_size=hidden_size, z_channels=z_channels, hidden_size_mult=tuple(hidden_size_mult),
                                      embed_dim=embed_dim, num_res_blocks=num_res_blocks, resolution=resolution,
                                      decoder_spatial_upsample=tuple(decoder_spatial_upsample),
                                      decoder_temporal_upsample=tuple(decoder_temporal_upsample))
        self.tile_sample_min_size = 512
        self.tile_sample_min_size_t = 17
        self.tile_latent_min_size = int(self.tile_sample_min_size / (2 ** (len(hidden_size_mult) - 1)))   # :326
        self.tile_latent_min_size_t = int((self.tile_sample_min_size_t - 1) / 4) + 1                      # :328
        self.tile_overlap_factor = 0.125
        self.use_tiling = False
        self._dtype = torch.bfloat16
        self._device = None
        self._handle = None

    @classmethod
    def from_config(cls, cfg: dict):
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        """modeling_videobase.py:42-53: directory with config.json + *.ckpt (the last one in glob order is loaded through init_from_ckpt);
        without a *.ckpt the diffusers layout of `super().from_pretrained`: diffusion_pytorch_model[.variant].safetensors / .bin, optional
        `subfolder`, the plain state dict loaded strictly."""
        from . import io as vio
        cfg, path = vio.find_vae_checkpoint(pretrained_model_name_or_path, subfolder=kwargs.get("subfolder"), variant=kwargs.get("variant"))
        model = cls.from_config(cfg)
        if "device" in kwargs or "dtype" in kwargs or "torch_dtype" in kwargs:
            model.to(kwargs.get("device"), kwargs.get("dtype", kwargs.get("torch_dtype")))
        if path.endswith(".ckpt"):
            model.init_from_ckpt(path)
        else:
            model.load_state_dict(dict(vio.load_vae_weight_file(path)), strict=True)
        return model

    def init_from_ckpt(self, path, ignore_keys=()):
        """modeling_causalvae.py:578-601 (EMA weights preferred, strict load of the tensors this engine uses)."""
        from . import io as vio
        sd = vio.select_vae_state_dict(torch.load(path, map_location="cpu"), ignore_keys)
        self.load_state_dict(sd, strict=True)

    def enable_tiling(self, use_tiling: bool = True):
        self.use_tiling = use_tiling

    def eval(self):
        return self

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if dtype is not None and dtype != self._dtype:
            if self._handle is not None:
                raise L.VlgError(-6, "dtype must be chosen before weights are loaded")
            L.torch_dtype_code(dtype)
            self._dtype = dtype
        if device is not None:
            self._device = torch.device(device)
        return self

    def _ensure_handle(self):
        if self._handle is not None:
            return
        if self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        c = self.config
        cfg = L.VaeConfig(hidden_size=c.hidden_size, z_channels=c.z_channels, embed_dim=c.embed_dim, num_res_blocks=c.num_res_blocks,
                          n_mult=len(c.hidden_size_mult), dtype=L.torch_dtype_code(self._dtype))
        for i, m in enumerate(c.hidden_size_mult):
            cfg.hidden_size_mult[i] = m
            cfg.spatial_upsample[i] = 1 if c.decoder_spatial_upsample[i] else 0
            cfg.temporal_upsample[i] = 1 if c.decoder_temporal_upsample[i] else 0
        h = C.c_void_p()
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vae_create(C.byref(cfg), C.byref(h)))
        self._handle = h

    def load_state_dict(self, state_dict, strict=True):
        self._ensure_handle()
        skipped = []
        with torch.cuda.device(self._device):
            for k, v in state_dict.items():
                if not L.load_tensor(L.lib().vlg_vae_load_tensor, self._handle, k, v):
                    skipped.append(k)
        return [], skipped

    def decoder_param_shapes(self):
        """name -> shape of every tensor the decode path needs (state-dict names of modeling_causalvae.py:151-262,369)."""
        c = self.config
        out = {}

        def conv(name, cout, cin, k):
            out[name + ".conv.weight"] = (cout, cin) + tuple(k)
            out[name + ".conv.bias"] = (cout,)

        def gn(name, ch):
            out[name + ".weight"] = (ch,)
            out[name + ".bias"] = (ch,)

        def res(p, cin, cout):
            gn(p + ".norm1", cin)
            conv(p + ".conv1", cout, cin, (3, 3, 3))
            gn(p + ".norm2", cout)
            conv(p + ".conv2", cout, cout, (3, 3, 3))
            if cin != cout:
                conv(p + ".nin_shortcut", cout, cin, (1, 1, 1))

        n = len(c.hidden_size_mult)
        block_in = c.hidden_size * c.hidden_size_mult[n - 1]
        conv("post_quant_conv", c.z_channels, c.embed_dim, (1, 1, 1))
        conv("decoder.conv_in", block_in, c.z_channels, (3, 3, 3))
        res("decoder.mid.block_1", block_in, block_in)
        gn("decoder.mid.attn_1.norm", block_in)
        for nm in ("q", "k", "v", "proj_out"):
            conv("decoder.mid.attn_1." + nm, block_in, block_in, (1, 1, 1))
        res("decoder.mid.block_2", block_in, block_in)
        for lvl in reversed(range(n)):
            block_out = c.hidden_size * c.hidden_size_mult[lvl]
            for j in range(c.num_res_blocks + 1):
                res("decoder.up.%d.block.%d" % (lvl, j), block_in, block_out)
                block_in = block_out
            if c.decoder_spatial_upsample[lvl]:
                conv("decoder.up.%d.upsample.conv" % lvl, block_in, block_in, (1, 3, 3))
        gn("decoder.norm_out", block_in)
        conv("decoder.conv_out", 3, block_in, (3, 3, 3))
        return out

    def init_random_weights(self, seed=0):
        """Random initialisation on the device (stand-in for the reference constructor's nn.Conv3d / GroupNorm init)."""
        self._ensure_handle()
        g = torch.Generator(device=self._device).manual_seed(seed)
        for name, shape in self.decoder_param_shapes().items():
            if name.endswith("conv.weight"):
                fan_in = shape[1] * shape[2] * shape[3] * shape[4]
                t = torch.randn(shape, generator=g, device=self._device) * (0.7 / fan_in ** 0.5)
            elif ".norm" in name and name.endswith(".weight"):
                t = torch.ones(shape, device=self._device)
            else:
                t = torch.zeros(shape, device=self._device)
            self.load_state_dict({name: t})
        return self

    @torch.no_grad()
    def decode(self, z):
        """z [B, embed_dim, t, h, w] -> float32 [B, 3, T, H, W] (modeling_causalvae.py:394-404)."""
        self._ensure_handle()
        if z.dim() != 5 or z.shape[1] != self.config.embed_dim:
            raise L.VlgError(-2, "z must be [B, %d, t, h, w], got %s" % (self.config.embed_dim, tuple(z.shape)))
        if self.use_tiling and (z.shape[-1] > self.tile_latent_min_size or z.shape[-2] > self.tile_latent_min_size
                                or z.shape[-3] > self.tile_latent_min_size_t):
            return self.tiled_decode(z)
        return self._decode_plain(z)

    # ---- tiling (what modeling_causalvae.py:424-570 computes) ------------------------------------------------------------
    # Time: the clip is cut into windows that share one frame; every window after the first contributes everything but its first
    # output frame.  Space: overlapping square tiles in raster order; libvlg's vlg_tile_blend cross-fades each finished tile in place
    # against its upper and left neighbours and writes the part that is kept straight into the output canvas (no Python per-row
    # loops, no torch.cat of cropped tiles).
    @staticmethod
    def _time_windows(length, window):
        """[lo, hi) windows of `window` frames at stride window - 1 (adjacent windows share a frame); a tail shorter than a full
        window becomes its own window, one that would overrun is clipped."""
        starts = list(range(0, length, window - 1))
        if len(starts) == 1:
            return [(0, length)]
        spans = [[lo, nxt + 1] for lo, nxt in zip(starts[:-1], starts[1:])]
        if spans[-1][1] > length:
            spans[-1][1] = length
        elif spans[-1][1] < length:
            spans.append([starts[-1], length])
        return [tuple(sp) for sp in spans]

    def _composite(self, x, tile_in, stride_in, run, extent, keep):
        """Runs `run` on every tile_in x tile_in window of x's last two axes (raster order, stride stride_in) and assembles the
        blended result: each tile fades over `extent` cells into the tiles above / left of it and contributes its first `keep`
        rows and columns."""
        ys = list(range(0, x.shape[3], stride_in))
        xs = list(range(0, x.shape[4], stride_in))
        lib = L.lib()
        st = L.stream_ptr(self._device)
        done = {}                    # (row, col) -> finished (already faded) tile: the row above and the current row stay alive
        canvas = None
        cy = 0
        for r, y_in in enumerate(ys):
            cx = 0
            for c, x_in in enumerate(xs):
                tile = run(x[:, :, :, y_in:y_in + tile_in, x_in:x_in + tile_in]).contiguous()
                Bq, Cq, Tq, th, tw = [int(v) for v in tile.shape]
                if canvas is None:
                    # every tile of a row has the same height and every tile of a column the same width, so the canvas size
                    # follows from the first tile's scale factor
                    fy, fx = th / min(tile_in, x.shape[3]), tw / min(tile_in, x.shape[4])
                    Hc = sum(min(keep, int(round(fy * min(tile_in, x.shape[3] - yy)))) for yy in ys)
                    Wc = sum(min(keep, int(round(fx * min(tile_in, x.shape[4] - xx)))) for xx in xs)
                    canvas = torch.empty((Bq, Cq, Tq, Hc, Wc), dtype=torch.float32, device=self._device)
                up = done.get((r - 1, c))
                lf = done.get((r, c - 1))
                with torch.cuda.device(self._device):
                    L.check(lib.vlg_tile_blend(L.ptr(tile), L.ptr(up), L.ptr(lf), C.c_int64(Bq * Cq * Tq), th, tw,
                                               0 if up is None else int(up.shape[3]), 0 if lf is None else int(lf.shape[4]), extent,
                                               L.ptr(canvas), int(canvas.shape[3]), int(canvas.shape[4]), cy, cx, keep, keep, st))
                done[(r, c)] = tile
                cx += min(keep, tw)
                row_h = min(keep, th)
            for c in range(len(xs)):
                done.pop((r - 1, c), None)
            cy += row_h
        return canvas

    def tiled_decode(self, z):
        parts = []
        for n, (lo, hi) in enumerate(self._time_windows(z.shape[2], self.tile_latent_min_size_t)):
            frames = self.tiled_decode2d(z[:, :, lo:hi])
            parts.append(frames if n == 0 else frames[:, :, 1:])
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=2)

    def tiled_decode2d(self, z):
        """Latent tiles of tile_latent_min_size cells at stride size*(1-overlap); decoded tiles fade over tile_sample_min_size*overlap
        pixels and keep tile_sample_min_size minus that many."""
        fade = int(self.tile_sample_min_size * self.tile_overlap_factor)
        return self._composite(z.to(self._device), self.tile_latent_min_size, int(self.tile_latent_min_size * (1 - self.tile_overlap_factor)),
                               self._decode_plain, fade, self.tile_sample_min_size - fade)

    @torch.no_grad()
    def encode(self, x):
        """x float [B,3,T,H,W] -> DiagonalGaussianDistribution over [B, embed_dim, (T-1)/4+1, H/8, W/8] (modeling_causalvae.py:382-392)."""
        self._ensure_handle()
        if x.dim() != 5 or x.shape[1] != 3:
            raise L.VlgError(-2, "x must be [B,3,T,H,W], got %s" % (tuple(x.shape),))
        if self.use_tiling and (x.shape[-1] > self.tile_sample_min_size or x.shape[-2] > self.tile_sample_min_size
                                or x.shape[-3] > self.tile_sample_min_size_t):
            return self.tiled_encode(x)
        return DiagonalGaussianDistribution(self._encode_moments(x))

    @torch.no_grad()
    def tiled_encode(self, x):
        """modeling_causalvae.py:444-466: windows of tile_sample_min_size_t frames sharing one frame (the repeated latent frame of
        every later window is dropped), each window tiled spatially by tiled_encode2d."""
        parts = []
        for n, (lo, hi) in enumerate(self._time_windows(x.shape[2], self.tile_sample_min_size_t)):
            mom = self.tiled_encode2d(x[:, :, lo:hi], return_moments=True)
            parts.append(mom if n == 0 else mom[:, :, 1:])
        return DiagonalGaussianDistribution(parts[0] if len(parts) == 1 else torch.cat(parts, dim=2))

    @torch.no_grad()
    def tiled_encode2d(self, x, return_moments=False):
        """modeling_causalvae.py:491-530: tile_sample_min_size-pixel tiles at stride size*(1-overlap), moments of neighbouring tiles
        faded over tile_latent_min_size*overlap latent cells, each tile keeping tile_latent_min_size minus that many."""
        fade = int(self.tile_latent_min_size * self.tile_overlap_factor)
        moments = self._composite(x.to(self._device), self.tile_sample_min_size, int(self.tile_sample_min_size * (1 - self.tile_overlap_factor)),
                                  self._encode_moments, fade, self.tile_latent_min_size - fade)
        return moments if return_moments else DiagonalGaussianDistribution(moments)

    def _encode_moments(self, x):
        B, _, T, H, W = [int(v) for v in x.shape]
        xf = x.to(device=self._device, dtype=torch.float32).contiguous()
        n = len(self.config.hidden_size_mult) - 1
        t = T
        for _ in range(2):
            t = (t - 1) // 2 + 1 if t > 1 else t
        mom = torch.empty((B, 2 * self.config.embed_dim, t, H >> n, W >> n), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vae_encode(self._handle, L.ptr(xf), B, T, H, W, L.ptr(mom), L.stream_ptr(self._device)))
        return mom

    def _decode_plain(self, z):
        B, _, t, hh, ww = [int(s) for s in z.shape]
        zf = z.to(device=self._device, dtype=torch.float32).contiguous()
        T, H, W = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(L.lib().vlg_vae_out_shape(self._handle, t, hh, ww, C.byref(T), C.byref(H), C.byref(W)))
        out = torch.empty((B, 3, T.value, H.value, W.value), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vae_decode(self._handle, L.ptr(zf), B, t, hh, ww, L.ptr(out), L.stream_ptr(self._device)))
        return out

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().vlg_vae_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


VAE_models = {'VAE-16': CausalVAEModel}
