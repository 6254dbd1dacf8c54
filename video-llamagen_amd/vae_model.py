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
        """modeling_videobase.py:42-53: directory with config.json + *.ckpt (the last one in glob order is loaded)."""
        from . import io as vio
        cfg, ckpt = vio.find_vae_checkpoint(pretrained_model_name_or_path)
        model = cls.from_config(cfg)
        if "device" in kwargs or "dtype" in kwargs:
            model.to(kwargs.get("device"), kwargs.get("dtype"))
        model.init_from_ckpt(ckpt)
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

    # ---- tiling (modeling_causalvae.py:424-443,468-570): host-level orchestration over plain decodes; tiles overlap and are
    # blended linearly, temporal chunks overlap by one latent frame whose first decoded frame is dropped ----------------------
    @staticmethod
    def blend_v(a, b, blend_extent):
        blend_extent = min(a.shape[3], b.shape[3], blend_extent)
        for y in range(blend_extent):
            b[:, :, :, y, :] = a[:, :, :, -blend_extent + y, :] * (1 - y / blend_extent) + b[:, :, :, y, :] * (y / blend_extent)
        return b

    @staticmethod
    def blend_h(a, b, blend_extent):
        blend_extent = min(a.shape[4], b.shape[4], blend_extent)
        for x in range(blend_extent):
            b[:, :, :, :, x] = a[:, :, :, :, -blend_extent + x] * (1 - x / blend_extent) + b[:, :, :, :, x] * (x / blend_extent)
        return b

    def tiled_decode(self, x):
        t = x.shape[2]
        t_chunk_idx = [i for i in range(0, t, self.tile_latent_min_size_t - 1)]
        if len(t_chunk_idx) == 1 and t_chunk_idx[0] == 0:
            t_chunk_start_end = [[0, t]]
        else:
            t_chunk_start_end = [[t_chunk_idx[i], t_chunk_idx[i + 1] + 1] for i in range(len(t_chunk_idx) - 1)]
            if t_chunk_start_end[-1][-1] > t:
                t_chunk_start_end[-1][-1] = t
            elif t_chunk_start_end[-1][-1] < t:
                t_chunk_start_end.append([t_chunk_idx[-1], t])
        dec_ = []
        for idx, (start, end) in enumerate(t_chunk_start_end):
            dec = self.tiled_decode2d(x[:, :, start:end])
            dec_.append(dec[:, :, 1:] if idx != 0 else dec)
        return torch.cat(dec_, dim=2)

    def tiled_decode2d(self, z):
        overlap_size = int(self.tile_latent_min_size * (1 - self.tile_overlap_factor))
        blend_extent = int(self.tile_sample_min_size * self.tile_overlap_factor)
        row_limit = self.tile_sample_min_size - blend_extent
        rows = []
        for i in range(0, z.shape[3], overlap_size):
            row = []
            for j in range(0, z.shape[4], overlap_size):
                tile = z[:, :, :, i:i + self.tile_latent_min_size, j:j + self.tile_latent_min_size]
                row.append(self._decode_plain(tile))          # post_quant_conv + decoder on the GPU (libvlg)
            rows.append(row)
        result_rows = []
        for i, row in enumerate(rows):
            result_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self.blend_v(rows[i - 1][j], tile, blend_extent)
                if j > 0:
                    tile = self.blend_h(row[j - 1], tile, blend_extent)
                result_row.append(tile[:, :, :, :row_limit, :row_limit])
            result_rows.append(torch.cat(result_row, dim=4))
        return torch.cat(result_rows, dim=3)

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
        """modeling_causalvae.py:444-466: temporal chunks of tile_sample_min_size_t frames sharing one frame (the repeated latent frame
        of every later chunk is dropped), each chunk tiled spatially by tiled_encode2d."""
        t = x.shape[2]
        starts = list(range(0, t, self.tile_sample_min_size_t - 1))
        if len(starts) == 1:
            spans = [[0, t]]
        else:
            spans = [[starts[i], starts[i + 1] + 1] for i in range(len(starts) - 1)]
            if spans[-1][1] > t:
                spans[-1][1] = t
            elif spans[-1][1] < t:
                spans.append([starts[-1], t])
        parts = []
        for n, (a, b) in enumerate(spans):
            mom = self.tiled_encode2d(x[:, :, a:b], return_moments=True)
            parts.append(mom if n == 0 else mom[:, :, 1:])
        return DiagonalGaussianDistribution(torch.cat(parts, dim=2))

    @torch.no_grad()
    def tiled_encode2d(self, x, return_moments=False):
        """modeling_causalvae.py:491-530: tile_sample_min_size-pixel tiles at stride size*(1-overlap), moments of neighbouring tiles
        blended over tile_latent_min_size*overlap latent cells, each tile cropped to the stride in latent cells."""
        stride = int(self.tile_sample_min_size * (1 - self.tile_overlap_factor))
        extent = int(self.tile_latent_min_size * self.tile_overlap_factor)
        keep = self.tile_latent_min_size - extent
        grid = [[self._encode_moments(x[:, :, :, i:i + self.tile_sample_min_size, j:j + self.tile_sample_min_size])
                 for j in range(0, x.shape[4], stride)] for i in range(0, x.shape[3], stride)]
        out_rows = []
        for i, row in enumerate(grid):
            cells = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self.blend_v(grid[i - 1][j], tile, extent)
                if j > 0:
                    tile = self.blend_h(row[j - 1], tile, extent)
                cells.append(tile[:, :, :, :keep, :keep])
            out_rows.append(torch.cat(cells, dim=4))
        moments = torch.cat(out_rows, dim=3)
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
