"""Host mirror of the image tokenizer's decode side (tokenizer/tokenizer_image/vq_model.py).

`VQ_models[name](codebook_size=16384, codebook_embed_dim=8)` -> object with `.load_state_dict`, `.eval`, `.to`,
`.decode_code(code_b, shape, channel_first=True)` (vq_model.py:52-55) and `.quantize_indices(z)` (the argmin of
VectorQuantizer.forward, vq_model.py:215-233).  Compute runs in libvlg.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List

import torch

from . import _lib as L


@dataclass
class ModelArgs:
    codebook_size: int = 16384
    codebook_embed_dim: int = 8
    codebook_l2_norm: bool = True
    codebook_show_usage: bool = True
    commit_loss_beta: float = 0.25
    entropy_loss_ratio: float = 0.0
    encoder_ch_mult: List[int] = field(default_factory=lambda: [1, 1, 2, 2, 4])
    decoder_ch_mult: List[int] = field(default_factory=lambda: [1, 1, 2, 2, 4])
    z_channels: int = 256
    dropout_p: float = 0.0


class VQModel:
    def __init__(self, config: ModelArgs):
        self.config = config
        # activation/weight dtype of the encoder/decoder convs.  fp32 as in the reference's scripts (vq_model.to(device), sample_t2i.py:41-48):
        # fp32 MFMA kernels; `.to(dtype=torch.bfloat16)` selects the 10x faster bf16 kernels (pixels within 3 % of the output range)
        self._dtype = torch.float32
        self._device = None
        self._handle = None
        self.training = False

    def eval(self):
        self.training = False
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
        cfg = L.VqConfig(codebook_size=c.codebook_size, codebook_embed_dim=c.codebook_embed_dim, z_channels=c.z_channels,
                         ch=128, n_mult=len(c.decoder_ch_mult), num_res_blocks=2, l2_norm=1 if c.codebook_l2_norm else 0,
                         dtype=L.torch_dtype_code(self._dtype))
        for i, m in enumerate(c.decoder_ch_mult):
            cfg.ch_mult[i] = m
        h = C.c_void_p()
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vq_create(C.byref(cfg), C.byref(h)))
        self._handle = h

    def load_state_dict(self, state_dict, strict=True):
        self._ensure_handle()
        skipped = []
        with torch.cuda.device(self._device):
            for k, v in state_dict.items():
                if not L.load_tensor(L.lib().vlg_vq_load_tensor, self._handle, k, v):
                    skipped.append(k)       # encoder.*, quant_conv.*, codebook_used: not on the decode path
        return [], skipped

    def decoder_param_shapes(self):
        """name -> shape of the decode-side tensors (vq_model.py:32-39,137-167,207)."""
        c = self.config
        ch, mult, out = 128, list(c.decoder_ch_mult), {}

        def conv(n, co, ci, k):
            out[n + ".weight"] = (co, ci, k, k)
            out[n + ".bias"] = (co,)

        def gn(n, cch):
            out[n + ".weight"] = (cch,)
            out[n + ".bias"] = (cch,)

        def res(p, ci, co):
            gn(p + ".norm1", ci)
            conv(p + ".conv1", co, ci, 3)
            gn(p + ".norm2", co)
            conv(p + ".conv2", co, co, 3)
            if ci != co:
                conv(p + ".nin_shortcut", co, ci, 1)

        def attn(p, cch):
            gn(p + ".norm", cch)
            for n in ("q", "k", "v", "proj_out"):
                conv(p + "." + n, cch, cch, 1)

        out["quantize.embedding.weight"] = (c.codebook_size, c.codebook_embed_dim)
        conv("post_quant_conv", c.z_channels, c.codebook_embed_dim, 1)
        nres = len(mult)
        block_in = ch * mult[nres - 1]
        conv("decoder.conv_in", block_in, c.z_channels, 3)
        res("decoder.mid.0", block_in, block_in)
        attn("decoder.mid.1", block_in)
        res("decoder.mid.2", block_in, block_in)
        for li, lvl in enumerate(reversed(range(nres))):
            block_out = ch * mult[lvl]
            for j in range(3):
                res("decoder.conv_blocks.%d.res.%d" % (li, j), block_in, block_out)
                block_in = block_out
                if lvl == nres - 1:
                    attn("decoder.conv_blocks.%d.attn.%d" % (li, j), block_in)
            if lvl != 0:
                conv("decoder.conv_blocks.%d.upsample.conv" % li, block_in, block_in, 3)
        gn("decoder.norm_out", block_in)
        conv("decoder.conv_out", 3, block_in, 3)
        return out

    def init_random_weights(self, seed=0):
        """Random initialisation on the device (stand-in for the reference constructor's default nn.Conv2d init)."""
        self._ensure_handle()
        g = torch.Generator(device=self._device).manual_seed(seed)
        for name, shape in self.decoder_param_shapes().items():
            if len(shape) == 4:
                t = torch.randn(shape, generator=g, device=self._device) * (0.7 / (shape[1] * shape[2] * shape[3]) ** 0.5)
            elif name == "quantize.embedding.weight":
                t = torch.randn(shape, generator=g, device=self._device)
            elif ".norm" in name and name.endswith(".weight"):
                t = torch.ones(shape, device=self._device)
            else:
                t = torch.zeros(shape, device=self._device)
            self.load_state_dict({name: t})
        return self

    @torch.no_grad()
    def decode_code(self, code_b, shape=None, channel_first=True):
        """code_b int [B, h*w] (or flat), shape = [B, C, h, w] -> float32 [B, 3, 16h, 16w] (NCHW)."""
        self._ensure_handle()
        if shape is None or not channel_first:
            raise L.VlgError(-3, "decode_code needs shape=[B,C,h,w] with channel_first=True (the only form the reference's callers use)")
        B, _, gh, gw = [int(s) for s in shape]
        codes = code_b.to(device=self._device, dtype=torch.int32).contiguous().view(-1)
        if codes.numel() != B * gh * gw:
            raise L.VlgError(-2, "code tensor has %d entries, shape needs %d" % (codes.numel(), B * gh * gw))
        up = 2 ** (len(self.config.decoder_ch_mult) - 1)
        out = torch.empty((B, 3, gh * up, gw * up), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vq_decode_code(self._handle, L.ptr(codes), B, gh, gw, L.ptr(out), L.stream_ptr(self._device)))
        return out

    @torch.no_grad()
    def encode(self, x):
        """VQModel.encode (vq_model.py:41-45): x float [B,3,H,W] in [-1,1] -> (quant [B,C,h,w], (None, None, None, 0),
        (None, None, indices int32 [B*h*w])).  quant = normalised codebook rows of the indices (the value of z_q, :233,254)."""
        self._ensure_handle()
        x = x.to(device=self._device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        f = 2 ** (len(self.config.encoder_ch_mult) - 1)
        idx = torch.empty((B * (H // f) * (W // f),), dtype=torch.int32, device=self._device)
        z = torch.empty((B, self.config.codebook_embed_dim, H // f, W // f), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vq_encode(self._handle, L.ptr(x), B, H, W, L.ptr(idx), L.ptr(z), L.stream_ptr(self._device)))
        self.last_z = z
        return None, (None, None, None, 0), (None, None, idx)

    @torch.no_grad()
    def quantize_indices(self, z):
        """min_encoding_indices of VectorQuantizer.forward (vq_model.py:215-233): z float [B, C, H, W] -> int32 [B*H*W]."""
        self._ensure_handle()
        z = z.to(device=self._device, dtype=torch.float32).contiguous()
        B, Cc, H, W = z.shape
        if Cc != self.config.codebook_embed_dim:
            raise L.VlgError(-2, "z has %d channels, codebook_embed_dim is %d" % (Cc, self.config.codebook_embed_dim))
        idx = torch.empty((B * H * W,), dtype=torch.int32, device=self._device)
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vq_argmin(self._handle, L.ptr(z), B, H, W, L.ptr(idx), L.stream_ptr(self._device)))
        return idx

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().vlg_vq_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


def codebook_argmin(z_flat, codebook):
    """Codebook.forward argmin (tokenizer/tokenizer_video/vqvae.py:161-170): z [n, dim], codebook [n_codes, dim] -> int32 [n]."""
    z = z_flat.to(dtype=torch.float32).contiguous()
    e = codebook.to(device=z.device, dtype=torch.float32).contiguous()
    idx = torch.empty((z.shape[0],), dtype=torch.int32, device=z.device)
    with torch.cuda.device(z.device):
        L.check(L.lib().vlg_codebook_argmin(L.ptr(z), L.ptr(e), z.shape[0], e.shape[0], e.shape[1], L.ptr(idx), L.stream_ptr(z.device)))
    return idx


def VQ_8(**kwargs):
    return VQModel(ModelArgs(encoder_ch_mult=[1, 2, 2, 4], decoder_ch_mult=[1, 2, 2, 4], **kwargs))


def VQ_16(**kwargs):
    return VQModel(ModelArgs(encoder_ch_mult=[1, 1, 2, 2, 4], decoder_ch_mult=[1, 1, 2, 2, 4], **kwargs))


VQ_models = {'VQ-16': VQ_16, 'VQ-8': VQ_8}
