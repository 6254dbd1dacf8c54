"""Host mirror of the VideoGPT-style VQ-VAE's decode side (tokenizer/tokenizer_video/vqvae.py:17-86):
`VQVAE(args).decode(encodings[B,t,h,w]) -> [B,3,4t,4h,4w]`, and the `Codebook` nearest neighbour (`encode_indices`).
Never wired to a GPT in the reference; provided for completeness of the tokenizer decode path."""
import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib as L
from .vq_model import codebook_argmin


class Codebook:
    """Codebook in eval mode (tokenizer/tokenizer_video/vqvae.py:130-213 == CausalVideoVAE/causalvideovae/model/modules/quant.py:8-100):
    `forward(z[b,c,t,h,w]) -> dict(embeddings, encodings, commitment_loss, perplexity)`, `dictionary_lookup(encodings)`.  The EMA
    update and the data-dependent initialisation only exist in training and are not part of this path."""

    def __init__(self, n_codes, embedding_dim):
        self.n_codes, self.embedding_dim = n_codes, embedding_dim
        self.embeddings = None
        self.training = False

    def eval(self):
        return self

    def to(self, device=None, dtype=None):
        if self.embeddings is not None and device is not None and not isinstance(device, torch.dtype):
            self.embeddings = self.embeddings.to(device)
        return self

    def load_state_dict(self, state_dict, strict=True):
        e = state_dict["embeddings"].detach().to(torch.float32)
        if tuple(e.shape) != (self.n_codes, self.embedding_dim):
            raise L.VlgError(-2, "size mismatch for embeddings: %s vs (%d, %d)" % (tuple(e.shape), self.n_codes, self.embedding_dim))
        self.embeddings = e.to("cuda").contiguous()
        return [], [k for k in state_dict if k != "embeddings"]      # N / z_avg: EMA statistics, training only

    @torch.no_grad()
    def forward(self, z):
        if self.embeddings is None:
            raise L.VlgError(-6, "embeddings was never loaded")
        if z.dim() != 5 or z.shape[1] != self.embedding_dim:
            raise L.VlgError(-2, "z must be [b, %d, t, h, w], got %s" % (self.embedding_dim, tuple(z.shape)))
        dev = self.embeddings.device
        zf = z.to(dev, torch.float32).contiguous()
        B, Cc = int(z.shape[0]), int(z.shape[1])
        n_pos = int(z.shape[2] * z.shape[3] * z.shape[4])
        enc = torch.empty((B,) + tuple(z.shape[2:]), dtype=torch.int32, device=dev)
        emb = torch.empty_like(zf)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().vlg_codebook_forward(L.ptr(zf), L.ptr(self.embeddings), B, Cc, C.c_int64(n_pos), self.n_codes, L.ptr(enc), L.ptr(emb),
                                                 L.ptr(stats), L.stream_ptr(dev)))
        return dict(embeddings=emb, encodings=enc.long(), commitment_loss=stats[0], perplexity=stats[1])

    __call__ = forward

    def dictionary_lookup(self, encodings):
        return torch.nn.functional.embedding(encodings.to(self.embeddings.device), self.embeddings)


class VQVAE:
    def __init__(self, args=None, **kw):
        a = dict(embedding_dim=256, n_codes=2048, n_hiddens=240, n_res_layers=4, downsample=(4, 4, 4))   # vqvae.py:78-86
        if args is not None:
            a.update({k: getattr(args, k) for k in a if hasattr(args, k)})
        a.update(kw)
        self.args = SimpleNamespace(**a)
        ups = {int(d).bit_length() - 1 for d in self.args.downsample}
        if len(ups) != 1:
            raise L.VlgError(-3, "anisotropic downsample %s is not supported" % (self.args.downsample,))
        self.n_upsample = ups.pop()
        self.embedding_dim, self.n_codes = self.args.embedding_dim, self.args.n_codes
        self._dtype = torch.bfloat16
        self._device = None
        self._handle = None
        self._codebook = None

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
        a = self.args
        cfg = L.VqvaeConfig(n_hiddens=a.n_hiddens, embedding_dim=a.embedding_dim, n_codes=a.n_codes, n_res_layers=a.n_res_layers,
                            n_head=2, n_upsample=self.n_upsample, dtype=L.torch_dtype_code(self._dtype))
        h = C.c_void_p()
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vqvae_create(C.byref(cfg), C.byref(h)))
        self._handle = h

    def load_state_dict(self, state_dict, strict=True):
        self._ensure_handle()
        skipped = []
        with torch.cuda.device(self._device):
            for k, v in state_dict.items():
                if k == "codebook.embeddings":
                    self._codebook = v.detach().to(self._device, torch.float32).contiguous()
                if not L.load_tensor(L.lib().vlg_vqvae_load_tensor, self._handle, k, v):
                    skipped.append(k)
        return [], skipped

    @torch.no_grad()
    def decode(self, encodings):
        """encodings int [B,t,h,w] -> float32 [B,3,t*2^n,h*2^n,w*2^n] (vqvae.py:48-51)."""
        self._ensure_handle()
        if encodings.dim() != 4:
            raise L.VlgError(-2, "encodings must be [B,t,h,w]")
        B, t, hh, ww = [int(s) for s in encodings.shape]
        codes = encodings.to(device=self._device, dtype=torch.int32).contiguous()
        f = 2 ** self.n_upsample
        out = torch.empty((B, 3, t * f, hh * f, ww * f), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_vqvae_decode(self._handle, L.ptr(codes), B, t, hh, ww, L.ptr(out), L.stream_ptr(self._device)))
        return out

    @torch.no_grad()
    def encode_indices(self, z):
        """Codebook.forward argmin (vqvae.py:161-170): z [B,C,t,h,w] -> int32 [B,t,h,w]."""
        if self._codebook is None:
            raise L.VlgError(-6, "codebook.embeddings was never loaded")
        B, Cc, t, hh, ww = z.shape
        flat = z.to(self._device, torch.float32).permute(0, 2, 3, 4, 1).reshape(-1, Cc).contiguous()
        return codebook_argmin(flat, self._codebook).view(B, t, hh, ww)

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().vlg_vqvae_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
