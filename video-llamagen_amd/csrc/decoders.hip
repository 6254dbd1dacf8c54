// vlg_vq / vlg_vae handles: weight store + forward passes of the VQ-16 image decoder and the CausalVideoVAE decoder.
//
// Replaces: VQModel.decode_code / VectorQuantizer (tokenizer/tokenizer_image/vq_model.py:47-55,128-276) and
// CausalVAEModel.decode (CausalVideoVAE/causalvideovae/model/causal_vae/modeling_causalvae.py:151-262,394-404).
// Activations live channels-last [B,T,H,W,C] in the handle dtype; conv weights are re-laid out once at load time to
// [Cout][taps][Cin]; GroupNorm affine params, biases and the codebook stay fp32.
#include <algorithm>
#include <memory>

#include "conv_kernels.h"

using namespace vlg;

namespace {

struct Param {
  DevBuf buf;
  std::vector<int64_t> shape;  // conv: {Cout, Cin, kt, kh, kw}; other: source shape
  bool conv = false;
};

struct Act {
  void* p = nullptr;
  int slot = -1;
  int B = 0, T = 1, H = 0, W = 0, C = 0;
  double* gn_part = nullptr;   // GroupNorm statistics partials left by the convolution that produced this tensor (conv_forward gn_part), or null
  int gn_nblk = 0, gn_slot = -1;
  void disown() {   // a copy taken for its shape: it owns neither the buffer nor the statistics of the original
    slot = gn_slot = -1;
    gn_part = nullptr;
    gn_nblk = 0;
  }
  long long P() const { return (long long)T * H * W; }
  long long numel() const { return (long long)B * P() * C; }
};

struct Store {
  int dtype = VLG_BF16;
  size_t esz = 2;
  std::map<std::string, Param> params;
  std::vector<std::unique_ptr<DevBuf>> pool;
  std::vector<bool> busy;
  DevBuf stats, staging;
  hipStream_t st = nullptr;
  bool linear_as_conv = false;   // 2-D nn.Linear weights stored like 1x1 conv weights (tokenizer_video attention)

  bool skip(const std::string& n) const {
    return n.rfind("loss", 0) == 0 || n.find("codebook_used") != std::string::npos;   // training-only tensors
  }

  int load(const char* name, const void* data, const int64_t* shape, int ndim, int src_dtype, int on_dev, int32_t* consumed) {
    if (consumed) *consumed = 0;
    const std::string n(name);
    if (skip(n)) return VLG_OK;
    VLG_CHECK(ndim >= 1 && ndim <= 5, VLG_ERR_BAD_SHAPE, "%s: ndim %d", name, ndim);
    int64_t total = 1;
    for (int i = 0; i < ndim; ++i) total *= shape[i];
    VLG_CHECK(total > 0, VLG_ERR_BAD_SHAPE, "%s: empty tensor", name);
    Param& p = params[n];
    VLG_TRY(staging.reserve((size_t)total * sizeof(float)));
    VLG_TRY(upload_convert(staging.p, VLG_F32, data, src_dtype, on_dev, total, nullptr));
    const bool is_convt = n.find(".convt.weight") != std::string::npos;
    const bool lin = linear_as_conv && ndim == 2 && n.size() > 7 && n.compare(n.size() - 7, 7, ".weight") == 0 &&
                     n.find("embeddings") == std::string::npos;
    if (is_convt && ndim == 5) {  // ConvTranspose3d weight [Cin, Cout, k, k, k] -> [Cout][taps][Cin]
      const int Cin = (int)shape[0], Cout = (int)shape[1];
      p.shape = {Cout, Cin, shape[2], shape[3], shape[4]};
      p.conv = true;
      VLG_TRY(p.buf.reserve((size_t)total * esz));
      const int taps = (int)(shape[2] * shape[3] * shape[4]);
      if (dtype == VLG_BF16)
        VLG_TRY(relayout_convt_weight<bf16>(staging.as<float>(), p.buf.as<bf16>(), Cin, Cout, taps, nullptr));
      else
        VLG_TRY(relayout_convt_weight<float>(staging.as<float>(), p.buf.as<float>(), Cin, Cout, taps, nullptr));
      VLG_HIP(hipStreamSynchronize(nullptr));
    } else if (ndim >= 4 || lin) {  // conv weight [Cout, Cin, (kt,) kh, kw]; Linear [out, in] == 1x1 conv
      const int Cout = (int)shape[0], Cin = (int)shape[1];
      const int kt = ndim == 5 ? (int)shape[2] : 1, kh = lin ? 1 : (int)shape[ndim - 2], kw = lin ? 1 : (int)shape[ndim - 1];
      p.shape = {Cout, Cin, kt, kh, kw};
      p.conv = true;
      VLG_TRY(p.buf.reserve((size_t)total * esz));
      if (dtype == VLG_BF16)
        VLG_TRY(relayout_conv_weight<bf16>(staging.as<float>(), p.buf.as<bf16>(), Cout, Cin, kt * kh * kw, nullptr));
      else
        VLG_TRY(relayout_conv_weight<float>(staging.as<float>(), p.buf.as<float>(), Cout, Cin, kt * kh * kw, nullptr));
      VLG_HIP(hipStreamSynchronize(nullptr));
    } else {
      p.shape.assign(shape, shape + ndim);
      p.conv = false;
      VLG_TRY(p.buf.reserve((size_t)total * sizeof(float)));
      VLG_HIP(hipMemcpy(p.buf.p, staging.p, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice));
    }
    if (consumed) *consumed = 1;
    return VLG_OK;
  }

  const Param* find(const std::string& n) const {
    auto it = params.find(n);
    return it == params.end() ? nullptr : &it->second;
  }
  bool has(const std::string& n) const { return params.count(n) != 0; }

  int get(size_t bytes, Act& a) {
    int best = -1;
    for (size_t i = 0; i < pool.size(); ++i)
      if (!busy[i] && pool[i]->bytes >= bytes && (best < 0 || pool[i]->bytes < pool[best]->bytes)) best = (int)i;
    if (best < 0) {
      for (size_t i = 0; i < pool.size(); ++i)
        if (!busy[i] && (best < 0 || pool[i]->bytes > pool[best]->bytes)) best = (int)i;
      if (best < 0) {
        pool.emplace_back(new DevBuf());
        busy.push_back(false);
        best = (int)pool.size() - 1;
      } else {
        VLG_HIP(hipStreamSynchronize(st));  // buffer may still be read by queued kernels before it is re-allocated
      }
      VLG_TRY(pool[best]->reserve(bytes));
    }
    busy[best] = true;
    a.p = pool[best]->p;
    a.slot = best;
    return VLG_OK;
  }
  void put(Act& a) {
    if (a.slot >= 0) busy[a.slot] = false;
    if (a.gn_slot >= 0) busy[a.gn_slot] = false;
    a.slot = a.gn_slot = -1;
    a.p = nullptr;
    a.gn_part = nullptr;
    a.gn_nblk = 0;
  }
  void release_all() {
    for (size_t i = 0; i < busy.size(); ++i) busy[i] = false;
  }
};

template <typename T>
struct Net {
  Store& s;
  hipStream_t st;
  std::string csuf;  // ".conv" for CausalConv3d wrappers, "" for nn.Conv2d

  int tmode = 0;     // 1: SamePadConv3d (symmetric zero time pad) instead of CausalConv3d

  // stride 2 (encoders): Downsample / SpatialDownsample2x = zero pad (0,1) bottom/right only, conv k3 s2 p0
  int conv_down2(const Act& x, const std::string& name, Act& y) { return conv(x, name, 0, nullptr, y, nullptr, 2); }

  int conv(const Act& x, const std::string& name, int up, const Act* residual, Act& y, float* planar_out = nullptr, int stride = 1) {
    const Param* w = s.find(name + csuf + ".weight");
    const Param* b = s.find(name + csuf + ".bias");
    VLG_CHECK(w && w->conv, VLG_ERR_STATE, "weight %s%s.weight was never loaded", name.c_str(), csuf.c_str());
    VLG_CHECK(w->shape[1] == x.C, VLG_ERR_BAD_SHAPE, "%s: Cin %lld != activation channels %d", name.c_str(), (long long)w->shape[1], x.C);
    ConvDesc d;
    d.B = x.B; d.Ti = x.T; d.Hi = x.H; d.Wi = x.W; d.Cin = x.C;
    d.To = x.T; d.Ho = (x.H << up) / stride; d.Wo = (x.W << up) / stride; d.Cout = (int)w->shape[0];
    d.sh = stride;
    if (stride == 2) d.ph0 = d.pw0 = 0;
    d.kt = (int)w->shape[2]; d.kh = (int)w->shape[3]; d.kw = (int)w->shape[4];
    d.up = up;
    d.tmode = tmode;
    y.B = d.B; y.T = d.To; y.H = d.Ho; y.W = d.Wo; y.C = d.Cout;
    if (residual) VLG_CHECK(residual->numel() == y.numel(), VLG_ERR_BAD_SHAPE, "%s: residual shape mismatch", name.c_str());
    if (!planar_out) VLG_TRY(s.get((size_t)y.numel() * sizeof(T), y));
    // every 3x3 output of a decoder is followed by a GroupNorm (norm2, the next block's norm1, norm_out): let the convolution's epilogue
    // accumulate its statistics (no separate read pass over the tensor); tensors from other producers keep the statistics kernel
    y.gn_part = nullptr;
    y.gn_nblk = 0;
    y.gn_slot = -1;
    static const bool gnf_off = getenv("VLG_GN_FUSE") != nullptr && atoi(getenv("VLG_GN_FUSE")) == 0;   // A/B knob
    if (!gnf_off && !planar_out && d.kh == 3 && d.Cout % 128 == 0 && stride == 1) {
      Act tmp;
      VLG_TRY(s.get(conv_gn_part_doubles(d) * sizeof(double), tmp));
      y.gn_part = reinterpret_cast<double*>(tmp.p);
      y.gn_slot = tmp.slot;
    }
    VLG_TRY(conv_forward<T>(d, (const T*)x.p, w->buf.as<T>(), b ? b->buf.as<float>() : nullptr, residual ? (const T*)residual->p : nullptr,
                            planar_out ? nullptr : (T*)y.p, planar_out, st, y.gn_part, &y.gn_nblk));
    if (y.gn_part && y.gn_nblk == 0) {   // the launch that ran could not provide them
      s.busy[y.gn_slot] = false;
      y.gn_part = nullptr;
      y.gn_slot = -1;
    }
    return VLG_OK;
  }

  int gn(const Act& x, const std::string& name, bool swish, Act& y) {
    const Param* g = s.find(name + ".weight");
    const Param* b = s.find(name + ".bias");
    VLG_CHECK(g && b, VLG_ERR_STATE, "GroupNorm %s was never loaded", name.c_str());
    VLG_CHECK(g->shape[0] == x.C, VLG_ERR_BAD_SHAPE, "%s: %lld channels vs activation %d", name.c_str(), (long long)g->shape[0], x.C);
    y = x;
    y.disown();
    VLG_TRY(s.get((size_t)x.numel() * sizeof(T), y));
    VLG_TRY(s.stats.reserve(group_norm_scratch_bytes(x.B, x.P())));
    return group_norm<T>((const T*)x.p, (T*)y.p, g->buf.as<float>(), b->buf.as<float>(), s.stats.as<double>(), x.B, x.P(), x.C, 1e-6f,
                         swish, st, x.gn_part, x.gn_nblk);
  }

  // ResnetBlock (vq_model.py:299-314) / ResnetBlock3D (resnet_block.py:158-172); consumes x
  int resblock(Act& x, const std::string& p, Act& out) {
    Act h, h2, h3, xs;
    VLG_TRY(gn(x, p + ".norm1", true, h));
    VLG_TRY(conv(h, p + ".conv1", 0, nullptr, h2));
    s.put(h);
    VLG_TRY(gn(h2, p + ".norm2", true, h3));
    s.put(h2);
    const bool shortcut = s.has(p + ".nin_shortcut" + csuf + ".weight");
    if (shortcut) {
      VLG_TRY(conv(x, p + ".nin_shortcut", 0, nullptr, xs));
      s.put(x);
    } else {
      xs = x;
      x.slot = -1;
      x.gn_slot = -1;   // (the statistics buffer went with it)
    }
    VLG_TRY(conv(h3, p + ".conv2", 0, &xs, out));
    s.put(h3);
    s.put(xs);
    return VLG_OK;
  }

  // AttnBlock (vq_model.py:327-351) / AttnBlock3D incl. the Q12 reinterpretation (attention.py:52-76); consumes x
  int attn(Act& x, const std::string& p, bool q12, Act& out) {
    Act hn, q, k, v, o;
    VLG_TRY(gn(x, p + ".norm", false, hn));
    VLG_TRY(conv(hn, p + ".q", 0, nullptr, q));
    VLG_TRY(conv(hn, p + ".k", 0, nullptr, k));
    VLG_TRY(conv(hn, p + ".v", 0, nullptr, v));
    s.put(hn);
    const int HW = x.H * x.W;
    if (q12 && x.T > 1) {
      for (Act* a : {&q, &k, &v}) {
        Act tmp = *a;
        tmp.disown();
        VLG_TRY(s.get((size_t)a->numel() * sizeof(T), tmp));
        VLG_TRY(q12_permute<T>((const T*)a->p, (T*)tmp.p, x.B, x.T, HW, x.C, false, st));
        s.put(*a);
        *a = tmp;
      }
    }
    o = q;
    o.disown();
    VLG_TRY(s.get((size_t)q.numel() * sizeof(T), o));
    VLG_TRY(spatial_attention<T>((const T*)q.p, (const T*)k.p, (const T*)v.p, (T*)o.p, x.B * x.T, HW, x.C, st));
    s.put(q);
    s.put(k);
    s.put(v);
    if (q12 && x.T > 1) {
      Act tmp = o;
      tmp.disown();
      VLG_TRY(s.get((size_t)o.numel() * sizeof(T), tmp));
      VLG_TRY(q12_permute<T>((const T*)o.p, (T*)tmp.p, x.B, x.T, HW, x.C, true, st));
      s.put(o);
      o = tmp;
    }
    VLG_TRY(conv(o, p + ".proj_out", 0, &x, out));
    s.put(o);
    s.put(x);
    return VLG_OK;
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// VQ
// ---------------------------------------------------------------------------------------------------------------
struct vlg_vq {
  vlg_vq_config cfg;
  Store s;
};

extern "C" int vlg_vq_create(const vlg_vq_config* cfg, vlg_vq_t** out) {
  VLG_CHECK(cfg && out, VLG_ERR_BAD_ARG, "vlg_vq_create: null argument");
  VLG_CHECK(cfg->dtype == VLG_F32 || cfg->dtype == VLG_BF16, VLG_ERR_UNSUPPORTED, "vlg_vq_create: dtype %d", cfg->dtype);
  VLG_CHECK(cfg->n_mult >= 1 && cfg->n_mult <= 8 && cfg->codebook_size > 0 && cfg->codebook_embed_dim > 0, VLG_ERR_BAD_ARG,
            "vlg_vq_create: bad config");
  std::unique_ptr<vlg_vq> h(new vlg_vq());
  h->cfg = *cfg;
  h->s.dtype = cfg->dtype;
  h->s.esz = dtype_size(cfg->dtype);
  *out = h.release();
  return VLG_OK;
}
extern "C" int vlg_vq_destroy(vlg_vq_t* h) {
  if (h) {
    (void)hipDeviceSynchronize();
    delete h;
  }
  return VLG_OK;
}
extern "C" int vlg_vq_load_tensor(vlg_vq_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t src_dtype,
                                  int32_t on_dev, int32_t* consumed) {
  VLG_CHECK(h && name && data && shape, VLG_ERR_BAD_ARG, "vlg_vq_load_tensor: null argument");
  if (!strcmp(name, "quantize.embedding.weight"))
    VLG_CHECK(ndim == 2 && shape[0] == h->cfg.codebook_size && shape[1] == h->cfg.codebook_embed_dim, VLG_ERR_BAD_SHAPE,
              "size mismatch for quantize.embedding.weight");
  return h->s.load(name, data, shape, ndim, src_dtype, on_dev, consumed);
}

template <typename T>
static int vq_decode_impl(vlg_vq* h, const int32_t* codes, int B, int gh, int gw, float* out, hipStream_t st) {
  Store& s = h->s;
  s.st = st;
  s.release_all();
  Net<T> net{s, st, ""};
  const Param* E = s.find("quantize.embedding.weight");
  VLG_CHECK(E, VLG_ERR_STATE, "quantize.embedding.weight was never loaded");
  Act z;
  z.B = B; z.T = 1; z.H = gh; z.W = gw; z.C = h->cfg.codebook_embed_dim;
  VLG_TRY(s.get((size_t)z.numel() * sizeof(T), z));
  VLG_TRY(codebook_lookup<T>(E->buf.as<float>(), codes, (T*)z.p, (long long)B * gh * gw, h->cfg.codebook_size, z.C, h->cfg.l2_norm != 0, st));
  Act a, b;
  VLG_TRY(net.conv(z, "post_quant_conv", 0, nullptr, a));   // vq_model.py:48
  s.put(z);
  VLG_TRY(net.conv(a, "decoder.conv_in", 0, nullptr, b));    // :175
  s.put(a);
  VLG_TRY(net.resblock(b, "decoder.mid.0", a));
  VLG_TRY(net.attn(a, "decoder.mid.1", false, b));
  VLG_TRY(net.resblock(b, "decoder.mid.2", a));
  const int nres = h->cfg.n_mult;
  for (int li = 0; li < nres; ++li) {                         // :182-188 (conv_blocks are stored top level first)
    for (int j = 0; j <= h->cfg.num_res_blocks; ++j) {
      const std::string p = "decoder.conv_blocks." + std::to_string(li);
      VLG_TRY(net.resblock(a, p + ".res." + std::to_string(j), b));
      if (li == 0) {
        VLG_TRY(net.attn(b, p + ".attn." + std::to_string(j), false, a));
      } else {
        a = b;
        b.disown();
      }
    }
    if (li != nres - 1) {
      VLG_TRY(net.conv(a, "decoder.conv_blocks." + std::to_string(li) + ".upsample.conv", 1, nullptr, b));  // nearest 2x + conv (:375-377)
      s.put(a);
      a = b;
      b.disown();
    }
  }
  VLG_TRY(net.gn(a, "decoder.norm_out", true, b));            // :191-192
  s.put(a);
  Act y;
  VLG_TRY(net.conv(b, "decoder.conv_out", 0, nullptr, y, out));
  s.put(b);
  return VLG_OK;
}

extern "C" int vlg_vq_decode_code(vlg_vq_t* h, const int32_t* d_codes, int32_t B, int32_t gh, int32_t gw, float* d_out, void* stream) {
  VLG_CHECK(h && d_codes && d_out && B > 0 && gh > 0 && gw > 0, VLG_ERR_BAD_ARG, "vlg_vq_decode_code: bad argument");
  if (h->s.dtype == VLG_BF16) return vq_decode_impl<bf16>(h, d_codes, B, gh, gw, d_out, (hipStream_t)stream);
  return vq_decode_impl<float>(h, d_codes, B, gh, gw, d_out, (hipStream_t)stream);
}

extern "C" int vlg_vq_argmin(vlg_vq_t* h, const float* d_z, int32_t B, int32_t Hh, int32_t Ww, int32_t* d_idx, void* stream) {
  VLG_CHECK(h && d_z && d_idx && B > 0 && Hh > 0 && Ww > 0, VLG_ERR_BAD_ARG, "vlg_vq_argmin: bad argument");
  const Param* E = h->s.find("quantize.embedding.weight");
  VLG_CHECK(E, VLG_ERR_STATE, "quantize.embedding.weight was never loaded");
  const int C = h->cfg.codebook_embed_dim;
  const long long hw = (long long)Hh * Ww;
  // z is NCHW: row (b, pos) element c at b*C*hw + c*hw + pos   (the 'b c h w -> b h w c' of vq_model.py:217)
  return codebook_argmin(d_z, 1, hw, hw, (long long)C * hw, E->buf.as<float>(), (long long)B * hw, h->cfg.codebook_size, C,
                         h->cfg.l2_norm != 0, d_idx, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// CausalVideoVAE
// ---------------------------------------------------------------------------------------------------------------
struct vlg_vae {
  vlg_vae_config cfg;
  Store s;
  int enc_time_down[8] = {0, 1, 1, 0, 0, 0, 0, 0};   // encoder_temporal_downsample default ("", TimeDownsample2x x2, "") (:291-296)
};

extern "C" int vlg_vae_create(const vlg_vae_config* cfg, vlg_vae_t** out) {
  VLG_CHECK(cfg && out, VLG_ERR_BAD_ARG, "vlg_vae_create: null argument");
  VLG_CHECK(cfg->dtype == VLG_F32 || cfg->dtype == VLG_BF16, VLG_ERR_UNSUPPORTED, "vlg_vae_create: dtype %d", cfg->dtype);
  VLG_CHECK(cfg->n_mult >= 1 && cfg->n_mult <= 8 && cfg->embed_dim > 0 && cfg->z_channels > 0, VLG_ERR_BAD_ARG, "vlg_vae_create: bad config");
  std::unique_ptr<vlg_vae> h(new vlg_vae());
  h->cfg = *cfg;
  h->s.dtype = cfg->dtype;
  h->s.esz = dtype_size(cfg->dtype);
  *out = h.release();
  return VLG_OK;
}
extern "C" int vlg_vae_destroy(vlg_vae_t* h) {
  if (h) {
    (void)hipDeviceSynchronize();
    delete h;
  }
  return VLG_OK;
}
extern "C" int vlg_vae_load_tensor(vlg_vae_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t src_dtype,
                                   int32_t on_dev, int32_t* consumed) {
  VLG_CHECK(h && name && data && shape, VLG_ERR_BAD_ARG, "vlg_vae_load_tensor: null argument");
  return h->s.load(name, data, shape, ndim, src_dtype, on_dev, consumed);
}

extern "C" int vlg_vae_out_shape(vlg_vae_t* h, int32_t t, int32_t hh, int32_t ww, int32_t* T, int32_t* H, int32_t* W) {
  VLG_CHECK(h && T && H && W, VLG_ERR_BAD_ARG, "vlg_vae_out_shape: null argument");
  int tt = t, y = hh, x = ww;
  for (int i = h->cfg.n_mult - 1; i >= 0; --i) {
    if (h->cfg.spatial_upsample[i]) {
      y *= 2;
      x *= 2;
    }
    if (h->cfg.temporal_upsample[i] && tt > 1) tt = 2 * tt - 1;
  }
  *T = tt;
  *H = y;
  *W = x;
  return VLG_OK;
}

template <typename T>
static int vae_decode_impl(vlg_vae* h, const float* z, int B, int t, int hh, int ww, float* out, hipStream_t st) {
  Store& s = h->s;
  s.st = st;
  s.release_all();
  Net<T> net{s, st, ".conv"};
  Act a, b;
  a.B = B; a.T = t; a.H = hh; a.W = ww; a.C = h->cfg.embed_dim;
  VLG_TRY(s.get((size_t)a.numel() * sizeof(T), a));
  VLG_TRY(planar_f32_to_cl<T>(z, (T*)a.p, B, a.C, a.P(), st));
  VLG_TRY(net.conv(a, "post_quant_conv", 0, nullptr, b));     // modeling_causalvae.py:401-402
  s.put(a);
  VLG_TRY(net.conv(b, "decoder.conv_in", 0, nullptr, a));      // :244
  s.put(b);
  VLG_TRY(net.resblock(a, "decoder.mid.block_1", b));
  VLG_TRY(net.attn(b, "decoder.mid.attn_1", true, a));
  VLG_TRY(net.resblock(a, "decoder.mid.block_2", b));
  a = b;
  b.disown();
  for (int lvl = h->cfg.n_mult - 1; lvl >= 0; --lvl) {         // :249-257
    const std::string p = "decoder.up." + std::to_string(lvl);
    for (int j = 0; j <= h->cfg.num_res_blocks; ++j) {
      VLG_TRY(net.resblock(a, p + ".block." + std::to_string(j), b));
      a = b;
      b.disown();
    }
    if (h->cfg.spatial_upsample[lvl]) {
      VLG_TRY(net.conv(a, p + ".upsample.conv", 1, nullptr, b)); // SpatialUpsample2x: nearest 2x + (1,3,3) causal conv
      s.put(a);
      a = b;
      b.disown();
    }
    if (h->cfg.temporal_upsample[lvl] && a.T > 1) {
      b = a;
      b.disown();
      b.T = 2 * a.T - 1;
      VLG_TRY(s.get((size_t)b.numel() * sizeof(T), b));
      VLG_TRY(time_upsample2x<T>((const T*)a.p, (T*)b.p, a.B, a.T, (long long)a.H * a.W * a.C, st));
      s.put(a);
      a = b;
      b.disown();
    }
  }
  VLG_TRY(net.gn(a, "decoder.norm_out", true, b));
  s.put(a);
  Act y;
  VLG_TRY(net.conv(b, "decoder.conv_out", 0, nullptr, y, out));
  s.put(b);
  return VLG_OK;
}

extern "C" int vlg_vae_decode(vlg_vae_t* h, const float* d_z, int32_t B, int32_t t, int32_t hh, int32_t ww, float* d_out, void* stream) {
  VLG_CHECK(h && d_z && d_out && B > 0 && t > 0 && hh > 0 && ww > 0, VLG_ERR_BAD_ARG, "vlg_vae_decode: bad argument");
  if (h->s.dtype == VLG_BF16) return vae_decode_impl<bf16>(h, d_z, B, t, hh, ww, d_out, (hipStream_t)stream);
  return vae_decode_impl<float>(h, d_z, B, t, hh, ww, d_out, (hipStream_t)stream);
}


// ---------------------------------------------------------------------------------------------------------------
// encode side
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
static int vq_encode_impl(vlg_vq* h, const float* x_planar, int B, int Hh, int Ww, int32_t* idx, float* z_out, hipStream_t st) {
  Store& s = h->s;
  s.st = st;
  s.release_all();
  Net<T> net{s, st, ""};
  const Param* E = s.find("quantize.embedding.weight");
  VLG_CHECK(E, VLG_ERR_STATE, "quantize.embedding.weight was never loaded");
  const int nres = h->cfg.n_mult;
  VLG_CHECK(Hh % (1 << (nres - 1)) == 0 && Ww % (1 << (nres - 1)) == 0, VLG_ERR_BAD_SHAPE, "image size must be divisible by %d", 1 << (nres - 1));
  Act a, b;
  a.B = B; a.T = 1; a.H = Hh; a.W = Ww; a.C = 3;
  VLG_TRY(s.get((size_t)a.numel() * sizeof(T), a));
  VLG_TRY(planar_f32_to_cl<T>(x_planar, (T*)a.p, B, 3, (long long)Hh * Ww, st));
  VLG_TRY(net.conv(a, "encoder.conv_in", 0, nullptr, b));   // vq_model.py:106
  s.put(a);
  a = b;
  b.disown();
  for (int li = 0; li < nres; ++li) {                       // :108-114
    const std::string p = "encoder.conv_blocks." + std::to_string(li);
    for (int j = 0; j < h->cfg.num_res_blocks; ++j) {
      VLG_TRY(net.resblock(a, p + ".res." + std::to_string(j), b));
      if (li == nres - 1) {
        VLG_TRY(net.attn(b, p + ".attn." + std::to_string(j), false, a));
      } else {
        a = b;
        b.disown();
      }
    }
    if (li != nres - 1) {
      VLG_TRY(net.conv_down2(a, p + ".downsample.conv", b));
      s.put(a);
      a = b;
      b.disown();
    }
  }
  VLG_TRY(net.resblock(a, "encoder.mid.0", b));
  VLG_TRY(net.attn(b, "encoder.mid.1", false, a));
  VLG_TRY(net.resblock(a, "encoder.mid.2", b));
  VLG_TRY(net.gn(b, "encoder.norm_out", true, a));
  s.put(b);
  VLG_TRY(net.conv(a, "encoder.conv_out", 0, nullptr, b));
  s.put(a);
  VLG_TRY(net.conv(b, "quant_conv", 0, nullptr, a));          // vq_model.py:43
  s.put(b);
  // argmin over the codebook on the channels-last z (fp32 copy for the exact reference arithmetic)
  Act zf;
  zf.B = a.B; zf.T = 1; zf.H = a.H; zf.W = a.W; zf.C = a.C;
  VLG_TRY(s.get((size_t)a.numel() * sizeof(float), zf));
  const long long P = (long long)a.H * a.W;
  VLG_TRY(cl_to_planar_f32<T>((const T*)a.p, (float*)zf.p, a.B, a.C, P, st));
  if (z_out) VLG_HIP(hipMemcpyAsync(z_out, zf.p, (size_t)a.numel() * sizeof(float), hipMemcpyDeviceToDevice, st));
  VLG_TRY(codebook_argmin((const float*)zf.p, 1, P, P, (long long)a.C * P, E->buf.as<float>(), (long long)a.B * P, h->cfg.codebook_size, a.C,
                          h->cfg.l2_norm != 0, idx, st));
  s.put(a);
  s.put(zf);
  return VLG_OK;
}

extern "C" int vlg_vq_encode(vlg_vq_t* h, const float* d_x, int32_t B, int32_t Hh, int32_t Ww, int32_t* d_idx, float* d_z, void* stream) {
  VLG_CHECK(h && d_x && d_idx && B > 0 && Hh > 0 && Ww > 0, VLG_ERR_BAD_ARG, "vlg_vq_encode: bad argument");
  if (h->s.dtype == VLG_BF16) return vq_encode_impl<bf16>(h, d_x, B, Hh, Ww, d_idx, d_z, (hipStream_t)stream);
  return vq_encode_impl<float>(h, d_x, B, Hh, Ww, d_idx, d_z, (hipStream_t)stream);
}

template <typename T>
static int vae_encode_impl(vlg_vae* h, const float* x_planar, int B, int Tn, int Hh, int Ww, float* moments, hipStream_t st) {
  Store& s = h->s;
  s.st = st;
  s.release_all();
  Net<T> net{s, st, ".conv"};
  Act a, b;
  a.B = B; a.T = Tn; a.H = Hh; a.W = Ww; a.C = 3;
  VLG_TRY(s.get((size_t)a.numel() * sizeof(T), a));
  VLG_TRY(planar_f32_to_cl<T>(x_planar, (T*)a.p, B, 3, a.P(), st));
  VLG_TRY(net.conv(a, "encoder.conv_in", 0, nullptr, b));     // modeling_causalvae.py:128
  s.put(a);
  a = b;
  b.disown();
  for (int lvl = 0; lvl < h->cfg.n_mult; ++lvl) {            // :129-140
    const std::string p = "encoder.down." + std::to_string(lvl);
    for (int j = 0; j < h->cfg.num_res_blocks; ++j) {
      VLG_TRY(net.resblock(a, p + ".block." + std::to_string(j), b));
      a = b;
      b.disown();
    }
    if (s.has(p + ".downsample.conv.conv.weight")) {         // SpatialDownsample2x
      VLG_CHECK(a.H % 2 == 0 && a.W % 2 == 0, VLG_ERR_BAD_SHAPE, "odd spatial size at encoder level %d", lvl);
      VLG_TRY(net.conv_down2(a, p + ".downsample.conv", b));
      s.put(a);
      a = b;
      b.disown();
    }
    if (lvl < 8 && h->enc_time_down[lvl] && a.T > 1) {       // TimeDownsample2x
      b = a;
      b.disown();
      b.T = (a.T - 1) / 2 + 1;
      VLG_TRY(s.get((size_t)b.numel() * sizeof(T), b));
      VLG_TRY(time_downsample2x<T>((const T*)a.p, (T*)b.p, a.B, a.T, (long long)a.H * a.W * a.C, st));
      s.put(a);
      a = b;
      b.disown();
    }
  }
  VLG_TRY(net.resblock(a, "encoder.mid.block_1", b));
  VLG_TRY(net.attn(b, "encoder.mid.attn_1", true, a));
  VLG_TRY(net.resblock(a, "encoder.mid.block_2", b));
  VLG_TRY(net.gn(b, "encoder.norm_out", true, a));
  s.put(b);
  VLG_TRY(net.conv(a, "encoder.conv_out", 0, nullptr, b));
  s.put(a);
  Act y;
  VLG_TRY(net.conv(b, "quant_conv", 0, nullptr, y, moments)); // :389
  s.put(b);
  return VLG_OK;
}

extern "C" int vlg_vae_encode(vlg_vae_t* h, const float* d_x, int32_t B, int32_t T_, int32_t Hh, int32_t Ww, float* d_moments, void* stream) {
  VLG_CHECK(h && d_x && d_moments && B > 0 && T_ > 0 && Hh > 0 && Ww > 0, VLG_ERR_BAD_ARG, "vlg_vae_encode: bad argument");
  if (h->s.dtype == VLG_BF16) return vae_encode_impl<bf16>(h, d_x, B, T_, Hh, Ww, d_moments, (hipStream_t)stream);
  return vae_encode_impl<float>(h, d_x, B, T_, Hh, Ww, d_moments, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// tokenizer_video VQ-VAE decode
// ---------------------------------------------------------------------------------------------------------------
struct vlg_vqvae {
  vlg_vqvae_config cfg;
  Store s;
};

extern "C" int vlg_vqvae_create(const vlg_vqvae_config* cfg, vlg_vqvae_t** out) {
  VLG_CHECK(cfg && out, VLG_ERR_BAD_ARG, "vlg_vqvae_create: null argument");
  VLG_CHECK(cfg->dtype == VLG_F32 || cfg->dtype == VLG_BF16, VLG_ERR_UNSUPPORTED, "vlg_vqvae_create: dtype %d", cfg->dtype);
  VLG_CHECK(cfg->n_hiddens > 0 && cfg->n_head > 0 && cfg->n_hiddens % cfg->n_head == 0 && cfg->n_hiddens / cfg->n_head <= 128 &&
                cfg->n_codes > 0 && cfg->embedding_dim > 0 && cfg->n_res_layers >= 0 && cfg->n_upsample >= 1,
            VLG_ERR_BAD_ARG, "vlg_vqvae_create: bad config");
  std::unique_ptr<vlg_vqvae> h(new vlg_vqvae());
  h->cfg = *cfg;
  h->s.dtype = cfg->dtype;
  h->s.esz = dtype_size(cfg->dtype);
  h->s.linear_as_conv = true;
  *out = h.release();
  return VLG_OK;
}
extern "C" int vlg_vqvae_destroy(vlg_vqvae_t* h) {
  if (h) {
    (void)hipDeviceSynchronize();
    delete h;
  }
  return VLG_OK;
}
extern "C" int vlg_vqvae_load_tensor(vlg_vqvae_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim,
                                     int32_t src_dtype, int32_t on_dev, int32_t* consumed) {
  VLG_CHECK(h && name && data && shape, VLG_ERR_BAD_ARG, "vlg_vqvae_load_tensor: null argument");
  const std::string n(name);
  if (n.rfind("pre_vq_conv", 0) == 0 || n == "codebook.N" || n == "codebook.z_avg" || n.find("num_batches_tracked") != std::string::npos) {
    if (consumed) *consumed = 0;   // encode / EMA-training side
    return VLG_OK;
  }
  return h->s.load(name, data, shape, ndim, src_dtype, on_dev, consumed);
}

template <typename T>
static int vqvae_decode_impl(vlg_vqvae* h, const int32_t* codes, int B, int t, int hh, int ww, float* out, hipStream_t st) {
  Store& s = h->s;
  s.st = st;
  s.release_all();
  Net<T> net{s, st, ".conv"};
  net.tmode = 1;
  const int NH = h->cfg.n_hiddens, nhd = h->cfg.n_head, dk = NH / nhd;
  const Param* E = s.find("codebook.embeddings");
  VLG_CHECK(E && E->shape.size() == 2 && E->shape[0] == h->cfg.n_codes && E->shape[1] == h->cfg.embedding_dim, VLG_ERR_STATE,
            "codebook.embeddings missing or mis-shaped");
  Act z, x;
  z.B = B; z.T = t; z.H = hh; z.W = ww; z.C = h->cfg.embedding_dim;
  VLG_TRY(s.get((size_t)z.numel() * sizeof(T), z));
  VLG_TRY(codebook_lookup<T>(E->buf.as<float>(), codes, (T*)z.p, (long long)B * t * hh * ww, h->cfg.n_codes, z.C, false, st));   // vqvae.py:49
  VLG_TRY(net.conv(z, "post_vq_conv", 0, nullptr, x));                                                                          // :50
  s.put(z);
  auto bn = [&](const Act& in, const std::string& p, Act& o) -> int {
    const Param *g = s.find(p + ".weight"), *b = s.find(p + ".bias"), *rm = s.find(p + ".running_mean"), *rv = s.find(p + ".running_var");
    VLG_CHECK(g && b && rm && rv, VLG_ERR_STATE, "BatchNorm %s was never loaded", p.c_str());
    VLG_CHECK(g->shape[0] == in.C, VLG_ERR_BAD_SHAPE, "%s: channel mismatch", p.c_str());
    o = in;
    o.disown();
    VLG_TRY(s.get((size_t)in.numel() * sizeof(T), o));
    return bn_relu<T>((const T*)in.p, (T*)o.p, g->buf.as<float>(), b->buf.as<float>(), rm->buf.as<float>(), rv->buf.as<float>(),
                      (long long)in.B * in.P(), in.C, true, st);
  };
  for (int i = 0; i < h->cfg.n_res_layers; ++i) {   // AttentionResidualBlock, vqvae.py:107-125
    const std::string p = "decoder.res_stack." + std::to_string(i) + ".block.";
    Act a, b2, c;
    VLG_TRY(bn(x, p + "0", a));
    VLG_TRY(net.conv(a, p + "2", 0, nullptr, b2));
    s.put(a);
    VLG_TRY(bn(b2, p + "3", a));
    s.put(b2);
    VLG_TRY(net.conv(a, p + "5", 0, nullptr, b2));
    s.put(a);
    VLG_TRY(bn(b2, p + "6", c));
    s.put(b2);
    // AxialBlock (vqvae.py:89-104): attn_w + attn_h + attn_t, each MultiHeadAttention(q = k = v = c)
    Act outs[3];
    const char* names[3] = {"attn_w", "attn_h", "attn_t"};
    const int axes[3] = {3, 2, 1};
    Net<T> lin{s, st, ""};
    for (int k = 0; k < 3; ++k) {
      const std::string q = p + "8." + names[k] + ".";
      Act qq, kk, vv, oo;
      VLG_TRY(lin.conv(c, q + "w_qs", 0, nullptr, qq));
      VLG_TRY(lin.conv(c, q + "w_ks", 0, nullptr, kk));
      VLG_TRY(lin.conv(c, q + "w_vs", 0, nullptr, vv));
      oo = qq;
      oo.disown();
      VLG_TRY(s.get((size_t)qq.numel() * sizeof(T), oo));
      VLG_TRY(axial_attention<T>((const T*)qq.p, (const T*)kk.p, (const T*)vv.p, (T*)oo.p, c.B, c.T, c.H, c.W, nhd, dk, axes[k], st));
      s.put(qq);
      s.put(kk);
      s.put(vv);
      VLG_TRY(lin.conv(oo, q + "fc", 0, nullptr, outs[k]));
      s.put(oo);
    }
    s.put(c);
    Act nx = x;
    nx.disown();
    VLG_TRY(s.get((size_t)x.numel() * sizeof(T), nx));
    VLG_TRY(add4<T>((const T*)x.p, (const T*)outs[0].p, (const T*)outs[1].p, (const T*)outs[2].p, (T*)nx.p, x.numel(), st));
    for (auto& o : outs) s.put(o);
    s.put(x);
    x = nx;
  }
  Act y;
  VLG_TRY(bn(x, "decoder.res_stack." + std::to_string(h->cfg.n_res_layers), y));
  s.put(x);
  for (int i = 0; i < h->cfg.n_upsample; ++i) {   // vqvae.py:266-272
    const std::string p = "decoder.convts." + std::to_string(i) + ".convt";
    const Param *w = s.find(p + ".weight"), *b = s.find(p + ".bias");
    VLG_CHECK(w && w->conv && b, VLG_ERR_STATE, "%s was never loaded", p.c_str());
    VLG_CHECK(w->shape[1] == y.C && w->shape[2] == 4, VLG_ERR_BAD_SHAPE, "%s: shape mismatch", p.c_str());
    const bool last = i == h->cfg.n_upsample - 1;
    Act o;
    o.B = y.B; o.T = 2 * y.T; o.H = 2 * y.H; o.W = 2 * y.W; o.C = (int)w->shape[0];
    if (!last) VLG_TRY(s.get((size_t)o.numel() * sizeof(T), o));
    VLG_TRY(conv_transpose_k4s2<T>((const T*)y.p, w->buf.as<T>(), b->buf.as<float>(), last ? nullptr : (T*)o.p, last ? out : nullptr, y.B, y.T,
                                   y.H, y.W, y.C, o.C, !last, st));
    s.put(y);
    y = o;
  }
  return VLG_OK;
}

extern "C" int vlg_vqvae_decode(vlg_vqvae_t* h, const int32_t* d_codes, int32_t B, int32_t t, int32_t hh, int32_t ww, float* d_out,
                                void* stream) {
  VLG_CHECK(h && d_codes && d_out && B > 0 && t > 0 && hh > 0 && ww > 0, VLG_ERR_BAD_ARG, "vlg_vqvae_decode: bad argument");
  if (h->s.dtype == VLG_BF16) return vqvae_decode_impl<bf16>(h, d_codes, B, t, hh, ww, d_out, (hipStream_t)stream);
  return vqvae_decode_impl<float>(h, d_codes, B, t, hh, ww, d_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Spatial tile compositing for the tiled VAE passes (CausalVAEModel.tiled_decode2d / tiled_encode2d with blend_v / blend_h,
// modeling_causalvae.py:424-443,491-570).  Tiles are visited in raster order; each one is cross-faded IN PLACE against the
// finished tile above it (over its first rows) and then against the finished tile to its left (over its first columns), so that
// later neighbours fade against the already-faded values exactly as the reference's in-place loops do; the kept top-left
// keep_h x keep_w part lands in the output canvas in the same pass.  One thread per element, no inter-thread dependence:
// an element only needs its own old value plus one element of each neighbour tile.
//   weights: (float)(1 - y / e) and (float)(y / e) from double arithmetic, products and sum rounded separately (no FMA) - the
//   rounding sequence of `a * (1 - y / e) + b * (y / e)` on fp32 tensors with Python-float scalars.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void tile_blend_kernel(float* __restrict__ tile, const float* __restrict__ above, const float* __restrict__ left,
                                                         int64_t planes, int th, int tw, int ah, int lw, int ev, int eh, float* __restrict__ canvas,
                                                         int ch, int cw, int y0, int x0, int keep_h, int keep_w) {
  const int64_t n = planes * th * tw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % tw);
    const int y = (int)((i / tw) % th);
    const int64_t pl = i / ((int64_t)tw * th);
    float v = tile[i];
    if (above && y < ev) {
      const double f = (double)y / (double)ev;
      const float wa = (float)(1.0 - f), wb = (float)f;
      v = __fadd_rn(__fmul_rn(above[(pl * ah + (ah - ev + y)) * tw + x], wa), __fmul_rn(v, wb));
    }
    if (left && x < eh) {
      // the left neighbour's value may itself have been faded against ITS upper neighbour: it is read from the finished tile
      const double f = (double)x / (double)eh;
      const float wa = (float)(1.0 - f), wb = (float)f;
      v = __fadd_rn(__fmul_rn(left[(pl * th + y) * lw + (lw - eh + x)], wa), __fmul_rn(v, wb));
    }
    tile[i] = v;
    if (y < keep_h && x < keep_w) canvas[(pl * ch + (y0 + y)) * cw + (x0 + x)] = v;
  }
}
}  // namespace

extern "C" int vlg_tile_blend(float* d_tile, const float* d_above, const float* d_left, int64_t planes, int32_t th, int32_t tw, int32_t above_h,
                              int32_t left_w, int32_t extent, float* d_canvas, int32_t canvas_h, int32_t canvas_w, int32_t y0, int32_t x0,
                              int32_t keep_h, int32_t keep_w, void* stream) {
  VLG_CHECK(d_tile && d_canvas && planes > 0 && th > 0 && tw > 0 && extent >= 0, VLG_ERR_BAD_ARG, "vlg_tile_blend: bad argument");
  VLG_CHECK(!d_above || above_h > 0, VLG_ERR_BAD_ARG, "vlg_tile_blend: above_h must be positive when d_above is given");
  VLG_CHECK(!d_left || left_w > 0, VLG_ERR_BAD_ARG, "vlg_tile_blend: left_w must be positive when d_left is given");
  const int kh = std::min(keep_h, th), kw = std::min(keep_w, tw);
  VLG_CHECK(y0 >= 0 && x0 >= 0 && kh >= 0 && kw >= 0 && y0 + kh <= canvas_h && x0 + kw <= canvas_w, VLG_ERR_BAD_SHAPE,
            "vlg_tile_blend: kept region [%d+%d, %d+%d] exceeds the %d x %d canvas", y0, kh, x0, kw, canvas_h, canvas_w);
  const int ev = d_above ? std::min(std::min(above_h, th), extent) : 0;
  const int eh = d_left ? std::min(std::min(left_w, tw), extent) : 0;
  const int64_t n = planes * th * tw;
  const int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 4096);
  tile_blend_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(d_tile, ev > 0 ? d_above : nullptr, eh > 0 ? d_left : nullptr, planes, th, tw, above_h,
                                                             left_w, ev, eh, d_canvas, canvas_h, canvas_w, y0, x0, kh, kw);
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Unit entry points of the decoder kernels (per-op parity against the reference's CausalConv3d,
// SpatialUpsample2x / SpatialDownsample2x convolutions, Normalize + swish, TimeUpsample2x).  They run the very kernels the handles
// use on caller-provided planar fp32 tensors in the reference's layouts; scratch is a process-wide grow-only pool, so these
// calls are for tests and tools (one host thread), not for the hot path.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct UnitScratch {
  DevBuf xcl, wcl, ycl, stats;
};
UnitScratch& unit_scratch() {
  static UnitScratch s;
  return s;
}

template <typename T>
int unit_conv(const float* x, const float* w, const float* b, int B, int Cin, int T_, int H, int W, int Cout, int kt, int kh, int kw, int stride,
              int up, float* out, hipStream_t st) {
  UnitScratch& u = unit_scratch();
  const long long P = (long long)T_ * H * W;
  const int taps = kt * kh * kw;
  VLG_HIP(hipStreamSynchronize(st));
  VLG_TRY(u.xcl.reserve((size_t)B * P * Cin * sizeof(T)));
  VLG_TRY(u.wcl.reserve((size_t)Cout * taps * Cin * sizeof(T)));
  VLG_TRY(planar_f32_to_cl<T>(x, u.xcl.as<T>(), B, Cin, P, st));
  VLG_TRY(relayout_conv_weight<T>(w, u.wcl.as<T>(), Cout, Cin, taps, st));
  ConvDesc d;
  d.B = B; d.Ti = T_; d.Hi = H; d.Wi = W; d.Cin = Cin;
  d.To = T_; d.Ho = (H << up) / stride; d.Wo = (W << up) / stride; d.Cout = Cout;
  d.kt = kt; d.kh = kh; d.kw = kw;
  d.up = up;
  d.sh = stride;
  if (stride == 2) d.ph0 = d.pw0 = 0;      // SpatialDownsample2x: zero pad (0,1) bottom / right only (updownsample.py:106-121)
  return conv_forward<T>(d, u.xcl.as<T>(), u.wcl.as<T>(), b, nullptr, nullptr, out, st);
}

template <typename T>
int unit_gn(const float* x, const float* gamma, const float* beta, int B, int C, long long P, float eps, bool swish, float* out, hipStream_t st) {
  UnitScratch& u = unit_scratch();
  VLG_HIP(hipStreamSynchronize(st));
  VLG_TRY(u.xcl.reserve((size_t)B * P * C * sizeof(T)));
  VLG_TRY(u.ycl.reserve((size_t)B * P * C * sizeof(T)));
  VLG_TRY(u.stats.reserve(group_norm_scratch_bytes(B, P)));
  VLG_TRY(planar_f32_to_cl<T>(x, u.xcl.as<T>(), B, C, P, st));
  VLG_TRY(group_norm<T>(u.xcl.as<T>(), u.ycl.as<T>(), gamma, beta, u.stats.as<double>(), B, P, C, eps, swish, st));
  return cl_to_planar_f32<T>(u.ycl.as<T>(), out, B, C, P, st);
}

template <typename T>
int unit_timeup(const float* x, int B, int C, int T_, long long HW, float* out, hipStream_t st) {
  UnitScratch& u = unit_scratch();
  VLG_HIP(hipStreamSynchronize(st));
  const int To = T_ > 1 ? 2 * T_ - 1 : 1;
  VLG_TRY(u.xcl.reserve((size_t)B * T_ * HW * C * sizeof(T)));
  VLG_TRY(u.ycl.reserve((size_t)B * To * HW * C * sizeof(T)));
  VLG_TRY(planar_f32_to_cl<T>(x, u.xcl.as<T>(), B, C, (long long)T_ * HW, st));
  if (T_ > 1) {
    VLG_TRY(time_upsample2x<T>(u.xcl.as<T>(), u.ycl.as<T>(), B, T_, HW * C, st));
    return cl_to_planar_f32<T>(u.ycl.as<T>(), out, B, C, (long long)To * HW, st);
  }
  return cl_to_planar_f32<T>(u.xcl.as<T>(), out, B, C, HW, st);
}
}  // namespace

extern "C" int vlg_causal_conv3d(const float* d_x, const float* d_w, const float* d_bias, int32_t B, int32_t Cin, int32_t T_, int32_t H, int32_t W,
                                 int32_t Cout, int32_t kt, int32_t kh, int32_t kw, int32_t stride_hw, int32_t nearest_up, int32_t dtype,
                                 float* d_out, void* stream) {
  VLG_CHECK(d_x && d_w && d_out && B > 0 && Cin > 0 && T_ > 0 && H > 0 && W > 0 && Cout > 0, VLG_ERR_BAD_ARG, "vlg_causal_conv3d: bad argument");
  VLG_CHECK((kt == 1 || kt == 3) && (kh == 1 || kh == 3) && kh == kw, VLG_ERR_UNSUPPORTED, "vlg_causal_conv3d: kernel %dx%dx%d", kt, kh, kw);
  VLG_CHECK((stride_hw == 1 || stride_hw == 2) && (nearest_up == 0 || nearest_up == 1) && !(stride_hw == 2 && nearest_up), VLG_ERR_UNSUPPORTED,
            "vlg_causal_conv3d: stride %d / upsample %d", stride_hw, nearest_up);
  if (stride_hw == 2) VLG_CHECK(kh == 3 && H % 2 == 0 && W % 2 == 0, VLG_ERR_BAD_SHAPE, "vlg_causal_conv3d: stride 2 needs a 3x3 kernel and even H, W");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VLG_BF16) return unit_conv<bf16>(d_x, d_w, d_bias, B, Cin, T_, H, W, Cout, kt, kh, kw, stride_hw, nearest_up, d_out, st);
  if (dtype == VLG_F32) return unit_conv<float>(d_x, d_w, d_bias, B, Cin, T_, H, W, Cout, kt, kh, kw, stride_hw, nearest_up, d_out, st);
  set_error("vlg_causal_conv3d: dtype %d", dtype);
  return VLG_ERR_UNSUPPORTED;
}

extern "C" int vlg_group_norm(const float* d_x, const float* d_gamma, const float* d_beta, int32_t B, int32_t C, int64_t P, float eps, int32_t swish,
                              int32_t dtype, float* d_out, void* stream) {
  VLG_CHECK(d_x && d_gamma && d_beta && d_out && B > 0 && C > 0 && C % 32 == 0 && P > 0, VLG_ERR_BAD_ARG, "vlg_group_norm: bad argument (32 groups)");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VLG_BF16) return unit_gn<bf16>(d_x, d_gamma, d_beta, B, C, P, eps, swish != 0, d_out, st);
  if (dtype == VLG_F32) return unit_gn<float>(d_x, d_gamma, d_beta, B, C, P, eps, swish != 0, d_out, st);
  set_error("vlg_group_norm: dtype %d", dtype);
  return VLG_ERR_UNSUPPORTED;
}

extern "C" int vlg_time_upsample2x(const float* d_x, int32_t B, int32_t C, int32_t T_, int64_t HW, int32_t dtype, float* d_out, void* stream) {
  VLG_CHECK(d_x && d_out && B > 0 && C > 0 && T_ > 0 && HW > 0, VLG_ERR_BAD_ARG, "vlg_time_upsample2x: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VLG_BF16) return unit_timeup<bf16>(d_x, B, C, T_, HW, d_out, st);
  if (dtype == VLG_F32) return unit_timeup<float>(d_x, B, C, T_, HW, d_out, st);
  set_error("vlg_time_upsample2x: dtype %d", dtype);
  return VLG_ERR_UNSUPPORTED;
}

extern "C" int vlg_conv_timing(int32_t enable) { return conv_timing_enable(enable != 0); }
extern "C" int vlg_conv_timing_read(double* ms_sum, double* flop_sum, int64_t* launches) {
  long long n = 0;
  VLG_TRY(conv_timing_read(ms_sum, flop_sum, &n));
  if (launches) *launches = n;
  return VLG_OK;
}
