// vlg_gpt handle: weights, KV cache, workspaces and the prefill + decode loop (HIP-graph replayed).
//
// Replaces: Transformer (autoregressive/models/gpt.py:262-371; t2v: gpt_video.py:270-431) and generate()
// (autoregressive/models/generate.py:127-180; t2v skeleton generate_video_diff.py:185-228).
//
// HBM layout (all in the handle dtype T unless noted):
//   weights        one allocation per state-dict tensor, [out, in] row-major exactly as nn.Linear stores them;
//                  w1 and w3 share one [2F, D] tensor so SwiGLU's two projections are one weight stream.
//   KV cache       [L][Bp][H][S][hd] for K and for V; S = roundup8(T + N) (gpt.py:322).  Only rows 0..p are ever read.
//   freqs          fp32 [cls + vae_t*g*g, hd/2, 2]; the first cls rows are zero (Q1).
//   slabs          fp32 [splits][M][N] split-K partial sums, consumed by the next kernel.
//   StepState      {pos, step} in device memory, bumped by a 1-thread kernel so the decode step graph is replayable.
#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>

#include "conv_kernels.h"
#include "gpt_kernels.h"

using namespace vlg;

struct Lane {
  DevBuf kcache, vcache, ws, attn_ws, x, xn, q, ao, g, t1, condT, hl, latT, y, logits, state, cur_tok, cur_lat;
  DevBuf d_cemb, d_ys, d_mod, d_h, d_g, d_g1, d_out, d_x, d_x2;   // DiffLoss head (d_ys / d_mod hold all S steps on the fused path)
  DevBuf dp_xbuf;                                           // persistent DiffLoss sampler: exchange buffer
  DevBuf pd_xbuf;                                           // persistent decode step: hand-off granules (zeroed when a generate() call starts)
  DevBuf maskbuf;                                           // this lane's rows of the caller's emb_mask (stable address for the cached graph)
  std::vector<uint64_t> ptr_key() const {                   // every address a captured decode step can hold
    std::vector<uint64_t> k;
    for (const DevBuf* b : {&kcache, &vcache, &ws, &attn_ws, &x, &xn, &q, &ao, &g, &t1, &condT, &hl, &latT, &y, &logits, &state, &cur_tok,
                            &cur_lat, &d_cemb, &d_ys, &d_mod, &d_h, &d_g, &d_g1, &d_out, &d_x, &d_x2, &maskbuf, &dp_xbuf, &pd_xbuf})
      k.push_back((uint64_t)(uintptr_t)b->p);
    return k;
  }
  hipStream_t st = nullptr;
  hipEvent_t ev = nullptr;
  ~Lane() {
    if (st) (void)hipStreamDestroy(st);
    if (ev) (void)hipEventDestroy(ev);
  }
};

struct vlg_gpt {
  vlg_gpt_config cfg;
  int D, H, hd, L, F, V, Tc, C, cd, vae_t, grid, npos;
  int dtype;
  size_t esz;
  struct Spec {
    std::vector<int64_t> shape;  // expected source shape
    std::string target;          // tensor in `w`
    int64_t row_off;             // row offset inside target (w1/w3 merge)
  };
  std::map<std::string, Spec> specs;
  std::map<std::string, Tensor> w;
  DevBuf freqs;
  // DiffLoss head (VLG_HEAD_HIDDEN)
  int dW = 0, dDepth = 0, dS = 0;
  std::vector<DdpmCoef> dcoef;       // per respaced step
  std::vector<float> dsincos;        // [S][256] timestep embedding inputs
  DevBuf dtemb;                      // [S][W] time_embed(t) table, handle dtype
  DevBuf dadaln_bias;                // fp32 copy of diffloss.adaln_all.bias (bias operand of the batched modulation GEMM)
  DevBuf dcoef_dev;                  // DdpmCoef[S] on the device (persistent sampler)
  int kv_block = 0;                  // sessions: positions per KV block (0 = one contiguous slot of max length per row)
  int kv_pool_blocks = 0;            //           blocks in the pool, scratch block included (0 = enough for every row at full length)
  float cfg_iter = 1.0f;             // DiffLoss.sample's cfg (generate_video_diff.py:89-91): != 1 pairs rows b and b + B/2 inside the sampler
  unsigned* fault_host = nullptr;    // pinned, device-visible fault word(s) of the persistent kernels (vlg_gpt_status)
  unsigned* fault_dev = nullptr;
  int spin_max = 0;                  // option debug_spin_max (0 = default bound)
  bool pdecode = true;               // decode layers as one persistent launch per step (pdecode.hip) where the shape allows
  int pd_rows = 0;                   // ... up to this many cache rows (0 = the measured rule of pd_use())
  bool weights_fm = true;            // stream the fragment-major weight copies (option "weights_fm"; results are bit-identical either way)
  bool act_fm = true;                // keep the fused decode chain's activations A-fragment-major (option "act_fm"; bit-identical either way)
  const int32_t* teach_ids = nullptr;   // vlg_gpt_set_teacher: forced inputs [B][N] (token heads) ...
  const float* teach_lat = nullptr;     //   ... or [B][N][C] fp32 (latent heads); caller-owned device memory
  int pos_offset = 0;                // benchmarks ("debug_pos_offset"): decode as if this many tokens had already been generated (zeroed cache rows)
  DevBuf pd_layers_dev;              // PdLayer[L]: weight pointers of every layer for the persistent kernel
  bool pd_fm = false;                // ... which are the fragment-major copies
  bool dl_persist_on = true;         // DiffLoss.sample as one persistent launch per token (diffloss_persist.hip) where the shape allows
  int dl_rows = 0;                   // ... its rows per group: 0 = chosen by the batch, 4 / 8 forced (option dl_persist = 4 / 8)
  bool dtemb_ready = false;

  // per-generate state: activations, KV cache, step state and the stream the decode loop runs on
  Lane lane;
  struct Session;                    // iteration-level batching (vlg_gpt_session_*)
  std::unique_ptr<Session> ses;
  hipStream_t s_int = nullptr;   // weight uploads
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  bool use_graph = true;
  // The instantiated decode-step graph of the last generate(), reused while every value and address baked into it is unchanged
  // (shape, sampling parameters, options, lane buffers, noise / trace pointers).  Outputs go through `outbuf` (handle-owned, stable
  // address) and are copied to the caller's buffer on the caller's stream at the end, so fresh output tensors do not invalidate it.
  struct GraphCache {
    std::vector<uint64_t> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t stream = nullptr;   // the stream its launches were enqueued on
    void drop() {
      if (stream) (void)hipStreamSynchronize(stream);
      if (exec) (void)hipGraphExecDestroy(exec);
      if (graph) (void)hipGraphDestroy(graph);
      exec = nullptr;
      graph = nullptr;
      key.clear();
    }
  } gc;
  DevBuf outbuf;                     // [B, N] int32 ids or [B, N, C] fp32 latents of the running call
  long long graphs_built = 0;        // instantiations so far (tests: a repeated call must not add one)
  long long pd_steps = 0, chain_steps = 0;   // decode steps recorded on the persistent / the per-layer path (vlg_gpt_counter)
  bool fuse_gemm = true;             // decode: fused skinny GEMMs (RMSNorm prologue; residual / RoPE+scatter / SwiGLU epilogues)
  bool fuse_swiglu = true;           // w1/w3 GEMM with the SiLU*mul epilogue
  bool time_attn = false;            // eager decode loop with HIP events around layer 0's split-KV attention kernel
  std::vector<hipEvent_t> attn_ev;   // 2 per decode step
  double attn_ms_sum = 0, attn_bytes_sum = 0;
  double attn_pair_overhead_ms = 0;   // mean elapsed time of an EMPTY event pair on the launch stream (calibration of the bracket itself)
  long long attn_launches = 0;
  double bytes_w = 0, bytes_kv = 0, bytes_other = 0;

  ~vlg_gpt() {
    gc.drop();
    if (s_int) (void)hipStreamDestroy(s_int);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    if (fault_host) (void)hipHostFree(fault_host);
    for (auto e : attn_ev) (void)hipEventDestroy(e);
  }
  const void* W(const std::string& n) const { return w.at(n).buf.p; }
  const void* Wfm(const std::string& n) const {   // fragment-major copy, or null where the shape does not tile / it was never built
    const Tensor& t = w.at(n);
    return t.fm_stale ? nullptr : t.fm.p;
  }
};

// create_diffusion(timestep_respacing=str(n), noise_schedule="cosine", learn_sigma=True): diffusion/__init__.py:11-47,
// gaussian_diffusion.py:116-150,153-202, respace.py:41-82 - float64 on the host like the reference.
static void build_diffusion_schedule(int n, std::vector<DdpmCoef>& coef, std::vector<float>& sincos) {
  const int T = 1000;
  auto ab = [](double t) {
    const double c = cos((t + 0.008) / 1.008 * M_PI / 2);
    return c * c;
  };
  std::vector<double> acp(T);
  double prod = 1.0;
  for (int i = 0; i < T; ++i) {
    double beta = 1 - ab((double)(i + 1) / T) / ab((double)i / T);
    if (beta > 0.999) beta = 0.999;
    prod *= (1.0 - beta);
    acp[i] = prod;
  }
  std::vector<char> use(T, 0);
  const double frac = n <= 1 ? 1.0 : (double)(T - 1) / (n - 1);
  double cur = 0.0;
  for (int k = 0; k < n; ++k) {
    use[(int)nearbyint(cur)] = 1;   // Python round(): half to even, as nearbyint in the default rounding mode
    cur += frac;
  }
  std::vector<double> b;
  std::vector<int> tmap;
  double last = 1.0;
  for (int i = 0; i < T; ++i)
    if (use[i]) {
      b.push_back(1 - acp[i] / last);
      last = acp[i];
      tmap.push_back(i);
    }
  const int S = (int)b.size();
  std::vector<double> ac(S), acprev(S), pv(S);
  double pr = 1.0;
  for (int i = 0; i < S; ++i) {
    acprev[i] = pr;
    pr *= (1.0 - b[i]);
    ac[i] = pr;
    pv[i] = b[i] * (1.0 - acprev[i]) / (1.0 - ac[i]);
  }
  coef.resize(S);
  for (int i = 0; i < S; ++i) {
    DdpmCoef c;
    c.sqrt_recip = (float)sqrt(1.0 / ac[i]);
    c.sqrt_recipm1 = (float)sqrt(1.0 / ac[i] - 1);
    c.coef1 = (float)(b[i] * sqrt(acprev[i]) / (1.0 - ac[i]));
    c.coef2 = (float)((1.0 - acprev[i]) * sqrt(1.0 - b[i]) / (1.0 - ac[i]));
    c.min_log = (float)log(S > 1 ? pv[i == 0 ? 1 : i] : pv[0]);
    c.max_log = (float)log(b[i]);
    c.nonzero = i != 0;
    coef[i] = c;
  }
  // TimestepEmbedder.timestep_embedding (diffloss.py:73-91): [cos(t f) | sin(t f)], f = exp(-ln(1e4) k / 128), float math
  sincos.assign((size_t)S * 256, 0.f);
  for (int i = 0; i < S; ++i)
    for (int k = 0; k < 128; ++k) {
      const float f = expf(-logf(10000.0f) * (float)k / 128.0f);
      const float a = (float)tmap[i] * f;
      sincos[(size_t)i * 256 + k] = cosf(a);
      sincos[(size_t)i * 256 + 128 + k] = sinf(a);
    }
}

static int ffn_hidden(int dim, int multiple_of) {
  int hidden = 4 * dim;
  hidden = (int)(2 * (long long)hidden / 3);
  return hidden % multiple_of == 0 ? hidden : hidden + multiple_of - (hidden % multiple_of);
}

// RoPE table, gpt.py:407-420 / gpt_video.py:532-552 (float math as torch does on CPU)
static void rope_table_host(int grid, int vae_t, int hd, float base, int cls, std::vector<float>& out) {
  const int half = hd / 2;
  const int nf = half / 2;  // arange(0, half, 2)[: half//2]
  std::vector<float> freqs(nf);
  for (int i = 0; i < nf; ++i) freqs[i] = 1.0f / powf(base, (float)(2 * i) / (float)half);
  const int npos = cls + vae_t * grid * grid;
  out.assign((size_t)npos * half * 2, 0.f);
  for (int tt = 0; tt < vae_t; ++tt)
    for (int r = 0; r < grid; ++r)
      for (int c = 0; c < grid; ++c) {
        float* row = out.data() + ((size_t)cls + ((size_t)tt * grid + r) * grid + c) * half * 2;
        for (int i = 0; i < nf; ++i) {
          const float fr = (float)r * freqs[i], fc = (float)c * freqs[i];
          row[2 * i] = cosf(fr);
          row[2 * i + 1] = sinf(fr);
          row[2 * (nf + i)] = cosf(fc);
          row[2 * (nf + i) + 1] = sinf(fc);
        }
      }
}

extern "C" int vlg_rope_table(int32_t grid, int32_t vae_t, int32_t head_dim, float base, int32_t cls, float* host_out) {
  VLG_CHECK(host_out && grid > 0 && vae_t > 0 && head_dim > 0 && head_dim % 4 == 0 && cls >= 0, VLG_ERR_BAD_ARG,
            "vlg_rope_table: bad argument");
  std::vector<float> t;
  rope_table_host(grid, vae_t, head_dim, base, cls, t);
  memcpy(host_out, t.data(), t.size() * sizeof(float));
  return VLG_OK;
}

extern "C" int vlg_gpt_create(const vlg_gpt_config* cfg, vlg_gpt_t** out) {
  VLG_CHECK(cfg && out, VLG_ERR_BAD_ARG, "vlg_gpt_create: null argument");
  VLG_CHECK(cfg->dim > 0 && cfg->n_head > 0 && cfg->n_layer > 0 && cfg->dim % cfg->n_head == 0, VLG_ERR_BAD_SHAPE,
            "vlg_gpt_create: dim %d not divisible by n_head %d", cfg->dim, cfg->n_head);
  VLG_CHECK(cfg->dtype == VLG_F32 || cfg->dtype == VLG_BF16, VLG_ERR_UNSUPPORTED, "vlg_gpt_create: dtype %d", cfg->dtype);
  VLG_CHECK(cfg->model_type >= VLG_C2I && cfg->model_type <= VLG_T2V, VLG_ERR_UNSUPPORTED, "please check model type");
  const int hd = cfg->dim / cfg->n_head;
  VLG_CHECK(hd == 32 || hd == 64 || hd == 96 || hd == 100 || hd == 128, VLG_ERR_UNSUPPORTED, "head_dim %d unsupported", hd);
  const int grid = (int)lround(sqrt((double)cfg->block_size));
  VLG_CHECK(grid * grid == cfg->block_size, VLG_ERR_BAD_SHAPE, "block_size %d is not a square", cfg->block_size);
  std::unique_ptr<vlg_gpt> h(new vlg_gpt());
  h->cfg = *cfg;
  h->D = cfg->dim;
  h->H = cfg->n_head;
  h->hd = hd;
  h->L = cfg->n_layer;
  h->F = ffn_hidden(cfg->dim, cfg->multiple_of > 0 ? cfg->multiple_of : 256);
  h->V = cfg->vocab_size;
  h->Tc = cfg->cls_token_num;
  h->C = cfg->vae_embed_dim;
  h->cd = cfg->caption_dim;
  h->grid = grid;
  h->vae_t = cfg->model_type == VLG_T2V ? (cfg->num_frames - 1) / cfg->t_downsample_size + 1 : 1;
  h->dtype = cfg->dtype;
  h->esz = dtype_size(cfg->dtype);
  if (cfg->model_type == VLG_C2I) VLG_CHECK(h->Tc == 1, VLG_ERR_BAD_SHAPE, "c2i needs cls_token_num == 1");
  if (cfg->model_type == VLG_T2V)
    VLG_CHECK(cfg->head == VLG_HEAD_ADAPTER2 || cfg->head == VLG_HEAD_HIDDEN, VLG_ERR_UNSUPPORTED, "t2v needs adapter2/hidden head");
  else
    VLG_CHECK(cfg->head == VLG_HEAD_LOGITS, VLG_ERR_UNSUPPORTED, "c2i/t2i need the logits head");

  const int64_t D = h->D, F = h->F, V = h->V;
  auto add = [&](const std::string& name, std::vector<int64_t> shape, const std::string& target = "", int64_t row_off = 0,
                 std::vector<int64_t> tshape = {}) {
    vlg_gpt::Spec s;
    s.shape = shape;
    s.target = target.empty() ? name : target;
    s.row_off = row_off;
    h->specs[name] = s;
    Tensor& t = h->w[s.target];
    if (t.shape.empty()) t.shape = tshape.empty() ? shape : tshape;
  };
  if (cfg->model_type == VLG_C2I) {
    add("cls_embedding.embedding_table.weight", {cfg->num_classes + 1, D});
  } else {
    add("cls_embedding.cap_proj.fc1.weight", {D, h->cd});
    add("cls_embedding.cap_proj.fc2.weight", {D, D});
    add("cls_embedding.uncond_embedding", {120, h->cd});
  }
  if (cfg->model_type == VLG_T2V) {
    add("vae_latent_adapter.fc1.weight", {D, h->C});
    add("vae_latent_adapter.fc2.weight", {D, D});
    if (cfg->head == VLG_HEAD_ADAPTER2) {
      add("vae_latent_adapter2.fc1.weight", {D, D});
      add("vae_latent_adapter2.fc2.weight", {h->C, D});
    }
    if (cfg->head == VLG_HEAD_HIDDEN) {   // diffloss.py:161-190
      VLG_CHECK(cfg->diffloss_w > 0 && cfg->diffloss_w % 4 == 0 && cfg->diffloss_d > 0 && cfg->num_sampling_steps > 0 &&
                    cfg->num_sampling_steps <= 1000, VLG_ERR_BAD_ARG, "bad DiffLoss configuration");
      h->dW = cfg->diffloss_w;
      h->dDepth = cfg->diffloss_d;
      const int64_t W = h->dW, dd = h->dDepth, Cc = h->C;
      const std::string p = "diffloss.net.";
      auto lin = [&](const std::string& n, int64_t o, int64_t i) {
        add(p + n + ".weight", {o, i});
        add(p + n + ".bias", {o});
      };
      lin("time_embed.mlp.0", W, 256);
      lin("time_embed.mlp.2", W, W);
      lin("cond_embed", W, D);
      lin("input_proj", W, Cc);
      lin("final_layer.linear", 2 * Cc, W);
      const int64_t MR = (3 * dd + 2) * W;
      for (int64_t b = 0; b < dd; ++b) {
        const std::string q = p + "res_blocks." + std::to_string(b) + ".";
        add(q + "in_ln.weight", {W});
        add(q + "in_ln.bias", {W});
        add(q + "mlp.0.weight", {W, W});
        add(q + "mlp.0.bias", {W});
        add(q + "mlp.2.weight", {W, W});
        add(q + "mlp.2.bias", {W});
        add(q + "adaLN_modulation.1.weight", {3 * W, W}, "diffloss.adaln_all.weight", b * 3 * W, {MR, W});
        add(q + "adaLN_modulation.1.bias", {3 * W}, "diffloss.adaln_all.bias", b * 3 * W, {MR});
      }
      add(p + "final_layer.adaLN_modulation.1.weight", {2 * W, W}, "diffloss.adaln_all.weight", dd * 3 * W, {MR, W});
      add(p + "final_layer.adaLN_modulation.1.bias", {2 * W}, "diffloss.adaln_all.bias", dd * 3 * W, {MR});
      build_diffusion_schedule(cfg->num_sampling_steps, h->dcoef, h->dsincos);
      h->dS = (int)h->dcoef.size();
    }
  } else {
    add("tok_embeddings.weight", {V, D});
    add("output.weight", {V, D});
  }
  for (int i = 0; i < h->L; ++i) {
    const std::string p = "layers." + std::to_string(i) + ".";
    add(p + "attention.wqkv.weight", {3 * D, D});
    add(p + "attention.wo.weight", {D, D});
    add(p + "feed_forward.w1.weight", {F, D}, p + "feed_forward.w13", 0, {2 * F, D});
    add(p + "feed_forward.w3.weight", {F, D}, p + "feed_forward.w13", F, {2 * F, D});
    add(p + "feed_forward.w2.weight", {D, F});
    add(p + "attention_norm.weight", {D});
    add(p + "ffn_norm.weight", {D});
  }
  add("norm.weight", {D});
  for (auto& kv : h->w) VLG_TRY(kv.second.buf.reserve((size_t)kv.second.numel() * h->esz));

  std::vector<float> tab;
  rope_table_host(grid, h->vae_t, hd, cfg->rope_base, h->Tc, tab);
  h->npos = h->Tc + h->vae_t * grid * grid;
  VLG_TRY(h->freqs.reserve(tab.size() * sizeof(float)));
  VLG_HIP(hipMemcpy(h->freqs.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
  VLG_HIP(hipStreamCreateWithFlags(&h->s_int, hipStreamNonBlocking));
  VLG_HIP(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
  VLG_HIP(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
  VLG_HIP(hipHostMalloc((void**)&h->fault_host, 64, hipHostMallocMapped));
  memset(h->fault_host, 0, 64);
  VLG_HIP(hipHostGetDevicePointer((void**)&h->fault_dev, h->fault_host, 0));
  *out = h.release();
  return VLG_OK;
}

extern "C" int vlg_gpt_destroy(vlg_gpt_t* h) {
  if (!h) return VLG_OK;
  (void)hipDeviceSynchronize();
  delete h;
  return VLG_OK;
}

extern "C" int vlg_gpt_load_tensor(vlg_gpt_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim,
                                   int32_t src_dtype, int32_t src_on_device, int32_t* consumed) {
  VLG_CHECK(h && name && data && shape, VLG_ERR_BAD_ARG, "vlg_gpt_load_tensor: null argument");
  if (consumed) *consumed = 0;
  auto it = h->specs.find(name);
  if (it == h->specs.end()) return VLG_OK;  // strict=False
  const auto& sp = it->second;
  bool ok = (int)sp.shape.size() == ndim;
  int64_t n = 1;
  for (int i = 0; ok && i < ndim; ++i) {
    ok = shape[i] == sp.shape[i];
    n *= shape[i];
  }
  VLG_CHECK(ok, VLG_ERR_BAD_SHAPE, "size mismatch for %s", name);
  VLG_CHECK(src_dtype == VLG_F32 || src_dtype == VLG_BF16, VLG_ERR_BAD_ARG, "bad src dtype");
  Tensor& t = h->w[sp.target];
  const int64_t row_elems = ndim > 1 ? n / shape[0] : 1;
  char* dst = (char*)t.buf.p + (size_t)sp.row_off * row_elems * h->esz;
  VLG_TRY(upload_convert(dst, h->dtype, data, src_dtype, src_on_device, n, h->s_int));
  t.loaded = true;
  t.fm_stale = true;
  if (consumed) *consumed = 1;
  return VLG_OK;
}

extern "C" int vlg_gpt_set_option(vlg_gpt_t* h, const char* key, int64_t value) {
  VLG_CHECK(h && key, VLG_ERR_BAD_ARG, "vlg_gpt_set_option: null");
  if (!strcmp(key, "graph")) {
    h->use_graph = value != 0;
    return VLG_OK;
  }
  if (!strcmp(key, "time_attn")) {
    h->time_attn = value != 0;
    return VLG_OK;
  }
  if (!strcmp(key, "fuse_gemm")) {
    h->fuse_gemm = value != 0;
    return VLG_OK;
  }
  if (!strcmp(key, "fuse_swiglu")) {
    h->fuse_swiglu = value != 0;
    return VLG_OK;
  }
  if (!strcmp(key, "kv_block")) {
    VLG_CHECK(value == 0 || (value >= 8 && value <= 1024 && (value & (value - 1)) == 0), VLG_ERR_BAD_ARG, "kv_block must be 0 or a power of two in 8..1024");
    h->kv_block = (int)value;
    return VLG_OK;
  }
  if (!strcmp(key, "kv_pool_blocks")) {
    VLG_CHECK(value >= 0 && value < (1 << 24), VLG_ERR_BAD_ARG, "kv_pool_blocks out of range");
    h->kv_pool_blocks = (int)value;
    return VLG_OK;
  }
  if (!strcmp(key, "debug_spin_max")) {
    VLG_CHECK(value >= 0 && value <= (1 << 24), VLG_ERR_BAD_ARG, "debug_spin_max out of range");
    h->spin_max = (int)value;
    return VLG_OK;
  }
  if (!strcmp(key, "pdecode")) {
    h->pdecode = value != 0;
    return VLG_OK;
  }
  if (!strcmp(key, "debug_pos_offset")) {
    VLG_CHECK(value >= 0 && value < (1 << 20), VLG_ERR_BAD_ARG, "debug_pos_offset out of range");
    h->pos_offset = (int)value;
    return VLG_OK;
  }
  if (!strcmp(key, "weights_fm")) {
    h->weights_fm = value != 0;
    return VLG_OK;
  }
  if (!strcmp(key, "act_fm")) {
    h->act_fm = value != 0;
    return VLG_OK;
  }
  if (!strcmp(key, "pd_rows")) {
    VLG_CHECK(value >= 0 && value <= 32, VLG_ERR_BAD_ARG, "pd_rows must be in 0..32");
    h->pd_rows = (int)value;
    return VLG_OK;
  }
  if (!strcmp(key, "dl_persist")) {   // 0 off, 1 on, 4 / 8: on with that many rows per workgroup group
    VLG_CHECK(value == 0 || value == 1 || value == 4 || value == 8, VLG_ERR_BAD_ARG, "dl_persist must be 0, 1, 4 or 8");
    h->dl_persist_on = value != 0;
    h->dl_rows = value > 1 ? (int)value : 0;
    return VLG_OK;
  }
  set_error("unknown option %s", key);
  return VLG_ERR_BAD_ARG;
}

extern "C" int vlg_gpt_set_teacher(vlg_gpt_t* h, const int32_t* d_ids, const float* d_latents) {
  VLG_CHECK(h, VLG_ERR_BAD_ARG, "vlg_gpt_set_teacher: null handle");
  VLG_CHECK(!(d_ids && d_latents), VLG_ERR_BAD_ARG, "vlg_gpt_set_teacher: ids or latents, not both");
  if (d_ids) VLG_CHECK(h->cfg.head == VLG_HEAD_LOGITS, VLG_ERR_BAD_ARG, "vlg_gpt_set_teacher: forced ids need a token head");
  if (d_latents) VLG_CHECK(h->cfg.head != VLG_HEAD_LOGITS, VLG_ERR_BAD_ARG, "vlg_gpt_set_teacher: forced latents need a latent head");
  h->teach_ids = d_ids;
  h->teach_lat = d_latents;
  return VLG_OK;
}

// a pending device-side fault (time-out inside a persistent kernel): report + clear
static int collect_fault(vlg_gpt* h) {
  const unsigned f = __atomic_load_n(h->fault_host, __ATOMIC_ACQUIRE);
  if (f == 0) return VLG_OK;
  __atomic_store_n(h->fault_host, 0u, __ATOMIC_RELEASE);
  const unsigned kind = f & 0xffff0000u;
  set_error("device fault 0x%08x: an in-launch wait of the %s ran out (index %u) - the grid was not fully resident (another kernel held compute "
            "units?) or debug_spin_max forced it; the results of that call are invalid",
            f, kind == kFaultDlPersist ? "persistent DiffLoss sampler" : (kind == kFaultDecode ? "persistent decode step" : "persistent kernel"), f & 0xffffu);
  return VLG_ERR_STATE;
}

extern "C" int vlg_gpt_graphs_built(vlg_gpt_t* h, int64_t* count) {
  VLG_CHECK(h && count, VLG_ERR_BAD_ARG, "vlg_gpt_graphs_built: null argument");
  *count = h->graphs_built;
  return VLG_OK;
}

extern "C" int vlg_gpt_counter(vlg_gpt_t* h, const char* key, int64_t* count) {
  VLG_CHECK(h && key && count, VLG_ERR_BAD_ARG, "vlg_gpt_counter: null argument");
  if (!strcmp(key, "pd_steps")) *count = h->pd_steps;
  else if (!strcmp(key, "chain_steps")) *count = h->chain_steps;
  else if (!strcmp(key, "graphs_built")) *count = h->graphs_built;
  else {
    set_error("vlg_gpt_counter: unknown key %s", key);
    return VLG_ERR_BAD_ARG;
  }
  return VLG_OK;
}

extern "C" int vlg_gpt_attn_event_overhead(vlg_gpt_t* h, double* ms_per_pair) {
  VLG_CHECK(h && ms_per_pair, VLG_ERR_BAD_ARG, "vlg_gpt_attn_event_overhead: null argument");
  *ms_per_pair = h->attn_pair_overhead_ms;
  return VLG_OK;
}

extern "C" int vlg_gpt_attn_timing(vlg_gpt_t* h, double* ms_sum, double* bytes_sum, int64_t* launches) {
  VLG_CHECK(h, VLG_ERR_BAD_ARG, "null handle");
  if (ms_sum) *ms_sum = h->attn_ms_sum;
  if (bytes_sum) *bytes_sum = h->attn_bytes_sum;
  if (launches) *launches = h->attn_launches;
  return VLG_OK;
}

extern "C" int vlg_gpt_last_algorithmic_bytes(vlg_gpt_t* h, double* wb, double* kb, double* ob) {
  VLG_CHECK(h, VLG_ERR_BAD_ARG, "null handle");
  if (wb) *wb = h->bytes_w;
  if (kb) *kb = h->bytes_kv;
  if (ob) *ob = h->bytes_other;
  return VLG_OK;
}

namespace {

// fragment-major copies of the four Linear weights of every layer (gpt_kernels.h: relayout_fragment_major), rebuilt after a (re)load.
// Twice the weight memory for these tensors - 1.5 GB more for GPT-XL, 6 GB for GPT-3B, on a 288 GB part.
template <typename T>
int ensure_fm(vlg_gpt* h) {
  bool any = false;
  for (int l = 0; l < h->L; ++l) {
    const std::string p = "layers." + std::to_string(l) + ".";
    for (const char* nm : {"attention.wqkv.weight", "attention.wo.weight", "feed_forward.w13", "feed_forward.w2.weight"}) {
      Tensor& t = h->w.at(p + nm);
      if (!t.fm_stale || !t.loaded || t.shape.size() != 2) continue;
      const int N = (int)t.shape[0], K = (int)t.shape[1];
      if (!fragment_major_ok(N, K, (int)sizeof(T))) continue;
      VLG_TRY(t.fm.reserve((size_t)N * K * sizeof(T)));
      VLG_TRY(relayout_fragment_major<T>(reinterpret_cast<const T*>(t.buf.p), reinterpret_cast<T*>(t.fm.p), N, K, h->s_int));
      t.fm_stale = false;
      any = true;
    }
  }
  // the per-step GEMMs of the fused decode path (token head: V x D, as large as a layer; latent adapters; DiffLoss condition embedding)
  for (const char* nm : {"output.weight", "vae_latent_adapter.fc2.weight", "vae_latent_adapter2.fc1.weight", "diffloss.net.cond_embed.weight"}) {
    auto it = h->w.find(nm);
    if (it == h->w.end()) continue;
    Tensor& t = it->second;
    if (!t.fm_stale || !t.loaded || t.shape.size() != 2) continue;
    const int N = (int)t.shape[0], K = (int)t.shape[1];
    if (!fragment_major_ok(N, K, (int)sizeof(T))) continue;
    VLG_TRY(t.fm.reserve((size_t)N * K * sizeof(T)));
    VLG_TRY(relayout_fragment_major<T>(reinterpret_cast<const T*>(t.buf.p), reinterpret_cast<T*>(t.fm.p), N, K, h->s_int));
    t.fm_stale = false;
    any = true;
  }
  if (any) VLG_HIP(hipStreamSynchronize(h->s_int));
  return VLG_OK;
}

// PdLayer[L] for the persistent decode kernel: weight buffers are allocated at create, so the pointers never change (built outside of
// any stream capture)
int ensure_pd_layers(vlg_gpt* h) {   // after ensure_fm: the table points at the fragment-major copies when every layer has them
  static const bool fm_off = getenv("VLG_GEMM_FM") != nullptr && atoi(getenv("VLG_GEMM_FM")) == 0;
  bool fm = !fm_off && h->weights_fm;
  for (int l = 0; l < h->L && fm; ++l) {
    const std::string p = "layers." + std::to_string(l) + ".";
    for (const char* nm : {"attention.wqkv.weight", "attention.wo.weight", "feed_forward.w13", "feed_forward.w2.weight"}) fm = fm && h->Wfm(p + nm) != nullptr;
  }
  if (h->pd_layers_dev.p != nullptr && h->pd_fm == fm) return VLG_OK;
  h->pd_fm = fm;
  std::vector<PdLayer> v(h->L);
  for (int l = 0; l < h->L; ++l) {
    const std::string p = "layers." + std::to_string(l) + ".";
    auto wp = [&](const char* nm) { return fm ? h->Wfm(p + nm) : h->W(p + nm); };
    v[l] = PdLayer{wp("attention.wqkv.weight"), wp("attention.wo.weight"), wp("feed_forward.w13"),
                   wp("feed_forward.w2.weight"), h->W(p + "attention_norm.weight"), h->W(p + "ffn_norm.weight")};
  }
  VLG_TRY(h->pd_layers_dev.reserve(v.size() * sizeof(PdLayer)));
  VLG_HIP(hipMemcpy(h->pd_layers_dev.p, v.data(), v.size() * sizeof(PdLayer), hipMemcpyHostToDevice));
  return VLG_OK;
}

// A Runner drives one set of buffers (Lane: activations, KV cache, step state, stream) through prefill / decode steps.  b0 / Btot
// place its rows inside a larger call (sessions prefill one slot at a time).  Splitting a batch into concurrent lanes on forked graph
// branches was measured slower in rounds 1 and 2 and is gone.
template <typename T>
struct Runner {
  vlg_gpt* h;
  Lane* ln;
  hipStream_t st;
  int B, Bp, N, S;     // lane batch, lane rows (2B with CFG), tokens to generate, cache length
  int b0, Btot;        // first sample of the lane within the call's batch, call batch
  const float* mask;   // device [B, Tc] (lane slice) or null
  int ev_slot = -1;    // >= 0: bracket layer 0's attention kernel with attn_ev[2*slot], [2*slot+1]
  const int32_t* row_pos = nullptr;    // sessions: per-row positions / token indices (device arrays); null = uniform StepState
  const int32_t* row_step = nullptr;
  int kv_row0 = 0, kv_rows = 0;        // this runner's rows are rows kv_row0.. of a cache holding kv_rows batch rows (0 = Bp): slot prefill
  const void* pending = nullptr;       // sessions of text-conditioned models: [rows][D] input rows of slots that start this step
  KvPages pages{};                     // sessions with a block-granular cache: block table of THIS runner's rows
  int pool_blocks = 0;                 //   and the number of blocks in the per-layer pool
  bool afm = false;                    // this decode step keeps x / attention output / SwiGLU output A-fragment-major (afm_ok)
  size_t kv_lstride() const {
    if (pages.table) return ((size_t)pool_blocks * h->H << pages.shift) * h->hd;
    return (size_t)(kv_rows ? kv_rows : Bp) * h->H * S * h->hd;
  }
  size_t kv_off() const { return pages.table ? 0 : (size_t)kv_row0 * h->H * S * h->hd; }
  StepState* state() { return ln->state.as<StepState>(); }
  template <typename U>
  const U* W(const std::string& n) {
    return reinterpret_cast<const U*>(h->W(n));
  }
  template <typename U>
  const U* Wfm(const std::string& n) {
    return reinterpret_cast<const U*>(h->Wfm(n));
  }

  int linear(const T* x, const std::string& wname, T* out, float* out_f32, int M, int Nn, int K, int act) {
    int sp = 1;
    VLG_TRY(gemm_slabs<T>(x, W<T>(wname), ln->ws.as<float>(), M, Nn, K, &sp, st));
    return reduce_store<T>(ln->ws.as<float>(), sp, out, out_f32, M, Nn, act, st);
  }

  // x [M, D] (rows m = b*Tq + t) -> xn = final-normed hidden
  int layers(int Tq, int max_pos) {
    const int M = Bp * Tq, D = h->D, H = h->H, hd = h->hd, F = h->F;
    T* x = ln->x.as<T>();
    T* xn = ln->xn.as<T>();
    float* ws = ln->ws.as<float>();
    const size_t lstride = kv_lstride();
    // (Round 4 measured the two wide matrices as one-pass 64-row GEMMs with RoPE-scatter / SwiGLU inside - tools/microbench/gemm_rows64_attempt.hip:
    // 65.0 us per layer for wqkv + w13 against 63.1 us for the slab GEMMs + their reduce launches here, config 5 2.17 vs 2.11 s - not adopted.)
    VLG_TRY(reduce_residual_rmsnorm<T>(nullptr, 0, x, W<T>("layers.0.attention_norm.weight"), xn, M, D, h->cfg.norm_eps, st));
    for (int l = 0; l < h->L; ++l) {
      const std::string p = "layers." + std::to_string(l) + ".";
      int sp = 1;
      T* kc = ln->kcache.as<T>() + lstride * l + kv_off();
      T* vc = ln->vcache.as<T>() + lstride * l + kv_off();
      VLG_TRY(gemm_slabs<T>(xn, W<T>(p + "attention.wqkv.weight"), ws, M, 3 * D, D, &sp, st, fm_on() ? Wfm<T>(p + "attention.wqkv.weight") : nullptr));
      VLG_TRY(qkv_rope_scatter<T>(ws, sp, ln->q.as<T>(), kc, vc, h->freqs.as<float>(), state(), M, Tq, H, hd, S, st, row_pos, pages));
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (l == 0 && ev_slot >= 0) {
        e0 = h->attn_ev[2 * ev_slot];
        e1 = h->attn_ev[2 * ev_slot + 1];
      }
      VLG_TRY(attn_rows<T>(ln->q.as<T>(), kc, vc, ln->ao.as<T>(), ln->attn_ws.as<float>(), state(), Bp, Tq, H, hd, S, max_pos, mask, B,
                           h->Tc, st, e0, e1, row_pos, pages));
      VLG_TRY(gemm_slabs<T>(ln->ao.as<T>(), W<T>(p + "attention.wo.weight"), ws, M, D, D, &sp, st, fm_on() ? Wfm<T>(p + "attention.wo.weight") : nullptr));
      VLG_TRY(reduce_residual_rmsnorm<T>(ws, sp, x, W<T>(p + "ffn_norm.weight"), xn, M, D, h->cfg.norm_eps, st));
      if (!h->fuse_swiglu || !gemm_swiglu<T>(xn, W<T>(p + "feed_forward.w13"), ln->g.as<T>(), M, F, D, st)) {
        VLG_TRY(gemm_slabs<T>(xn, W<T>(p + "feed_forward.w13"), ws, M, 2 * F, D, &sp, st, fm_on() ? Wfm<T>(p + "feed_forward.w13") : nullptr));
        VLG_TRY(reduce_silu_mul<T>(ws, sp, ln->g.as<T>(), M, F, st));
      }
      VLG_TRY(gemm_slabs<T>(ln->g.as<T>(), W<T>(p + "feed_forward.w2.weight"), ws, M, D, F, &sp, st, fm_on() ? Wfm<T>(p + "feed_forward.w2.weight") : nullptr));
      const std::string nxt = (l + 1 < h->L) ? "layers." + std::to_string(l + 1) + ".attention_norm.weight" : std::string("norm.weight");
      VLG_TRY(reduce_residual_rmsnorm<T>(ws, sp, x, W<T>(nxt), xn, M, D, h->cfg.norm_eps, st));
    }
    return VLG_OK;
  }

  // ---- fused decode path (Tq == 1): 6 launches per layer, no slabs ------------------------------------------------------
  bool fused_decode_ok() {
    const int D = h->D, F = h->F;
    if (!h->fuse_gemm || h->hd % 2 != 0) return false;
    // pro = false: only the tiling has to fit.  Where the in-kernel RMSNorm prologue does not cover the width (K * elem > 4 KB or more K
    // blocks than one pass holds: GPT-3B, D 3200) norm_gemm() runs the norm as its own launch in front of the same kernel.
    static const int nopro_rows = getenv("VLG_FUSED_NOPRO_ROWS") ? atoi(getenv("VLG_FUSED_NOPRO_ROWS")) : 32;
    const bool pro = Bp > nopro_rows;   // beyond that many rows the 64 x 64 slab GEMMs win (measured: GPT-3B, 64 rows, 2.58 vs 3.27 s)
    bool ok = gemm_fused_ok<T>(Bp, 3 * D, D, pro, EPI_QKV) && gemm_fused_ok<T>(Bp, D, D, false, EPI_RESID) &&
              gemm_fused_ok<T>(Bp, F, D, pro, EPI_SWIGLU) && gemm_fused_ok<T>(Bp, D, F, false, EPI_RESID);
    if (h->cfg.head == VLG_HEAD_LOGITS) ok = ok && gemm_fused_ok<T>(Bp, h->V, D, pro, EPI_STORE);
    if (h->cfg.head == VLG_HEAD_ADAPTER2) ok = ok && gemm_fused_ok<T>(Bp, D, D, pro, EPI_STORE);
    if (h->cfg.head == VLG_HEAD_HIDDEN) ok = ok && gemm_fused_ok<T>(Bp, h->dW, D, pro, EPI_STORE);
    return ok;
  }
  bool fm_on() const {   // A/B knobs: option "weights_fm" = 0 / VLG_GEMM_FM=0 stream the row-major weights everywhere
    static const bool off = getenv("VLG_GEMM_FM") != nullptr && atoi(getenv("VLG_GEMM_FM")) == 0;
    return !off && h->weights_fm;
  }
  // y = epilogue(RMSNorm(x; norm_w) @ w^T): one launch with the norm as the GEMM's prologue, or norm + GEMM where the prologue does not
  // cover K (same rounding points: the explicit kernel is the slab path's, xn rounded to T either way)
  int norm_gemm(T* x, const T* norm_w, const T* w, int Nn, int K, int epi, FusedGemm& fa, const T* wfm = nullptr) {
    fa.eps = h->cfg.norm_eps;
    fa.wfm = fm_on() ? wfm : nullptr;
    if (gemm_fused_ok<T>(Bp, Nn, K, true, epi)) {
      fa.norm_w = norm_w;
      return gemm_fused<T>(x, w, Bp, Nn, K, true, epi, fa, st);
    }
    VLG_TRY(reduce_residual_rmsnorm<T>(nullptr, 0, x, norm_w, ln->xn.as<T>(), Bp, K, h->cfg.norm_eps, st));
    fa.norm_w = nullptr;
    return gemm_fused<T>(ln->xn.as<T>(), w, Bp, Nn, K, false, epi, fa, st);
  }

  // x [Bp, D] = residual stream (token / latent embeddings); on return x holds the last layer's output, NOT normed
  int layers_fused() {
    const int M = Bp, D = h->D, H = h->H, hd = h->hd, F = h->F;
    T* x = ln->x.as<T>();
    const size_t lstride = kv_lstride();
    for (int l = 0; l < h->L; ++l) {
      const std::string p = "layers." + std::to_string(l) + ".";
      T* kc = ln->kcache.as<T>() + lstride * l + kv_off();
      T* vc = ln->vcache.as<T>() + lstride * l + kv_off();
      FusedGemm fa;
      fa.qbuf = ln->q.as<T>();
      fa.kc = kc;
      fa.vc = vc;
      fa.freqs = h->freqs.as<float>();
      fa.state = state();
      fa.row_pos = row_pos;
      fa.pages = pages;
      fa.Tq = 1;
      fa.H = H;
      fa.hd = hd;
      fa.S = S;
      fa.a_fm = afm;
      VLG_TRY(norm_gemm(x, W<T>(p + "attention_norm.weight"), W<T>(p + "attention.wqkv.weight"), 3 * D, D, EPI_QKV, fa, Wfm<T>(p + "attention.wqkv.weight")));
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (l == 0 && ev_slot >= 0) {
        e0 = h->attn_ev[2 * ev_slot];
        e1 = h->attn_ev[2 * ev_slot + 1];
      }
      VLG_TRY(attn_rows<T>(ln->q.as<T>(), kc, vc, ln->ao.as<T>(), ln->attn_ws.as<float>(), state(), Bp, 1, H, hd, S, S - 1, mask, B, h->Tc,
                           st, e0, e1, row_pos, pages, afm ? D * (int)sizeof(T) / 64 : 0));
      FusedGemm fr;
      fr.h = x;
      fr.a_fm = fr.o_fm = afm;   // A = ao (then g), result = the residual stream x
      fr.wfm = fm_on() ? Wfm<T>(p + "attention.wo.weight") : nullptr;
      VLG_TRY(gemm_fused<T>(ln->ao.as<T>(), W<T>(p + "attention.wo.weight"), M, D, D, false, EPI_RESID, fr, st));
      FusedGemm fs;
      fs.out = ln->g.as<T>();
      fs.a_fm = fs.o_fm = afm;   // A = x, result = g
      VLG_TRY(norm_gemm(x, W<T>(p + "ffn_norm.weight"), W<T>(p + "feed_forward.w13"), F, D, EPI_SWIGLU, fs, Wfm<T>(p + "feed_forward.w13")));
      fr.wfm = fm_on() ? Wfm<T>(p + "feed_forward.w2.weight") : nullptr;
      VLG_TRY(gemm_fused<T>(ln->g.as<T>(), W<T>(p + "feed_forward.w2.weight"), M, D, F, false, EPI_RESID, fr, st));
    }
    return VLG_OK;
  }

  // ---- persistent decode step: all layers in one launch (pdecode.hip) -------------------------------------------------------
  bool pd_use() {
    static const int off = getenv("VLG_PDECODE") ? (atoi(getenv("VLG_PDECODE")) == 0) : 0;
    // Measured against the launch chain: faster up to 8 rows; up to 16 rows while the (row, head) attention items fit
    // one round of the grid (GPT-B / GPT-L under guidance: 8 classes = 16 rows x 16 heads = 256 items); slower beyond.  Option "pd_rows" /
    // VLG_PD_ROWS replace the rule by a plain row cap.
    static const int rows_env = getenv("VLG_PD_ROWS") ? atoi(getenv("VLG_PD_ROWS")) : 0;
    const int rows_cap = h->pd_rows > 0 ? h->pd_rows : rows_env;
    if (off || !h->pdecode || row_pos != nullptr || pages.table != nullptr || kv_rows != 0 || ev_slot >= 0) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (rows_cap > 0 ? Bp > rows_cap : !(Bp <= 8 || (Bp <= 16 && Bp * h->H <= cus))) return false;
    return ln->pd_xbuf.p != nullptr && h->pd_layers_dev.p != nullptr && pd_ok<T>(Bp, h->D, h->H, h->hd, h->F, S, cus);
  }
  int layers_pd() {
    PdArgs a{};
    a.layers = h->pd_layers_dev.as<PdLayer>();
    a.x = ln->x.p;
    a.kc = ln->kcache.p;
    a.vc = ln->vcache.p;
    a.kv_lstride = kv_lstride();
    a.freqs = h->freqs.as<float>();
    a.state = state();
    a.mask = mask;
    a.Bmask = B;
    a.Tc = h->Tc;
    a.xbuf = ln->pd_xbuf.p;
    a.fault = h->fault_dev;
    a.spin_max = h->spin_max;
    a.fm = h->pd_fm ? 1 : 0;
    a.L = h->L; a.M = Bp; a.D = h->D; a.H = h->H; a.hd = h->hd; a.F = h->F; a.S = S;
    a.eps = h->cfg.norm_eps;
    return pd_layers<T>(a, st);
  }

  // head on the un-normed residual stream x [Bp, D]: the final RMSNorm (gpt.py:370) is the head GEMM's prologue
  int head_fused(const vlg_sampling_params& sp, const float* noise, int32_t* out_ids, float* out_lat, float* trace) {
    const int D = h->D;
    FusedGemm fa;
    fa.a_fm = afm;   // the residual stream; the head's own results stay row-major
    const T* final_norm = W<T>("norm.weight");
    if (h->cfg.head == VLG_HEAD_LOGITS) {
      fa.out_f32 = ln->logits.as<float>();
      VLG_TRY(norm_gemm(ln->x.as<T>(), final_norm, W<T>("output.weight"), h->V, D, EPI_STORE, fa, Wfm<T>("output.weight")));
      return sample_rows(ln->logits.as<float>(), B, h->V, Bp > B, sp, noise, state(), 0, N, out_ids ? out_ids + (size_t)b0 * N : nullptr,
                         ln->cur_tok.as<int32_t>(), trace, nullptr, st, b0, Btot, row_step);
    }
    if (h->cfg.head == VLG_HEAD_ADAPTER2) {
      fa.out = ln->t1.as<T>();
      fa.act = ACT_GELU_TANH;
      VLG_TRY(norm_gemm(ln->x.as<T>(), final_norm, W<T>("vae_latent_adapter2.fc1.weight"), D, D, EPI_STORE, fa, Wfm<T>("vae_latent_adapter2.fc1.weight")));
      if (h->C <= 16)   // fc2 (N = C) + CFG combine + stores in one launch (3 before)
        return latent_out_fc2<T>(ln->t1.as<T>(), W<T>("vae_latent_adapter2.fc2.weight"), ln->cur_lat.as<float>(),
                                 out_lat + (size_t)b0 * N * h->C, trace, state(), B, Bp, h->C, D, N, sp.cfg_scale, sp.cfg_interval, st, b0, Btot, row_step);
      VLG_TRY(linear(ln->t1.as<T>(), "vae_latent_adapter2.fc2.weight", ln->y.as<T>(), nullptr, Bp, h->C, D, ACT_NONE));
      return latent_head_finish<T>(ln->y.as<T>(), ln->cur_lat.as<float>(), out_lat + (size_t)b0 * N * h->C, trace, state(), B, Bp, h->C, N,
                                   sp.cfg_scale, sp.cfg_interval, st, b0, Btot, row_step);
    }
    // hidden (DiffLoss): cond_embed(norm(x)) is the first op of the head (diffloss.py:227)
    fa.out = ln->d_cemb.as<T>();
    fa.bias = W<T>("diffloss.net.cond_embed.bias");
    VLG_TRY(norm_gemm(ln->x.as<T>(), final_norm, W<T>("diffloss.net.cond_embed.weight"), h->dW, D, EPI_STORE, fa, Wfm<T>("diffloss.net.cond_embed.weight")));
    return diffloss_head(nullptr, sp, noise, out_lat, trace);
  }

  // hl [Bp, D] (normed hidden of the last position) -> sampled token / latent for state->step
  int head(const T* hl, const vlg_sampling_params& sp, const float* noise, int32_t* out_ids, float* out_lat, float* trace) {
    const int D = h->D;
    if (h->cfg.head == VLG_HEAD_LOGITS) {
      VLG_TRY(linear(hl, "output.weight", nullptr, ln->logits.as<float>(), Bp, h->V, D, ACT_NONE));
      return sample_rows(ln->logits.as<float>(), B, h->V, Bp > B, sp, noise, state(), 0, N, out_ids ? out_ids + (size_t)b0 * N : nullptr,
                         ln->cur_tok.as<int32_t>(), trace, nullptr, st, b0, Btot, row_step);
    }
    if (h->cfg.head == VLG_HEAD_ADAPTER2) {
      VLG_TRY(linear(hl, "vae_latent_adapter2.fc1.weight", ln->t1.as<T>(), nullptr, Bp, D, D, ACT_GELU_TANH));
      VLG_TRY(linear(ln->t1.as<T>(), "vae_latent_adapter2.fc2.weight", ln->y.as<T>(), nullptr, Bp, h->C, D, ACT_NONE));
      return latent_head_finish<T>(ln->y.as<T>(), ln->cur_lat.as<float>(), out_lat + (size_t)b0 * N * h->C, trace, state(), B, Bp, h->C, N,
                                   sp.cfg_scale, sp.cfg_interval, st, b0, Btot, row_step);
    }
    return diffloss_head(hl, sp, noise, out_lat, trace);
  }

  int linear_b(const T* x, const std::string& wname, T* out, int M, int Nn, int K, int act) {
    int sps = 1;
    VLG_TRY(gemm_slabs<T>(x, W<T>(wname + ".weight"), ln->ws.as<float>(), M, Nn, K, &sps, st));
    return reduce_store<T>(ln->ws.as<float>(), sps, out, nullptr, M, Nn, act, st, W<T>(wname + ".bias"));
  }

  // DiffLoss.sample (diffloss.py:35-52): x_T ~ N(0,1); S reverse steps of {SimpleMLPAdaLN(x, t, c = z) -> p_sample}.
  // z = hl [B, D] (normed hidden of the last position).  All S steps are enqueued (and graph-captured) back to back.
  int diffloss_head(const T* z, const vlg_sampling_params& sp, const float* noise, float* out_lat, float* trace) {
    if (diffloss_fused_ok()) return diffloss_head_fused(z, sp, noise, out_lat, trace);
    VLG_CHECK(row_step == nullptr, VLG_ERR_UNSUPPORTED, "sessions with the DiffLoss head need the persistent sampler (fused GEMM path)");
    VLG_CHECK(h->cfg_iter == 1.0f, VLG_ERR_UNSUPPORTED, "cfg_iter != 1: the unfused DiffLoss sampler (option fuse_gemm = 0 / widths the fused GEMM does not tile) has no guidance");
    const int Wd = h->dW, C = h->C, dd = h->dDepth, S = h->dS, MR = (3 * dd + 2) * Wd;
    const std::string p = "diffloss.net.";
    T* cemb = ln->d_cemb.as<T>();
    T* ys = ln->d_ys.as<T>();
    T* mod = ln->d_mod.as<T>();
    T* hc = ln->d_h.as<T>();
    T* g = ln->d_g.as<T>();
    T* g1 = ln->d_g1.as<T>();
    T* dout = ln->d_out.as<T>();
    T* x = ln->d_x.as<T>();
    if (z) VLG_TRY(linear_b(z, p + "cond_embed", cemb, B, Wd, h->D, ACT_NONE));   // else: already produced by head_fused
    VLG_TRY(dl_init_x<T>(x, noise, state(), S, B, C, b0, Btot, sp.seed, st));
    for (int k = 0; k < S; ++k) {
      const int i = S - 1 - k;
      VLG_TRY(linear_b(x, p + "input_proj", hc, B, Wd, C, ACT_NONE));
      VLG_TRY(dl_make_y<T>(h->dtemb.as<T>() + (size_t)i * Wd, cemb, ys, B, Wd, st));
      VLG_TRY(linear_b(ys, "diffloss.adaln_all", mod, B, MR, Wd, ACT_NONE));
      for (int blk = 0; blk < dd; ++blk) {
        const std::string q = p + "res_blocks." + std::to_string(blk) + ".";
        const T* m0 = mod + (size_t)blk * 3 * Wd;   // [shift | scale | gate] (diffloss.py:125)
        VLG_TRY(dl_ln_modulate<T>(hc, W<T>(q + "in_ln.weight"), W<T>(q + "in_ln.bias"), m0, m0 + Wd, MR, g, B, Wd, st));
        VLG_TRY(linear_b(g, q + "mlp.0", g1, B, Wd, Wd, ACT_SILU));
        VLG_TRY(linear_b(g1, q + "mlp.2", g, B, Wd, Wd, ACT_NONE));
        VLG_TRY(dl_gated_residual<T>(hc, m0 + 2 * Wd, MR, g, B, Wd, st));
      }
      const T* mf = mod + (size_t)dd * 3 * Wd;      // [shift | scale] (diffloss.py:145)
      VLG_TRY(dl_ln_modulate<T>(hc, nullptr, nullptr, mf, mf + Wd, MR, g, B, Wd, st));
      VLG_TRY(linear_b(g, p + "final_layer.linear", dout, B, 2 * C, Wd, ACT_NONE));
      VLG_TRY(dl_ddpm_step<T>(x, dout, noise, state(), h->dcoef[i], k, S, B, C, b0, Btot, sp.temperature, sp.seed, st));
    }
    return dl_finish<T>(x, ln->cur_lat.as<float>(), out_lat + (size_t)b0 * N * C, trace, state(), B, C, N, b0, Btot, st);
  }

  // The same sampler with 8-12 instead of 27 launches per reverse step (800-1200 vs 2700 per token; the head is launch-latency bound):
  //  * the adaLN modulation vectors depend on (t, z) only, not on x: one [S*B, W] x [W, (3d+2)W] GEMM per token up front;
  //  * p_sample + the next evaluation's input projection (K = C) in one elementwise launch;
  //  * mlp.0 (+bias, SiLU) and final_layer.linear as single fused-GEMM launches; mlp.2 with bias, gate and residual in its
  //    epilogue (EPI_GATED); LayerNorm + modulate as the prologue of mlp.0 / final_layer.linear (gemm_ln_kernel) when W fits.
  bool diffloss_fused_ok() {
    const int Wd = h->dW, C = h->C;
    return h->fuse_gemm && C <= 16 && (2 * C) % 16 == 0 && gemm_fused_ok<T>(B, Wd, Wd, false, EPI_STORE) &&
           gemm_fused_ok<T>(B, Wd, Wd, false, EPI_GATED) && gemm_fused_ok<T>(B, 2 * C, Wd, false, EPI_STORE);
  }
  int diffloss_head_fused(const T* z, const vlg_sampling_params& sp, const float* noise, float* out_lat, float* trace) {
    const int Wd = h->dW, C = h->C, dd = h->dDepth, S = h->dS, MR = (3 * dd + 2) * Wd;
    const std::string p = "diffloss.net.";
    T* cemb = ln->d_cemb.as<T>();
    T* ys = ln->d_ys.as<T>();
    T* mod_all = ln->d_mod.as<T>();
    T* hc = ln->d_h.as<T>();
    T* g = ln->d_g.as<T>();
    T* g1 = ln->d_g1.as<T>();
    T* dout = ln->d_out.as<T>();
    T* xa = ln->d_x.as<T>();
    T* xb = ln->d_x2.as<T>();
    if (z) VLG_TRY(linear_b(z, p + "cond_embed", cemb, B, Wd, h->D, ACT_NONE));
    VLG_TRY(dl_make_y_all<T>(h->dtemb.as<T>(), cemb, ys, S, B, Wd, st));
    {   // [S*B, W] x [W, MR]: a real GEMM (74 GFLOP at S 100, B 32, W 1024) - the 128 x 128 MFMA tile kernel of the decoders, as a 1x1 conv
      ConvDesc cd;
      cd.B = 1; cd.Ti = cd.To = 1; cd.Hi = cd.Ho = 1; cd.Wi = cd.Wo = S * B; cd.Cin = Wd; cd.Cout = MR;
      cd.kt = cd.kh = cd.kw = 1; cd.up = 0;
      VLG_TRY(conv_forward<T>(cd, ys, W<T>("diffloss.adaln_all.weight"), h->dadaln_bias.as<float>(), nullptr, mod_all, nullptr, st));
    }
    const T* wip = W<T>(p + "input_proj.weight");
    const T* bip = W<T>(p + "input_proj.bias");
    const bool dcfg = h->cfg_iter != 1.0f;   // forward_with_cfg: pairs (b, b + B/2) - in one 4-row group of the persistent kernel, or the chain below
    const int n_half = dcfg ? B / 2 : 0;
    if (dcfg && B % 2 != 0) {
      set_error("cfg_iter != 1 pairs row b with row b + B/2 (diffloss.py:38-39): it needs an even batch, got %d rows", B);
      return VLG_ERR_BAD_SHAPE;
    }
    if (h->dl_persist_on && dl_persist_ok<T>(B, Wd, C, dd, h->dl_rows)) {
      // all S reverse steps in one persistent launch (2 depth all-gathers per step between the workgroups of a 4-row group)
      DlPersist dp{};
      for (int blk = 0; blk < dd; ++blk) {
        const std::string q = p + "res_blocks." + std::to_string(blk) + ".";
        dp.ln_w[blk] = W<T>(q + "in_ln.weight");
        dp.ln_b[blk] = W<T>(q + "in_ln.bias");
        dp.w0[blk] = W<T>(q + "mlp.0.weight");
        dp.b0[blk] = W<T>(q + "mlp.0.bias");
        dp.w2[blk] = W<T>(q + "mlp.2.weight");
        dp.b2[blk] = W<T>(q + "mlp.2.bias");
      }
      dp.wf = W<T>(p + "final_layer.linear.weight");
      dp.bf = W<T>(p + "final_layer.linear.bias");
      dp.wip = wip;
      dp.bip = bip;
      dp.mod_all = mod_all;
      dp.coef = h->dcoef_dev.as<DdpmCoef>();
      dp.noise = noise;
      dp.state = state();
      dp.xbuf = ln->dp_xbuf.p;
      dp.cur = ln->cur_lat.as<float>();
      dp.out_lat = out_lat + (size_t)b0 * N * C;
      dp.trace = trace;
      dp.depth = dd; dp.W = Wd; dp.C = C; dp.S = S; dp.B = B; dp.MR = MR; dp.N = N; dp.b_off = b0; dp.B_total = Btot;
      dp.temperature = sp.temperature;
      dp.seed = sp.seed;
      dp.fault = h->fault_dev;
      dp.spin_max = h->spin_max > 0 ? h->spin_max : (1 << 20);
      dp.n_half = n_half;
      dp.cfg = h->cfg_iter;
      dp.rows = h->dl_rows;
      dp.row_step = row_step;
      return dl_persist<T>(dp, st);
    }
    VLG_CHECK(row_step == nullptr, VLG_ERR_UNSUPPORTED, "sessions with the DiffLoss head need the persistent sampler (option dl_persist, a shape it covers)");
    // LayerNorm + modulate inside the GEMM that consumes it (8 launches per reverse step) where the width fits its prologue
    const bool ln_in_gemm = gemm_ln_fused_ok<T>(B, Wd, Wd) && gemm_ln_fused_ok<T>(B, 2 * C, Wd) && !getenv("VLG_DIFFLOSS_NO_LN_FUSE");
    DdpmCoef none{};
    VLG_TRY(dl_step_proj<T>(xb, xa, nullptr, noise, state(), none, -1, S, B, C, b0, Btot, sp.temperature, sp.seed, wip, bip, hc, Wd, st, n_half,
                            h->cfg_iter));
    for (int k = 0; k < S; ++k) {
      const int i = S - 1 - k;
      const T* mod = mod_all + (size_t)i * B * MR;
      for (int blk = 0; blk < dd; ++blk) {
        const std::string q = p + "res_blocks." + std::to_string(blk) + ".";
        const T* m0 = mod + (size_t)blk * 3 * Wd;   // [shift | scale | gate] (diffloss.py:125)
        if (ln_in_gemm) {
          LnGemm f0;
          f0.ln_w = W<T>(q + "in_ln.weight");
          f0.ln_b = W<T>(q + "in_ln.bias");
          f0.shift = m0;
          f0.scale = m0 + Wd;
          f0.mod_stride = MR;
          f0.out = g1;
          f0.bias = W<T>(q + "mlp.0.bias");
          f0.act = ACT_SILU;
          VLG_TRY(gemm_ln_fused<T>(hc, W<T>(q + "mlp.0.weight"), B, Wd, Wd, f0, st));
        } else {
          VLG_TRY(dl_ln_modulate<T>(hc, W<T>(q + "in_ln.weight"), W<T>(q + "in_ln.bias"), m0, m0 + Wd, MR, g, B, Wd, st));
          FusedGemm f0;
          f0.out = g1;
          f0.bias = W<T>(q + "mlp.0.bias");
          f0.act = ACT_SILU;
          VLG_TRY(gemm_fused<T>(g, W<T>(q + "mlp.0.weight"), B, Wd, Wd, false, EPI_STORE, f0, st));
        }
        FusedGemm f2;
        f2.h = hc;
        f2.bias = W<T>(q + "mlp.2.bias");
        f2.gate = m0 + 2 * Wd;
        f2.gate_stride = MR;
        VLG_TRY(gemm_fused<T>(g1, W<T>(q + "mlp.2.weight"), B, Wd, Wd, false, EPI_GATED, f2, st));
      }
      const T* mf = mod + (size_t)dd * 3 * Wd;      // [shift | scale] (diffloss.py:145)
      if (ln_in_gemm) {
        LnGemm ff;
        ff.shift = mf;
        ff.scale = mf + Wd;
        ff.mod_stride = MR;
        ff.out = dout;
        ff.bias = W<T>(p + "final_layer.linear.bias");
        VLG_TRY(gemm_ln_fused<T>(hc, W<T>(p + "final_layer.linear.weight"), B, 2 * C, Wd, ff, st));
      } else {
        VLG_TRY(dl_ln_modulate<T>(hc, nullptr, nullptr, mf, mf + Wd, MR, g, B, Wd, st));
        FusedGemm ff;
        ff.out = dout;
        ff.bias = W<T>(p + "final_layer.linear.bias");
        VLG_TRY(gemm_fused<T>(g, W<T>(p + "final_layer.linear.weight"), B, 2 * C, Wd, false, EPI_STORE, ff, st));
      }
      // x_{k+1} from x_k (ping-pong) and, unless this was the last step, the next evaluation's input projection
      T* xin = (k & 1) ? xb : xa;
      T* xout = (k & 1) ? xa : xb;
      VLG_TRY(dl_step_proj<T>(xin, xout, dout, noise, state(), h->dcoef[i], k, S, B, C, b0, Btot, sp.temperature, sp.seed, wip, bip,
                              k + 1 < S ? hc : nullptr, Wd, st, n_half, h->cfg_iter));
    }
    const T* xfin = (S & 1) ? xb : xa;
    return dl_finish<T>(xfin, ln->cur_lat.as<float>(), out_lat + (size_t)b0 * N * C, trace, state(), B, C, N, b0, Btot, st);
  }

  // time_embed(t) for every respaced step: [S,256] sincos -> Linear -> SiLU -> Linear (diffloss.py:93-96), once per handle
  int build_time_table() {
    const int Wd = h->dW, S = h->dS;
    DevBuf sc, t1;
    VLG_TRY(sc.reserve((size_t)S * 256 * sizeof(T)));
    VLG_TRY(t1.reserve((size_t)S * Wd * sizeof(T)));
    VLG_TRY(upload_convert(sc.p, DT<T>::code, h->dsincos.data(), VLG_F32, 0, (int64_t)S * 256, st));
    VLG_TRY(h->dtemb.reserve((size_t)S * Wd * sizeof(T)));
    VLG_TRY(linear_b(sc.as<T>(), "diffloss.net.time_embed.mlp.0", t1.as<T>(), S, Wd, 256, ACT_SILU));
    VLG_TRY(linear_b(t1.as<T>(), "diffloss.net.time_embed.mlp.2", h->dtemb.as<T>(), S, Wd, Wd, ACT_NONE));
    VLG_HIP(hipStreamSynchronize(st));
    const int MR = (3 * h->dDepth + 2) * Wd;
    VLG_TRY(h->dadaln_bias.reserve((size_t)MR * sizeof(float)));
    VLG_TRY(upload_convert(h->dadaln_bias.p, VLG_F32, W<T>("diffloss.adaln_all.bias"), DT<T>::code, 1, MR, st));
    VLG_TRY(h->dcoef_dev.reserve(h->dcoef.size() * sizeof(DdpmCoef)));
    VLG_HIP(hipMemcpy(h->dcoef_dev.p, h->dcoef.data(), h->dcoef.size() * sizeof(DdpmCoef), hipMemcpyHostToDevice));
    h->dtemb_ready = true;
    return VLG_OK;
  }

  // The activation matrices of a fused decode step (x, attention output, SwiGLU output) A-fragment-major (gpt_kernels.h afm_index)?
  // Where every consumer is a fused GEMM of this chain: the launch chain (not the persistent step, which hands activations over in its own
  // format) with the norms as GEMM prologues (the stand-alone RMSNorm reads rows); uniform positions and sessions alike.
  bool afm_ok() {
    static const bool off = getenv("VLG_ACT_FM") != nullptr && atoi(getenv("VLG_ACT_FM")) == 0;   // A/B knob
    const int D = h->D, F = h->F;
    if (off || !h->act_fm || !fused_decode_ok() || pd_use()) return false;
    if ((D * (int)sizeof(T)) % 64 != 0 || (F * (int)sizeof(T)) % 64 != 0) return false;
    bool ok = gemm_fused_ok<T>(Bp, 3 * D, D, true, EPI_QKV) && gemm_fused_ok<T>(Bp, F, D, true, EPI_SWIGLU);
    if (h->cfg.head == VLG_HEAD_LOGITS) ok = ok && gemm_fused_ok<T>(Bp, h->V, D, true, EPI_STORE);
    if (h->cfg.head == VLG_HEAD_ADAPTER2) ok = ok && gemm_fused_ok<T>(Bp, D, D, true, EPI_STORE);
    if (h->cfg.head == VLG_HEAD_HIDDEN) ok = ok && gemm_fused_ok<T>(Bp, h->dW, D, true, EPI_STORE);
    return ok;
  }

  int decode_step(const vlg_sampling_params& sp, const float* noise, int32_t* out_ids, float* out_lat, float* trace) {
    const int D = h->D;
    afm = false;
    if (h->cfg.model_type == VLG_T2V && h->fuse_gemm && h->C <= 16 && gemm_fused_ok<T>(Bp, D, D, false, EPI_STORE)) {
      // latent -> adapter.fc1 -> GELU in one launch, fc2 as one fused-GEMM launch (5 launches before)
      afm = afm_ok();
      VLG_TRY(latent_in_fc1<T>(ln->cur_lat.as<float>(), W<T>("vae_latent_adapter.fc1.weight"), ln->t1.as<T>(), B, Bp, h->C, D, st));
      FusedGemm f2;
      f2.o_fm = afm;
      f2.out = ln->x.as<T>();
      f2.wfm = fm_on() ? Wfm<T>("vae_latent_adapter.fc2.weight") : nullptr;
      VLG_TRY(gemm_fused<T>(ln->t1.as<T>(), W<T>("vae_latent_adapter.fc2.weight"), Bp, D, D, false, EPI_STORE, f2, st));
    } else if (h->cfg.model_type == VLG_T2V) {
      VLG_TRY(latent_to_rows<T>(ln->cur_lat.as<float>(), ln->latT.as<T>(), B, Bp, h->C, st));
      VLG_TRY(linear(ln->latT.as<T>(), "vae_latent_adapter.fc1.weight", ln->t1.as<T>(), nullptr, Bp, D, h->C, ACT_GELU_TANH));
      VLG_TRY(linear(ln->t1.as<T>(), "vae_latent_adapter.fc2.weight", ln->x.as<T>(), nullptr, Bp, D, D, ACT_NONE));
    } else {
      afm = afm_ok();
      VLG_TRY(gather_rows_i32<T>(W<T>("tok_embeddings.weight"), ln->cur_tok.as<int32_t>(), ln->x.as<T>(), Bp, D, h->V, st,
                                 afm ? D * (int)sizeof(T) / 64 : 0));
    }
    if (fused_decode_ok()) {
      if (pd_use()) {
        VLG_TRY(layers_pd());
        h->pd_steps += 1;
      } else {
        VLG_TRY(layers_fused());
        h->chain_steps += 1;
      }
      VLG_TRY(head_fused(sp, noise, out_ids, out_lat, trace));
    } else {
      VLG_TRY(layers(1, S - 1));
      VLG_TRY(head(ln->xn.as<T>(), sp, noise, out_ids, out_lat, trace));
    }
    VLG_TRY(teacher());
    return advance_state(state(), st);
  }
  // teacher forcing (vlg_gpt_set_teacher): behind the head, in front of the state advance
  int teacher() {
    if (h->teach_ids == nullptr && h->teach_lat == nullptr) return VLG_OK;
    return force_next_input(state(), h->teach_ids, h->teach_lat, ln->cur_tok.as<int32_t>(), h->C > 0 ? ln->cur_lat.as<float>() : nullptr, B, Bp,
                            h->C > 0 ? h->C : 1, N, b0, st);
  }

  // one iteration of the request scheduler: every row at its own position (StepState::row_pos / row_step), inputs per row_cls
  int session_step(const vlg_sampling_params& sp, const int32_t* row_cls, int32_t* out_ids, float* out_lat = nullptr) {
    const int D = h->D;
    if (h->cfg.model_type == VLG_T2V) {
      // continuous-latent models: every row's input is the adapter's image of the latent it produced last (decode_step's two forms); rows that
      // start a request take the projected last condition token their prefill left in `pending`
      afm = false;
      if (h->fuse_gemm && h->C <= 16 && gemm_fused_ok<T>(Bp, D, D, false, EPI_STORE)) {
        afm = afm_ok();
        VLG_TRY(latent_in_fc1<T>(ln->cur_lat.as<float>(), W<T>("vae_latent_adapter.fc1.weight"), ln->t1.as<T>(), B, Bp, h->C, D, st));
        FusedGemm f2;
        f2.o_fm = afm;
        f2.out = ln->x.as<T>();
        f2.wfm = fm_on() ? Wfm<T>("vae_latent_adapter.fc2.weight") : nullptr;
        VLG_TRY(gemm_fused<T>(ln->t1.as<T>(), W<T>("vae_latent_adapter.fc2.weight"), Bp, D, D, false, EPI_STORE, f2, st));
      } else {
        VLG_TRY(latent_to_rows<T>(ln->cur_lat.as<float>(), ln->latT.as<T>(), B, Bp, h->C, st));
        VLG_TRY(linear(ln->latT.as<T>(), "vae_latent_adapter.fc1.weight", ln->t1.as<T>(), nullptr, Bp, D, h->C, ACT_GELU_TANH));
        VLG_TRY(linear(ln->t1.as<T>(), "vae_latent_adapter.fc2.weight", ln->x.as<T>(), nullptr, Bp, D, D, ACT_NONE));
      }
      VLG_TRY(override_session_rows<T>(row_cls, reinterpret_cast<const T*>(pending), ln->x.as<T>(), Bp, D, st, afm ? D * (int)sizeof(T) / 64 : 0));
      if (fused_decode_ok()) {
        VLG_TRY(layers_fused());
        return head_fused(sp, nullptr, nullptr, out_lat, nullptr);
      }
      VLG_TRY(layers(1, S - 1));
      return head(ln->xn.as<T>(), sp, nullptr, nullptr, out_lat, nullptr);
    }
    const T* cls_table = h->cfg.model_type == VLG_C2I ? W<T>("cls_embedding.embedding_table.weight") : nullptr;
    afm = afm_ok();
    VLG_TRY(gather_session_rows<T>(cls_table, h->cfg.num_classes + 1, W<T>("tok_embeddings.weight"), h->V, row_cls, ln->cur_tok.as<int32_t>(),
                                   reinterpret_cast<const T*>(pending), ln->x.as<T>(), Bp, D, st, afm ? D * (int)sizeof(T) / 64 : 0));
    if (fused_decode_ok()) {
      VLG_TRY(layers_fused());
      return head_fused(sp, nullptr, out_ids, nullptr, nullptr);
    }
    VLG_TRY(layers(1, S - 1));
    return head(ln->xn.as<T>(), sp, nullptr, out_ids, nullptr, nullptr);
  }

  int prefill(const void* d_cond, const vlg_sampling_params& sp, const float* noise, int32_t* out_ids, float* out_lat, float* trace) {
    const int D = h->D, Tc = h->Tc;
    VLG_TRY(set_state(state(), 0, 0, st));
    if (h->cfg.model_type == VLG_C2I) {
      VLG_TRY(gather_rows_i64<T>(W<T>("cls_embedding.embedding_table.weight"), (const int64_t*)d_cond + b0, B, h->cfg.num_classes,
                                 ln->x.as<T>(), Bp, D, h->cfg.num_classes + 1, st));
    } else {
      const int M = Bp * Tc;
      VLG_TRY(build_text_cond<T>((const float*)d_cond + (size_t)b0 * Tc * h->cd, W<T>("cls_embedding.uncond_embedding"), ln->condT.as<T>(),
                                 B, Bp, Tc, h->cd, st));
      VLG_TRY(linear(ln->condT.as<T>(), "cls_embedding.cap_proj.fc1.weight", ln->t1.as<T>(), nullptr, M, D, h->cd, ACT_GELU_TANH));
      VLG_TRY(linear(ln->t1.as<T>(), "cls_embedding.cap_proj.fc2.weight", ln->x.as<T>(), nullptr, M, D, D, ACT_NONE));
    }
    VLG_TRY(layers(Tc, Tc - 1));
    const T* hl = ln->xn.as<T>();
    if (Tc > 1) {
      VLG_TRY(take_last_rows<T>(ln->xn.as<T>(), ln->hl.as<T>(), Bp, Tc, D, st));
      hl = ln->hl.as<T>();
    }
    VLG_TRY(head(hl, sp, noise, out_ids, out_lat, trace));
    VLG_TRY(teacher());   // state->step is 0 here: the first decode step is fed forced[:, 0]
    return set_state(state(), Tc + h->pos_offset, 1, st);   // pos_offset > 0: the rows in between are zero (generate_impl)
  }
};

int reserve_lane(vlg_gpt* h, Lane& ln, int B, int Bp, int S, int pool_blocks = 0, int kv_block = 0) {
  const int Tc = h->Tc, D = h->D, H = h->H, hd = h->hd, F = h->F;
  const int M = Bp * Tc;  // prefill rows
  const size_t e = h->esz;
  const size_t kv_elems = pool_blocks > 0 ? (size_t)h->L * pool_blocks * H * kv_block * hd : (size_t)h->L * Bp * H * S * hd;
  VLG_TRY(ln.kcache.reserve(kv_elems * e));
  VLG_TRY(ln.vcache.reserve(kv_elems * e));
  size_t wsf = 0;
  auto need = [&](int m, int n, int k) { wsf = std::max(wsf, gemm_ws_floats(m, n, k, (int)e)); };
  for (int m : {M, Bp}) {
    need(m, 3 * D, D);
    need(m, D, D);
    need(m, 2 * F, D);
    need(m, D, F);
    need(m, D, h->cd > 0 ? h->cd : D);
  }
  need(Bp, h->V > 0 ? h->V : D, D);
  need(Bp, D, h->C > 0 ? h->C : D);
  need(Bp, h->C > 0 ? h->C : D, D);
  if (h->cfg.head == VLG_HEAD_HIDDEN) {
    const int Wd = h->dW, MR = (3 * h->dDepth + 2) * Wd;
    need(Bp, MR, Wd);
    need(Bp, Wd, D);
    need(Bp, Wd, h->C);
    need(Bp, 2 * h->C, Wd);
    need(h->dS, Wd, 256);
    need(h->dS, Wd, Wd);
    VLG_TRY(ln.d_cemb.reserve((size_t)Bp * Wd * e));
    VLG_TRY(ln.d_ys.reserve((size_t)h->dS * Bp * Wd * e));
    VLG_TRY(ln.d_mod.reserve((size_t)h->dS * Bp * MR * e));
    VLG_TRY(ln.d_x2.reserve((size_t)Bp * h->C * e));
    VLG_TRY(ln.d_h.reserve((size_t)Bp * Wd * e));
    VLG_TRY(ln.d_g.reserve((size_t)Bp * Wd * e));
    VLG_TRY(ln.d_g1.reserve((size_t)Bp * Wd * e));
    VLG_TRY(ln.d_out.reserve((size_t)Bp * 2 * h->C * e));
    VLG_TRY(ln.d_x.reserve((size_t)Bp * h->C * e));
    VLG_TRY(ln.dp_xbuf.reserve(dl_persist_xbuf_bytes(Bp, Wd, (int)e)));
  }
  VLG_TRY(ln.ws.reserve(wsf * sizeof(float)));
  VLG_TRY(ln.attn_ws.reserve(attn_ws_floats(M, H, hd) * sizeof(float)));
  const size_t Mp = (size_t)std::max(M, round_up(Bp, 16));   // A-fragment-major activations pad their rows to whole 16-row tiles
  VLG_TRY(ln.x.reserve(Mp * D * e));
  VLG_TRY(ln.xn.reserve((size_t)M * D * e));
  VLG_TRY(ln.q.reserve((size_t)M * D * e));
  VLG_TRY(ln.ao.reserve(Mp * D * e));
  VLG_TRY(ln.g.reserve(Mp * F * e));
  VLG_TRY(ln.t1.reserve((size_t)M * D * e));
  VLG_TRY(ln.hl.reserve((size_t)Bp * D * e));
  if (h->cd > 0) VLG_TRY(ln.condT.reserve((size_t)M * h->cd * e));
  if (h->C > 0) {
    VLG_TRY(ln.latT.reserve((size_t)Bp * h->C * e));
    VLG_TRY(ln.y.reserve((size_t)Bp * h->C * e));
    VLG_TRY(ln.cur_lat.reserve((size_t)B * h->C * sizeof(float)));
  }
  if (h->V > 0) VLG_TRY(ln.logits.reserve((size_t)Bp * h->V * sizeof(float)));
  VLG_TRY(ln.cur_tok.reserve((size_t)Bp * sizeof(int32_t)));
  VLG_TRY(ln.state.reserve(sizeof(StepState)));
  if (Bp <= 32 && pool_blocks == 0) VLG_TRY(ln.pd_xbuf.reserve(pd_xbuf_bytes(Bp, D, H, hd, F, (int)e)));
  if (!ln.st) VLG_HIP(hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking));
  if (!ln.ev) VLG_HIP(hipEventCreateWithFlags(&ln.ev, hipEventDisableTiming));
  return VLG_OK;
}

// Persistent kernels (decode step, DiffLoss sampler) want every compute unit; two of them in flight on one device can starve each other
// until their bounded waits run out (VLG_ERR_STATE).  Inside ONE process the decode loops of different handles / host threads are
// therefore chained on the device: a loop waits for the event the previous loop recorded behind its last step.  (Other processes on
// the same GPU are out of reach: include/vlg.h, "EXCLUSIVITY".)
struct PersistGate {
  std::mutex mu;
  hipEvent_t ev[64] = {};
};
static PersistGate& persist_gate() {
  static PersistGate g;
  return g;
}

template <typename T>
int generate_impl(vlg_gpt* h, const void* d_cond, const float* d_mask, int B, int N, const vlg_sampling_params& sp,
                  const float* d_noise, int32_t* out_ids, float* out_lat, float* trace, hipStream_t caller) {
  const bool cfg_on = sp.cfg_scale > 1.0f;
  const int Tc = h->Tc, D = h->D, F = h->F;
  const int S = round_up(Tc + h->pos_offset + N, 8);  // gpt.py:322 (+ the benchmark's position offset, normally 0)
  VLG_CHECK(Tc + h->pos_offset + N <= h->npos, VLG_ERR_BAD_SHAPE, "max_new_tokens %d exceeds the RoPE table (%d positions after %d cond tokens)", N,
            h->npos - Tc - h->pos_offset, Tc + h->pos_offset);
  if (cfg_on && h->cfg.model_type != VLG_C2I)
    VLG_CHECK(Tc == 120, VLG_ERR_BAD_SHAPE, "CFG needs cls_token_num == 120 (uncond_embedding is [120, caption_dim], gpt.py:96)");
  for (auto& kv : h->w) VLG_CHECK(kv.second.loaded, VLG_ERR_STATE, "weight %s was never loaded", kv.first.c_str());

  const bool latent_out = h->cfg.head != VLG_HEAD_LOGITS;
  const size_t out_bytes = latent_out ? (size_t)B * N * h->C * sizeof(float) : (size_t)B * N * sizeof(int32_t);
  void* user_out = latent_out ? (void*)out_lat : (void*)out_ids;
  VLG_TRY(h->outbuf.reserve(out_bytes));
  if (latent_out)
    out_lat = h->outbuf.as<float>();
  else
    out_ids = h->outbuf.as<int32_t>();
  Lane* ln = &h->lane;
  VLG_TRY(reserve_lane(h, *ln, B, cfg_on ? 2 * B : B, S));
  VLG_TRY(ensure_fm<T>(h));
  VLG_TRY(ensure_pd_layers(h));
  if (d_mask) VLG_TRY(ln->maskbuf.reserve((size_t)B * Tc * sizeof(float)));
  Runner<T> r{h, ln, ln->st, B, cfg_on ? 2 * B : B, N, S, 0, B, d_mask ? ln->maskbuf.as<float>() : nullptr};
  if (h->cfg.head == VLG_HEAD_HIDDEN) {
    VLG_CHECK(!cfg_on, VLG_ERR_UNSUPPORTED, "the DiffLoss head runs with cfg_scale = 1 only (generate_video_diff.py:112-137 never assigns the CFG branch)");
    if (!h->dtemb_ready) VLG_TRY(r.build_time_table());
  }

  // ---- fork from the caller's stream ----------------------------------------------------------------------------
  VLG_HIP(hipEventRecord(h->ev_in, caller));
  VLG_HIP(hipStreamWaitEvent(r.st, h->ev_in, 0));
  if (d_mask) VLG_HIP(hipMemcpyAsync(ln->maskbuf.p, d_mask, (size_t)B * Tc * sizeof(float), hipMemcpyDeviceToDevice, r.st));
  if (ln->pd_xbuf.p) VLG_HIP(hipMemsetAsync(ln->pd_xbuf.p, 0, ln->pd_xbuf.bytes, r.st));   // hand-off tags count up from the call's first step
  if (h->pos_offset > 0) {   // late-context timing: cache rows Tc .. Tc + offset - 1 are read by every step and never written
    VLG_HIP(hipMemsetAsync(ln->kcache.p, 0, ln->kcache.bytes, r.st));
    VLG_HIP(hipMemsetAsync(ln->vcache.p, 0, ln->vcache.bytes, r.st));
  }
  hipStream_t s0 = r.st;
  const int steps = N - 1;
  // The gate covers EVERY persistent launch of the call: the DiffLoss head runs its persistent sampler already in the prefill (the
  // head of token 0), also when N == 1; the persistent decode step only runs in the decode loop.
  std::unique_lock<std::mutex> gate_lock;
  hipEvent_t gate_ev = nullptr;
  static const bool gate_off = getenv("VLG_PERSIST_GATE") != nullptr && atoi(getenv("VLG_PERSIST_GATE")) == 0;   // tests: show what the gate prevents
  const bool persist_head = h->cfg.head == VLG_HEAD_HIDDEN && h->dl_persist_on;
  if (!gate_off && (persist_head || (steps > 0 && h->pdecode))) {
    int dev = 0;
    VLG_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64) {
      PersistGate& g = persist_gate();
      gate_lock = std::unique_lock<std::mutex>(g.mu);
      if (!g.ev[dev]) VLG_HIP(hipEventCreateWithFlags(&g.ev[dev], hipEventDisableTiming));
      gate_ev = g.ev[dev];
      VLG_HIP(hipStreamWaitEvent(s0, gate_ev, 0));   // the previous call of this process on this device (any handle, any thread)
    }
  }
  // an error between here and the record below must still record the event behind whatever was enqueued: the next holder waits on it
  struct GateRelease {
    hipEvent_t ev;
    hipStream_t st;
    ~GateRelease() {
      if (ev) (void)hipEventRecord(ev, st);
    }
  } gate_release{gate_ev, s0};
  VLG_TRY(r.prefill(d_cond, sp, d_noise, out_ids, out_lat, trace));
  if (steps > 0) {
    if (h->time_attn) {
      // eager loop; HIP events (on the launch stream) around layer 0's attention kernel
      while ((int)h->attn_ev.size() < 2 * steps) {
        hipEvent_t ev;
        VLG_HIP(hipEventCreate(&ev));
        h->attn_ev.push_back(ev);
      }
      for (int i = 0; i < steps; ++i) {
        r.ev_slot = i;
        VLG_TRY(r.decode_step(sp, d_noise, out_ids, out_lat, trace));
      }
      r.ev_slot = -1;
      VLG_HIP(hipStreamSynchronize(r.st));
      h->attn_ms_sum = 0;
      h->attn_bytes_sum = 0;
      for (int i = 0; i < steps; ++i) {
        float ms = 0.f;
        VLG_HIP(hipEventElapsedTime(&ms, h->attn_ev[2 * i], h->attn_ev[2 * i + 1]));
        h->attn_ms_sum += ms;
        h->attn_bytes_sum += 2.0 * r.Bp * D * (double)(Tc + i + 1) * h->esz;  // K and V rows 0..p of one layer
      }
      h->attn_launches = steps;
      // what the bracket itself costs: back-to-back event pairs with nothing between them, same stream, same session
      {
        const int ncal = std::min(64, steps);
        for (int i = 0; i < ncal; ++i) {
          VLG_HIP(hipEventRecord(h->attn_ev[2 * i], s0));
          VLG_HIP(hipEventRecord(h->attn_ev[2 * i + 1], s0));
        }
        VLG_HIP(hipStreamSynchronize(s0));
        double tot = 0;
        for (int i = 0; i < ncal; ++i) {
          float ms = 0.f;
          VLG_HIP(hipEventElapsedTime(&ms, h->attn_ev[2 * i], h->attn_ev[2 * i + 1]));
          tot += ms;
        }
        h->attn_pair_overhead_ms = ncal > 0 ? tot / ncal : 0.0;
      }
    } else if (h->use_graph) {
      // one graph = one decode step
      auto fbits = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        return (uint64_t)u;
      };
      std::vector<uint64_t> key = {(uint64_t)B, (uint64_t)N, (uint64_t)S, (uint64_t)h->dtype, fbits(sp.cfg_scale),
                                   (uint64_t)(int64_t)sp.cfg_interval, fbits(sp.temperature), (uint64_t)(int64_t)sp.top_k, fbits(sp.top_p),
                                   (uint64_t)sp.sample_logits, sp.seed, (uint64_t)(uintptr_t)d_noise, (uint64_t)(uintptr_t)trace,
                                   (uint64_t)(uintptr_t)h->outbuf.p, (uint64_t)(uintptr_t)h->dtemb.p, (uint64_t)(uintptr_t)h->dadaln_bias.p,
                                   (uint64_t)((h->fuse_gemm ? 1 : 0) | (h->fuse_swiglu ? 2 : 0) | (h->pdecode ? 4 : 0) | (d_mask ? 32 : 0) | (h->dl_persist_on ? 128 : 0) | ((uint64_t)h->dl_rows << 8)),
                                   (uint64_t)(uintptr_t)h->dcoef_dev.p, (uint64_t)__builtin_bit_cast(uint32_t, h->cfg_iter), (uint64_t)(uintptr_t)r.st,
                                   (uint64_t)h->spin_max, (uint64_t)h->pd_rows, (uint64_t)h->pos_offset, (uint64_t)(uintptr_t)h->teach_ids, (uint64_t)(uintptr_t)h->teach_lat, (uint64_t)((h->weights_fm ? 1 : 0) | (h->act_fm ? 2 : 0))};
      {
        const auto pk = ln->ptr_key();
        key.insert(key.end(), pk.begin(), pk.end());
      }
      if (!h->gc.exec || h->gc.key != key) {
        h->gc.drop();
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        VLG_HIP(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        const int rc = r.decode_step(sp, d_noise, out_ids, out_lat, trace);
        hipError_t ee = hipStreamEndCapture(s0, &graph);
        if (rc != VLG_OK) {
          if (graph) (void)hipGraphDestroy(graph);
          return rc;
        }
        VLG_HIP(ee);
        hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
          (void)hipGraphDestroy(graph);
          VLG_HIP(ie);
        }
        h->gc.graph = graph;
        h->gc.exec = exec;
        h->gc.key = key;
        h->gc.stream = s0;
        h->graphs_built += 1;
      }
      for (int i = 0; i < steps; ++i) VLG_HIP(hipGraphLaunch(h->gc.exec, s0));   // no host wait: the call returns with the work enqueued
    } else {
      for (int i = 0; i < steps; ++i) VLG_TRY(r.decode_step(sp, d_noise, out_ids, out_lat, trace));
    }
  }
  if (gate_ev) {
    VLG_HIP(hipEventRecord(gate_ev, s0));
    gate_release.ev = nullptr;
    gate_lock.unlock();
  }
  // ---- join back into the caller's stream ---------------------------------------------------------------------------
  VLG_HIP(hipEventRecord(ln->ev, r.st));
  VLG_HIP(hipStreamWaitEvent(caller, ln->ev, 0));
  VLG_HIP(hipMemcpyAsync(user_out, h->outbuf.p, out_bytes, hipMemcpyDeviceToDevice, caller));

  // ---- algorithmic bytes of this call; weights are streamed once per lane and step ------------------
  const int Bp = cfg_on ? 2 * B : B;
  const size_t e = h->esz;
  double P = (double)h->L * (4.0 * D * D + 3.0 * D * F);
  if (h->cfg.head == VLG_HEAD_LOGITS) P += (double)D * h->V;
  if (h->cfg.model_type == VLG_T2V) P += 2.0 * D * D + 2.0 * D * h->C;
  h->bytes_w = (double)e * P * N;
  double kv = 0;
  for (int i = 0; i < N; ++i) {
    const double p = (i == 0) ? (Tc - 1) : (Tc + i - 1);
    kv += 2.0 * h->L * Bp * D * (p + 1) * e + 2.0 * h->L * Bp * D * e * (i == 0 ? Tc : 1);
  }
  h->bytes_kv = kv;
  h->bytes_other = (h->cfg.head == VLG_HEAD_LOGITS) ? 4.0 * Bp * h->V * N : 0.0;
  return VLG_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Iteration-level batching (the request front-end; reference: vLLM's model_runner.execute_model as used by
// autoregressive/serve/).  A session owns `rows` KV-cache slots of max_new_tokens + 1 positions.  Each step advances EVERY slot by
// one token, each at its own position: a slot starts a request (class id given: the c2i condition token at position 0),
// continues one, or idles.  One captured hipGraph per session; positions, token indices and inputs live in device arrays.
// ---------------------------------------------------------------------------------------------------------------
struct vlg_gpt::Session {
  Lane ln;
  int R = 0, Rp = 0, S = 0, maxN = 0;
  bool cfg = false;
  vlg_sampling_params sp{};
  DevBuf row_all, out_ids;            // row_all: [3][Rp] int32 = per-row position | token index | start code, refreshed by ONE copy per iteration
  int32_t* d_pos() const { return row_all.as<int32_t>(); }
  int32_t* d_step() const { return row_all.as<int32_t>() + Rp; }
  int32_t* d_cls() const { return row_all.as<int32_t>() + 2 * (size_t)Rp; }
  DevBuf out_lat;                     // continuous-latent models: [R][maxN][C] fp32 instead of out_ids
  DevBuf maskbuf, pending;            // text-conditioned models: [R][Tc] fp32 condition masks; [Rp][D] input rows of starting slots
  std::vector<char> prefilled;        // slot has a condition in its KV rows and waits for its first step
  std::vector<int32_t> pos;           // host mirror: -1 = idle, else input position of the slot's next step
  std::vector<int32_t> h_pos, h_step, h_cls;
  // per-iteration inputs travel through a ring of pinned staging slots [kStage][3][Rp], each guarded by an event: session_step returns once
  // the iteration is enqueued and the host prepares the next one while the device runs (a slot is reused only after its copies were consumed)
  static constexpr int kStage = 4;
  int32_t* h_stage = nullptr;
  hipEvent_t ev_stage[kStage] = {};
  bool stage_used[kStage] = {};
  int stage_i = 0;
  // block-granular cache (option kv_block): table [Rp][nblk_row] of pool block ids, 0 = the scratch block idle rows run on
  int bs_shift = 0, nblk_row = 0, pool_blocks = 0;
  DevBuf btab;
  std::vector<int32_t> h_btab, free_blocks;
  std::vector<int32_t> reserved;      // per slot: positions covered by its blocks
  bool btab_dirty = false;
  bool paged() const { return pool_blocks > 0; }
  KvPages pages(int row0 = 0) const {
    KvPages pg;
    if (paged()) {
      pg.table = btab.as<int32_t>() + (size_t)row0 * nblk_row;
      pg.stride = nblk_row;
      pg.shift = bs_shift;
    }
    return pg;
  }
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  ~Session() {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (h_stage) (void)hipHostFree(h_stage);
    for (hipEvent_t e : ev_stage)
      if (e) (void)hipEventDestroy(e);
  }
};

namespace {
template <typename T>
int session_begin_impl(vlg_gpt* h, int R, int maxN, const vlg_sampling_params& sp) {
  VLG_TRY(ensure_fm<T>(h));
  auto ses = std::make_unique<vlg_gpt::Session>();
  ses->R = R;
  ses->cfg = sp.cfg_scale > 1.0f;
  ses->Rp = ses->cfg ? 2 * R : R;
  ses->maxN = maxN;
  ses->S = round_up(h->Tc + maxN, 8);
  ses->sp = sp;
  if (h->kv_block > 0) {
    int sh = 0;
    while ((1 << sh) < h->kv_block) ++sh;
    ses->bs_shift = sh;
    ses->nblk_row = cdiv(ses->S, h->kv_block);
    VLG_CHECK(ses->nblk_row <= 256, VLG_ERR_BAD_SHAPE, "kv_block %d: %d blocks per row, at most 256", h->kv_block, ses->nblk_row);
    ses->pool_blocks = h->kv_pool_blocks > 0 ? h->kv_pool_blocks : ses->Rp * ses->nblk_row + 1;
    VLG_CHECK(ses->pool_blocks >= 2, VLG_ERR_BAD_ARG, "kv_pool_blocks %d: the pool needs the scratch block and at least one more", ses->pool_blocks);
    // the lane's cache buffers hold L pools of pool_blocks * H * BS * hd elements: expressed as (rows = pool_blocks, S = BS)
    VLG_TRY(reserve_lane(h, ses->ln, R, ses->Rp, ses->S, ses->pool_blocks, h->kv_block));
    ses->h_btab.assign((size_t)ses->Rp * ses->nblk_row, 0);
    VLG_TRY(ses->btab.reserve(ses->h_btab.size() * sizeof(int32_t)));
    VLG_HIP(hipMemset(ses->btab.p, 0, ses->h_btab.size() * sizeof(int32_t)));
    for (int i = ses->pool_blocks - 1; i >= 1; --i) ses->free_blocks.push_back(i);   // block 0 = scratch
    ses->reserved.assign(R, 0);
  } else {
    VLG_TRY(reserve_lane(h, ses->ln, R, ses->Rp, ses->S));
  }
  const size_t nb = (size_t)ses->Rp * sizeof(int32_t);
  VLG_TRY(ses->row_all.reserve(3 * nb));
  VLG_TRY(ses->out_ids.reserve((size_t)R * maxN * sizeof(int32_t)));
  VLG_HIP(hipMemset(ses->row_all.p, 0, 3 * nb));
  VLG_HIP(hipMemset(ses->ln.cur_tok.p, 0, (size_t)ses->Rp * sizeof(int32_t)));
  VLG_HIP(hipMemset(ses->out_ids.p, 0, (size_t)R * maxN * sizeof(int32_t)));
  if (h->cfg.model_type == VLG_T2V) {
    VLG_TRY(ses->out_lat.reserve((size_t)R * maxN * h->C * sizeof(float)));
    VLG_HIP(hipMemset(ses->out_lat.p, 0, (size_t)R * maxN * h->C * sizeof(float)));
    VLG_HIP(hipMemset(ses->ln.cur_lat.p, 0, (size_t)ses->Rp * h->C * sizeof(float)));   // idle rows feed a zero latent through the adapter
  }
  ses->pos.assign(R, -1);
  ses->prefilled.assign(R, 0);
  ses->h_pos.assign(ses->Rp, 0);
  ses->h_step.assign(ses->Rp, 0);
  const bool text = h->cfg.model_type != VLG_C2I;
  ses->h_cls.assign(ses->Rp, text ? -3 : h->cfg.num_classes);
  hipStream_t st = ses->ln.st;
  if (text) {   // every slot starts with an all-valid mask and a zero input row (idle slots park on it)
    VLG_TRY(ses->maskbuf.reserve((size_t)R * h->Tc * sizeof(float)));
    VLG_TRY(ses->pending.reserve((size_t)ses->Rp * h->D * sizeof(T)));
    std::vector<float> ones((size_t)R * h->Tc, 1.0f);
    VLG_HIP(hipMemcpy(ses->maskbuf.p, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
    VLG_HIP(hipMemset(ses->pending.p, 0, (size_t)ses->Rp * h->D * sizeof(T)));
  }
  VLG_TRY(set_state(ses->ln.state.as<StepState>(), 0, 0, st));
  Runner<T> r{h, &ses->ln, st, R, ses->Rp, maxN, ses->S, 0, R, text ? ses->maskbuf.as<float>() : nullptr};
  r.row_pos = ses->d_pos();
  r.row_step = ses->d_step();
  r.pending = ses->pending.p;
  r.pages = ses->pages();
  r.pool_blocks = ses->pool_blocks;
  // the DiffLoss head's per-handle tables (time embeddings of the respaced steps, fp32 modulation bias, DDPM coefficients on the device):
  // generate() builds them on first use - a handle whose first call is a session needs them just the same
  if (h->cfg.head == VLG_HEAD_HIDDEN && !h->dtemb_ready) {
    VLG_TRY(r.build_time_table());
    VLG_HIP(hipStreamSynchronize(st));
  }
  VLG_CHECK(h->cfg.head != VLG_HEAD_HIDDEN || (h->dtemb.p && h->dadaln_bias.p && h->dcoef_dev.p), VLG_ERR_STATE, "DiffLoss tables missing");
  if (h->use_graph) {
    VLG_HIP(hipStreamSynchronize(st));
    VLG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = r.session_step(ses->sp, ses->d_cls(), ses->out_ids.as<int32_t>(), ses->out_lat.as<float>());
    hipError_t ee = hipStreamEndCapture(st, &ses->graph);
    VLG_TRY(rc);
    VLG_HIP(ee);
    VLG_HIP(hipGraphInstantiate(&ses->exec, ses->graph, nullptr, nullptr, 0));
  }
  h->ses = std::move(ses);
  return VLG_OK;
}

template <typename T>
int session_step_impl(vlg_gpt* h, const int32_t* h_row_class) {
  vlg_gpt::Session& s = *h->ses;
  const int R = s.R, null_cls = h->cfg.num_classes;
  const bool text = h->cfg.model_type != VLG_C2I;
  const int first = h->Tc - 1;         // input position of the step that samples token 0 (the last condition token)
  // pass 1: validate every slot's code; nothing is changed before the whole step is known to be well-formed (a failure for slot b must
  // not leave slots < b advanced with nothing launched)
  for (int b = 0; b < R; ++b) {
    const int c = h_row_class[b];
    if (s.paged() && (c >= 0 || c == -3 || (c == -1 && s.pos[b] >= 0))) {
      const int next = (c == -1) ? s.pos[b] + 1 : (text ? first : 0);
      VLG_CHECK(next < s.reserved[b], VLG_ERR_STATE, "slot %d: position %d is outside its reserved KV blocks (%d positions; vlg_gpt_session_reserve)",
                b, next, s.reserved[b]);
    }
    if (c >= 0 && !text) {
      VLG_CHECK(c <= null_cls, VLG_ERR_BAD_ARG, "class id %d out of range", c);
    } else if (c == -3 && text) {
      VLG_CHECK(s.prefilled[b], VLG_ERR_STATE, "slot %d has no prefilled condition", b);
    } else if (c == -1 && s.pos[b] >= 0) {
      VLG_CHECK(s.pos[b] + 1 < first + s.maxN, VLG_ERR_BAD_SHAPE, "slot %d stepped past max_new_tokens %d", b, s.maxN);
    } else {
      VLG_CHECK(c < 0, VLG_ERR_BAD_ARG, "slot %d: start code %d does not fit this model type", b, c);
    }
  }
  // pass 2: advance the host mirror
  for (int b = 0; b < R; ++b) {
    const int c = h_row_class[b];
    int cls = -1, cls_partner = -1;
    if (c >= 0 && !text) {             // start a class-conditional request in this slot
      s.pos[b] = 0;
      cls = c;
      cls_partner = null_cls;          // unconditional partner row: null class at the start (generate.py:131)
    } else if (c == -3 && text) {      // start the request whose condition vlg_gpt_session_prefill put into this slot
      s.prefilled[b] = 0;
      s.pos[b] = first;
      cls = cls_partner = -3;          // input row = the projected last condition token (cond / uncond), left in `pending`
    } else if (c == -1 && s.pos[b] >= 0) {   // continue
      s.pos[b] += 1;
    } else {                           // idle (or told to stop): park the slot at position 0 on the null class / a zero row
      s.pos[b] = -1;
      cls = cls_partner = text ? -3 : null_cls;
      if (text && s.prefilled[b]) cls = cls_partner = -1;   // do not step on a waiting slot's KV row 0: feed a token row at ITS position
    }
    int p = s.pos[b] < 0 ? 0 : s.pos[b];
    if (text && s.pos[b] < 0 && s.prefilled[b]) p = first;  // a prefilled, not yet started slot idles at its first position
    s.h_pos[b] = p;
    s.h_step[b] = p >= first ? p - first : 0;   // token index
    s.h_cls[b] = cls;
    if (s.cfg) {
      s.h_pos[b + R] = p;
      s.h_step[b + R] = s.h_step[b];
      s.h_cls[b + R] = cls_partner;
    }
  }
  hipStream_t st = s.ln.st;
  const size_t nb = (size_t)s.Rp * sizeof(int32_t);
  static const bool sync_steps = getenv("VLG_SESSION_SYNC") != nullptr && atoi(getenv("VLG_SESSION_SYNC")) != 0;   // A/B knob: wait for every iteration
  bool wait_here = sync_steps;
  if (s.h_stage == nullptr) {
    VLG_HIP(hipHostMalloc(reinterpret_cast<void**>(&s.h_stage), (size_t)vlg_gpt::Session::kStage * 3 * nb, hipHostMallocDefault));
    for (hipEvent_t& e : s.ev_stage) VLG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int slot = s.stage_i % vlg_gpt::Session::kStage;
  if (s.stage_used[slot]) VLG_HIP(hipEventSynchronize(s.ev_stage[slot]));   // the iteration that last used this slot has consumed it
  int32_t* stg = s.h_stage + (size_t)slot * 3 * s.Rp;
  memcpy(stg, s.h_pos.data(), nb);
  memcpy(stg + s.Rp, s.h_step.data(), nb);
  memcpy(stg + 2 * (size_t)s.Rp, s.h_cls.data(), nb);
  VLG_HIP(hipMemcpyAsync(s.row_all.p, stg, 3 * nb, hipMemcpyHostToDevice, st));   // one copy: the three arrays sit side by side on both ends
  VLG_HIP(hipEventRecord(s.ev_stage[slot], st));
  s.stage_used[slot] = true;
  s.stage_i += 1;
  if (s.btab_dirty) {   // the block table is copied from pageable host memory that reserve / release keep editing: wait for it below
    VLG_HIP(hipMemcpyAsync(s.btab.p, s.h_btab.data(), s.h_btab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    s.btab_dirty = false;
    wait_here = true;
  }
  // a session of the DiffLoss head launches the persistent sampler: the same process-wide gate as generate() (one persistent launch on the
  // device at a time; held while this iteration is enqueued, the event recorded behind it is what the next holder waits for)
  std::unique_lock<std::mutex> gate_lock;
  struct GateRelease {
    hipEvent_t ev = nullptr;
    hipStream_t st = nullptr;
    ~GateRelease() {
      if (ev) (void)hipEventRecord(ev, st);
    }
  } gate_release;
  static const bool gate_off = getenv("VLG_PERSIST_GATE") != nullptr && atoi(getenv("VLG_PERSIST_GATE")) == 0;
  if (!gate_off && h->cfg.head == VLG_HEAD_HIDDEN && h->dl_persist_on) {
    int dev = 0;
    VLG_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64) {
      PersistGate& g = persist_gate();
      gate_lock = std::unique_lock<std::mutex>(g.mu);
      if (!g.ev[dev]) VLG_HIP(hipEventCreateWithFlags(&g.ev[dev], hipEventDisableTiming));
      VLG_HIP(hipStreamWaitEvent(st, g.ev[dev], 0));
      gate_release.ev = g.ev[dev];
      gate_release.st = st;
    }
  }
  if (s.exec) {
    VLG_HIP(hipGraphLaunch(s.exec, st));
  } else {
    Runner<T> r{h, &s.ln, st, R, s.Rp, s.maxN, s.S, 0, R, text ? s.maskbuf.as<float>() : nullptr};
    r.row_pos = s.d_pos();
    r.row_step = s.d_step();
    r.pending = s.pending.p;
    r.pages = s.pages();
    r.pool_blocks = s.pool_blocks;
    VLG_TRY(r.session_step(s.sp, s.d_cls(), s.out_ids.as<int32_t>(), s.out_lat.as<float>()));
  }
  // the iteration is enqueued; its inputs sit in a pinned staging slot of their own, so the host may go on (session_read / session_end wait
  // for the stream, everything else a caller does with the session is ordered behind it on the same stream)
  if (wait_here) VLG_HIP(hipStreamSynchronize(st));
  return VLG_OK;
}
}  // namespace

namespace {
// Condition of ONE request into the KV rows of its slot (and of its unconditional partner under guidance): positions 0 .. Tc-2 run
// through the prefill kernels (slab GEMMs, masked causal attention) against the slot's cache rows; the projected LAST condition token
// is kept as the slot's pending input row, so the step that samples token 0 is an ordinary iteration of the batch at position Tc-1.
template <typename T>
int session_prefill_impl(vlg_gpt* h, int slot, int n, const float* d_cond, const float* d_mask) {   // slots slot .. slot + n - 1
  vlg_gpt::Session& s = *h->ses;
  const int Tc = h->Tc, D = h->D, cd = h->cd;
  hipStream_t st = s.ln.st;
  float* mrow = s.maskbuf.as<float>() + (size_t)slot * Tc;
  if (s.paged())
    for (int i = 0; i < n; ++i)
      VLG_CHECK(s.reserved[slot + i] >= Tc, VLG_ERR_STATE, "slot %d: reserve its KV blocks first (vlg_gpt_session_reserve)", slot + i);
  // arguments are valid: whatever ran in the slots is over, the caller reuses them (their tokens were read with session_read)
  for (int i = 0; i < n; ++i) s.pos[slot + i] = -1;
  if (d_mask) {
    VLG_HIP(hipMemcpyAsync(mrow, d_mask, (size_t)n * Tc * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    std::vector<float> ones((size_t)n * Tc, 1.0f);
    VLG_HIP(hipMemcpyAsync(mrow, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice, st));
    VLG_HIP(hipStreamSynchronize(st));
  }
  if (s.paged()) {
    if (s.btab_dirty) {
      VLG_HIP(hipMemcpyAsync(s.btab.p, s.h_btab.data(), s.h_btab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
      VLG_HIP(hipStreamSynchronize(st));
      s.btab_dirty = false;
    }
  }
  for (int pass = 0; pass < (s.cfg ? 2 : 1); ++pass) {
    Runner<T> r{h, &s.ln, st, n, n, s.maxN, s.S, 0, n, mrow};
    r.kv_row0 = slot + pass * s.R;
    r.kv_rows = s.Rp;
    r.pages = s.pages(r.kv_row0);
    r.pool_blocks = s.pool_blocks;
    VLG_TRY(set_state(r.state(), 0, 0, st));
    // pass 0: the requests' caption features; pass 1: uncond_embedding (generate.py:138-139)
    VLG_TRY(build_text_cond<T>(d_cond, r.template W<T>("cls_embedding.uncond_embedding"), s.ln.condT.as<T>(), pass == 0 ? n : 0, n, Tc, cd, st));
    VLG_TRY(r.linear(s.ln.condT.as<T>(), "cls_embedding.cap_proj.fc1.weight", s.ln.t1.as<T>(), nullptr, n * Tc, D, cd, ACT_GELU_TANH));
    // the projected condition rows [n][Tc][D]: the last of each slot is kept as its pending input, the others run through the layers
    T* proj = n > 1 ? s.ln.xn.as<T>() : s.ln.x.as<T>();
    VLG_TRY(r.linear(s.ln.t1.as<T>(), "cls_embedding.cap_proj.fc2.weight", proj, nullptr, n * Tc, D, D, ACT_NONE));
    VLG_TRY(take_last_rows<T>(proj, s.pending.as<T>() + (size_t)r.kv_row0 * D, n, Tc, D, st));
    if (n > 1) VLG_TRY(drop_last_rows<T>(proj, s.ln.x.as<T>(), n, Tc, D, st));   // [n][Tc - 1][D], what layers(Tc - 1, ...) walks
    if (Tc > 1) VLG_TRY(r.layers(Tc - 1, Tc - 2));
  }
  VLG_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) s.prefilled[slot + i] = 1;
  return VLG_OK;
}
}  // namespace

extern "C" int vlg_gpt_session_prefill_batch(vlg_gpt_t* h, int32_t first_slot, int32_t n, const float* d_cond, const float* d_mask) {
  VLG_CHECK(h && d_cond, VLG_ERR_BAD_ARG, "vlg_gpt_session_prefill: null argument");
  VLG_CHECK(h->ses != nullptr, VLG_ERR_STATE, "no open session");
  VLG_CHECK(h->cfg.model_type == VLG_T2I || h->cfg.model_type == VLG_T2V, VLG_ERR_UNSUPPORTED, "vlg_gpt_session_prefill is for text-conditioned models");
  VLG_CHECK(n >= 1 && first_slot >= 0 && first_slot + n <= h->ses->R, VLG_ERR_BAD_ARG, "slots %d .. %d out of range", first_slot, first_slot + n - 1);
  return h->dtype == VLG_BF16 ? session_prefill_impl<bf16>(h, first_slot, n, d_cond, d_mask) : session_prefill_impl<float>(h, first_slot, n, d_cond, d_mask);
}

extern "C" int vlg_gpt_session_prefill(vlg_gpt_t* h, int32_t slot, const float* d_cond, const float* d_mask) {
  return vlg_gpt_session_prefill_batch(h, slot, 1, d_cond, d_mask);
}

extern "C" int vlg_gpt_set_option_f64(vlg_gpt_t* h, const char* key, double value) {
  VLG_CHECK(h && key, VLG_ERR_BAD_ARG, "vlg_gpt_set_option_f64: null argument");
  if (!strcmp(key, "cfg_iter")) {
    VLG_CHECK(value > 0.0 && value < 1e4, VLG_ERR_BAD_ARG, "cfg_iter %g out of range", value);
    VLG_CHECK(value == 1.0 || h->cfg.head == VLG_HEAD_HIDDEN, VLG_ERR_UNSUPPORTED, "cfg_iter is the DiffLoss head's guidance (hidden head only)");
    h->cfg_iter = (float)value;
    return VLG_OK;
  }
  set_error("vlg_gpt_set_option_f64: unknown key %s", key);
  return VLG_ERR_BAD_ARG;
}

extern "C" int vlg_gpt_session_reserve(vlg_gpt_t* h, int32_t slot, int32_t n_tokens) {
  VLG_CHECK(h && h->ses != nullptr, VLG_ERR_STATE, "vlg_gpt_session_reserve: no open session");
  vlg_gpt::Session& s = *h->ses;
  VLG_CHECK(slot >= 0 && slot < s.R && n_tokens > 0 && n_tokens <= s.maxN, VLG_ERR_BAD_ARG, "vlg_gpt_session_reserve: slot %d / %d tokens out of range", slot,
            n_tokens);
  if (!s.paged()) return VLG_OK;   // contiguous slots are sized for max_new_tokens already
  const int bs = 1 << s.bs_shift, rows = s.cfg ? 2 : 1;
  const int want = cdiv(h->Tc + n_tokens, bs), have = cdiv(s.reserved[slot], bs);
  if (want > have) {
    const int need = (want - have) * rows;
    if ((int)s.free_blocks.size() < need) {
      set_error("KV pool: slot %d needs %d more blocks, %d are free", slot, need, (int)s.free_blocks.size());
      return VLG_ERR_OOM;
    }
    for (int r = 0; r < rows; ++r)
      for (int j = have; j < want; ++j) {
        s.h_btab[(size_t)(slot + r * s.R) * s.nblk_row + j] = s.free_blocks.back();
        s.free_blocks.pop_back();
      }
    s.btab_dirty = true;
  }
  s.reserved[slot] = std::max(s.reserved[slot], std::min(want * bs, s.S));
  return VLG_OK;
}

extern "C" int vlg_gpt_session_release(vlg_gpt_t* h, int32_t slot) {
  VLG_CHECK(h && h->ses != nullptr, VLG_ERR_STATE, "vlg_gpt_session_release: no open session");
  vlg_gpt::Session& s = *h->ses;
  VLG_CHECK(slot >= 0 && slot < s.R, VLG_ERR_BAD_ARG, "vlg_gpt_session_release: slot %d out of range", slot);
  s.pos[slot] = -1;
  s.prefilled[slot] = 0;
  if (!s.paged()) return VLG_OK;
  // the steps that used these blocks are ahead on the session's stream of whatever a later owner of the blocks will enqueue there
  for (int r = 0; r < (s.cfg ? 2 : 1); ++r)
    for (int j = 0; j < s.nblk_row; ++j) {
      int32_t& e = s.h_btab[(size_t)(slot + r * s.R) * s.nblk_row + j];
      if (e != 0) s.free_blocks.push_back(e);
      e = 0;
    }
  s.reserved[slot] = 0;
  s.btab_dirty = true;
  return VLG_OK;
}

extern "C" int vlg_gpt_session_free_blocks(vlg_gpt_t* h, int32_t* n_free, int32_t* block_size) {
  VLG_CHECK(h && h->ses != nullptr && n_free, VLG_ERR_STATE, "vlg_gpt_session_free_blocks: no open session / null argument");
  *n_free = h->ses->paged() ? (int32_t)h->ses->free_blocks.size() : -1;
  if (block_size) *block_size = h->ses->paged() ? (1 << h->ses->bs_shift) : 0;
  return VLG_OK;
}

extern "C" int vlg_gpt_session_begin(vlg_gpt_t* h, int32_t rows, int32_t max_new_tokens, const vlg_sampling_params* sp) {
  VLG_CHECK(h && sp && rows > 0 && max_new_tokens > 0, VLG_ERR_BAD_ARG, "vlg_gpt_session_begin: bad argument");
  if (h->cfg.model_type == VLG_T2V) {   // continuous-latent models (round 4; the reference's serve path stops at c2i): no guidance, as generate_t2v's shipped mode
    VLG_CHECK(sp->cfg_scale <= 1.0f, VLG_ERR_UNSUPPORTED, "sessions of the continuous-latent models run without transformer guidance (cfg_scale 1)");
    VLG_CHECK(h->cfg.head == VLG_HEAD_ADAPTER2 || h->cfg.head == VLG_HEAD_HIDDEN, VLG_ERR_UNSUPPORTED, "t2v sessions: adapter2 or hidden (DiffLoss) head");
    if (h->cfg.head == VLG_HEAD_HIDDEN) {
      VLG_CHECK(h->cfg_iter == 1.0f, VLG_ERR_UNSUPPORTED, "sessions of the DiffLoss head run without its guidance (cfg_iter 1): a pair would span two slots");
      const bool ok = h->dl_persist_on && h->fuse_gemm &&
                      (h->dtype == VLG_BF16 ? dl_persist_ok<bf16>(rows, h->dW, h->C, h->dDepth, h->dl_rows) : dl_persist_ok<float>(rows, h->dW, h->C, h->dDepth, h->dl_rows));
      VLG_CHECK(ok, VLG_ERR_UNSUPPORTED, "sessions with the DiffLoss head need the persistent sampler: %d slots at width %d are outside what it covers", rows, h->dW);
    }
  } else {
    VLG_CHECK(h->cfg.head == VLG_HEAD_LOGITS && (h->cfg.model_type == VLG_C2I || h->cfg.model_type == VLG_T2I), VLG_ERR_UNSUPPORTED,
              "sessions cover the token models (class-conditional, serve/sample_c2i.py, and text-conditioned) and the continuous-latent video models");
  }
  if (sp->cfg_scale > 1.0f && h->cfg.model_type == VLG_T2I)
    VLG_CHECK(h->Tc == 120, VLG_ERR_BAD_SHAPE, "CFG needs cls_token_num == 120 (uncond_embedding is [120, caption_dim], gpt.py:96)");
  VLG_CHECK(h->Tc + max_new_tokens <= h->npos, VLG_ERR_BAD_SHAPE, "max_new_tokens %d exceeds the RoPE table", max_new_tokens);
  for (auto& kv : h->w) VLG_CHECK(kv.second.loaded, VLG_ERR_STATE, "weight %s was never loaded", kv.first.c_str());
  h->ses.reset();
  return h->dtype == VLG_BF16 ? session_begin_impl<bf16>(h, rows, max_new_tokens, *sp) : session_begin_impl<float>(h, rows, max_new_tokens, *sp);
}

extern "C" int vlg_gpt_session_step(vlg_gpt_t* h, const int32_t* h_row_class) {
  VLG_CHECK(h && h_row_class, VLG_ERR_BAD_ARG, "vlg_gpt_session_step: null argument");
  VLG_CHECK(h->ses != nullptr, VLG_ERR_STATE, "no open session");
  return h->dtype == VLG_BF16 ? session_step_impl<bf16>(h, h_row_class) : session_step_impl<float>(h, h_row_class);
}

extern "C" int vlg_gpt_session_read(vlg_gpt_t* h, int32_t row, int32_t n_tokens, int32_t* h_out) {
  VLG_CHECK(h && h_out && h->ses != nullptr, VLG_ERR_BAD_ARG, "vlg_gpt_session_read: bad argument / no open session");
  vlg_gpt::Session& s = *h->ses;
  VLG_CHECK(h->cfg.model_type != VLG_T2V, VLG_ERR_STATE, "vlg_gpt_session_read: the session's model produces latents (vlg_gpt_session_read_latents)");
  VLG_CHECK(row >= 0 && row < s.R && n_tokens >= 0 && n_tokens <= s.maxN, VLG_ERR_BAD_ARG, "vlg_gpt_session_read: row %d / %d tokens out of range",
            row, n_tokens);
  VLG_HIP(hipStreamSynchronize(s.ln.st));
  VLG_HIP(hipMemcpy(h_out, s.out_ids.as<int32_t>() + (size_t)row * s.maxN, (size_t)n_tokens * sizeof(int32_t), hipMemcpyDeviceToHost));
  return VLG_OK;
}

extern "C" int vlg_gpt_session_read_latents(vlg_gpt_t* h, int32_t row, int32_t n_tokens, float* h_out) {
  VLG_CHECK(h && h_out && h->ses != nullptr, VLG_ERR_BAD_ARG, "vlg_gpt_session_read_latents: bad argument / no open session");
  vlg_gpt::Session& s = *h->ses;
  VLG_CHECK(h->cfg.model_type == VLG_T2V && s.out_lat.p != nullptr, VLG_ERR_STATE, "vlg_gpt_session_read_latents: the session's model samples token ids (vlg_gpt_session_read)");
  VLG_CHECK(row >= 0 && row < s.R && n_tokens >= 0 && n_tokens <= s.maxN, VLG_ERR_BAD_ARG, "vlg_gpt_session_read_latents: row %d / %d tokens out of range",
            row, n_tokens);
  VLG_HIP(hipStreamSynchronize(s.ln.st));
  VLG_HIP(hipMemcpy(h_out, s.out_lat.as<float>() + (size_t)row * s.maxN * h->C, (size_t)n_tokens * h->C * sizeof(float), hipMemcpyDeviceToHost));
  return VLG_OK;
}

extern "C" int vlg_gpt_session_end(vlg_gpt_t* h) {
  VLG_CHECK(h, VLG_ERR_BAD_ARG, "vlg_gpt_session_end: null handle");
  if (h->ses) VLG_HIP(hipStreamSynchronize(h->ses->ln.st));
  h->ses.reset();
  return VLG_OK;
}

extern "C" int vlg_gpt_status(vlg_gpt_t* h, int32_t sync) {
  VLG_CHECK(h, VLG_ERR_BAD_ARG, "vlg_gpt_status: null handle");
  if (sync) {
    if (h->lane.st) VLG_HIP(hipStreamSynchronize(h->lane.st));
    if (h->ses && h->ses->ln.st) VLG_HIP(hipStreamSynchronize(h->ses->ln.st));
  }
  return collect_fault(h);
}

extern "C" int vlg_gpt_generate(vlg_gpt_t* h, const void* d_cond, const float* d_emb_mask, int32_t B, int32_t N,
                                const vlg_sampling_params* sp, const float* d_noise, int32_t* d_out_ids, float* d_out_lat,
                                float* d_trace, void* stream) {
  VLG_CHECK(h && d_cond && sp, VLG_ERR_BAD_ARG, "vlg_gpt_generate: null argument");
  VLG_CHECK(B > 0 && N > 0, VLG_ERR_BAD_ARG, "vlg_gpt_generate: B and N must be positive");
  VLG_TRY(collect_fault(h));   // an earlier call's time-out nobody asked about
  if (h->cfg.head == VLG_HEAD_LOGITS)
    VLG_CHECK(d_out_ids, VLG_ERR_BAD_ARG, "vlg_gpt_generate: d_out_ids is required for the logits head");
  else
    VLG_CHECK(d_out_lat, VLG_ERR_BAD_ARG, "vlg_gpt_generate: d_out_lat is required for a latent head");
  if (h->dtype == VLG_BF16)
    return generate_impl<bf16>(h, d_cond, d_emb_mask, B, N, *sp, d_noise, d_out_ids, d_out_lat, d_trace, (hipStream_t)stream);
  return generate_impl<float>(h, d_cond, d_emb_mask, B, N, *sp, d_noise, d_out_ids, d_out_lat, d_trace, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// unit entry points
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct Scratch {
  DevBuf ws, aux, state;
};
Scratch& scratch() {
  static Scratch s;
  return s;
}
}  // namespace

extern "C" int vlg_rmsnorm(const void* d_x, const void* d_w, void* d_out, int32_t rows, int32_t dim, float eps, int32_t dtype,
                           void* stream) {
  VLG_CHECK(d_x && d_w && d_out && rows > 0 && dim > 0, VLG_ERR_BAD_ARG, "vlg_rmsnorm: bad argument");
  hipStream_t st = (hipStream_t)stream;
  // the kernel updates h in place only when slabs are given; with ws == nullptr x is read-only
  if (dtype == VLG_BF16)
    return reduce_residual_rmsnorm<bf16>(nullptr, 0, (bf16*)d_x, (const bf16*)d_w, (bf16*)d_out, rows, dim, eps, st);
  if (dtype == VLG_F32)
    return reduce_residual_rmsnorm<float>(nullptr, 0, (float*)d_x, (const float*)d_w, (float*)d_out, rows, dim, eps, st);
  set_error("vlg_rmsnorm: dtype %d", dtype);
  return VLG_ERR_UNSUPPORTED;
}

extern "C" int vlg_linear(const void* d_x, const void* d_w, void* d_out, int32_t M, int32_t N, int32_t K, int32_t dtype,
                          void* stream) {
  VLG_CHECK(d_x && d_w && d_out && M > 0 && N > 0 && K > 0, VLG_ERR_BAD_ARG, "vlg_linear: bad argument");
  hipStream_t st = (hipStream_t)stream;
  VLG_HIP(hipStreamSynchronize(st));
  Scratch& s = scratch();
  VLG_TRY(s.ws.reserve(gemm_ws_floats(M, N, K, (int)dtype_size(dtype)) * sizeof(float)));
  int sp = 1;
  if (dtype == VLG_BF16) {
    VLG_TRY(gemm_slabs<bf16>((const bf16*)d_x, (const bf16*)d_w, s.ws.as<float>(), M, N, K, &sp, st));
    return reduce_store<bf16>(s.ws.as<float>(), sp, (bf16*)d_out, nullptr, M, N, ACT_NONE, st);
  }
  if (dtype == VLG_F32) {
    VLG_TRY(gemm_slabs<float>((const float*)d_x, (const float*)d_w, s.ws.as<float>(), M, N, K, &sp, st));
    return reduce_store<float>(s.ws.as<float>(), sp, (float*)d_out, nullptr, M, N, ACT_NONE, st);
  }
  set_error("vlg_linear: dtype %d", dtype);
  return VLG_ERR_UNSUPPORTED;
}

extern "C" int vlg_sample(const float* d_logits, int32_t B, int32_t V, int32_t cfg_on, const vlg_sampling_params* sp,
                          const float* d_noise, uint64_t step, int32_t* d_out_idx, float* d_out_probs, void* stream) {
  VLG_CHECK(d_logits && sp && d_out_idx && B > 0 && V > 0, VLG_ERR_BAD_ARG, "vlg_sample: bad argument");
  // unit call: output slot 0 of an N = 1 sequence; `step` only seeds the Philox stream / the cfg_interval rule
  vlg_sampling_params p = *sp;
  if (d_noise == nullptr) p.seed = sp->seed + 0x9E3779B97F4A7C15ull * step;
  return sample_rows(d_logits, B, V, cfg_on != 0, p, d_noise, nullptr, 0, 1, d_out_idx, nullptr, nullptr, d_out_probs,
                     (hipStream_t)stream);
}

extern "C" int vlg_attn_decode(const void* d_q, const void* d_k, const void* d_v, void* d_out, int32_t Bp, int32_t H, int32_t S,
                               int32_t hd, int32_t pos, const float* d_mask, int32_t Bmask, int32_t Tc, int32_t dtype, void* stream) {
  VLG_CHECK(d_q && d_k && d_v && d_out && Bp > 0 && H > 0 && S > 0 && pos >= 0 && pos < S, VLG_ERR_BAD_ARG,
            "vlg_attn_decode: bad argument");
  hipStream_t st = (hipStream_t)stream;
  Scratch& s = scratch();
  if (s.aux.bytes < attn_ws_floats(Bp, H, hd) * sizeof(float) || s.state.bytes < sizeof(StepState)) {
    VLG_HIP(hipStreamSynchronize(st));
    VLG_TRY(s.aux.reserve(attn_ws_floats(Bp, H, hd) * sizeof(float)));
    VLG_TRY(s.state.reserve(sizeof(StepState)));
  }

  VLG_TRY(set_state(s.state.as<StepState>(), pos, 0, st));
  if (dtype == VLG_BF16)
    return attn_rows<bf16>((const bf16*)d_q, (bf16*)d_k, (bf16*)d_v, (bf16*)d_out, s.aux.as<float>(), s.state.as<StepState>(),
                           Bp, 1, H, hd, S, pos, d_mask, Bmask > 0 ? Bmask : 1, Tc, st);
  if (dtype == VLG_F32)
    return attn_rows<float>((const float*)d_q, (float*)d_k, (float*)d_v, (float*)d_out, s.aux.as<float>(),
                            s.state.as<StepState>(), Bp, 1, H, hd, S, pos, d_mask, Bmask > 0 ? Bmask : 1, Tc, st);
  set_error("vlg_attn_decode: dtype %d", dtype);
  return VLG_ERR_UNSUPPORTED;
}
