// T5 text encoder for the conditioning step in front of the t2i / t2v path (reference: language/t5.py:60-81 wraps
// transformers.T5EncoderModel; flan-t5-xl = d_model 2048, d_kv 64, 32 heads, d_ff 5120, 24 layers, gated GELU).
// Pre-norm residual blocks: h += Wo . attn(T5LayerNorm(h));  h += wo . (gelu_new(wi_0 n) * wi_1 n);  out = T5LayerNorm(h).
// Attention has a bucketed relative position bias shared by all layers (taken from block 0), no 1/sqrt(d) scaling, and an additive
// padding mask.  The GEMMs are the slab kernels of the GPT prefill (gemm_slabs: 64 x 64 LDS-shared tiles at M = B x T rows), the
// residual + norm steps the same reduce_residual_rmsnorm (T5LayerNorm is an RMSNorm without bias, modeling_t5.py).
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "gpt_kernels.h"

using namespace vlg;

struct vlg_t5 {
  vlg_t5_config cfg;
  int dtype;
  size_t esz;
  std::map<std::string, Tensor> w;       // handle-dtype tensors by their transformers state-dict name (+ merged "...qkv", "...wi")
  std::map<std::string, int> parts;      // merged tensors: bit mask of the parts loaded so far
  DevBuf bias_tab;                       // fp32 [num_buckets][H]
  DevBuf x, xn, qkv, att, g, ws, bias;   // activations [M, .], fp32 slabs, per-call bias [H][T][T]
  hipStream_t st = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  ~vlg_t5() {
    if (st) (void)hipStreamDestroy(st);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
  }
};

namespace {

__device__ __forceinline__ float gelu_new_d(float x) {
  const float k = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}

// g[m][n] = rt(rt(gelu_new(rt(sum a))) * rt(sum b)), slab rows are [wi_0 | wi_1] outputs (T5DenseGatedActDense)
template <typename T>
__global__ __launch_bounds__(256) void reduce_gelu_mul_kernel(const float* __restrict__ ws, int splits, T* __restrict__ g, int M, int F) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)M * F) return;
  const int m = (int)(i / F), n = (int)(i % F);
  const size_t N2 = 2 * (size_t)F;
  float a = 0.f, b = 0.f;
  for (int k = 0; k < splits; ++k) {
    const float* row = ws + ((size_t)k * M + m) * N2;
    a += row[n];
    b += row[F + n];
  }
  a = DT<T>::rt(a);
  b = DT<T>::rt(b);
  DT<T>::st(g + i, DT<T>::rt(gelu_new_d(a)) * b);
}

// bias[h][i][j] = rt(table[bucket(j - i)][h])   (modeling_t5.py _relative_position_bucket, bidirectional)
template <typename T>
__global__ void t5_bias_kernel(const float* __restrict__ table, float* __restrict__ bias, int H, int Tn, int num_buckets, int max_distance) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * Tn * Tn) return;
  const int j = idx % Tn, i = (idx / Tn) % Tn, h = idx / (Tn * Tn);
  const int rel = j - i;
  const int nb = num_buckets / 2;
  int ret = rel > 0 ? nb : 0;
  const int n = rel < 0 ? -rel : rel;
  const int max_exact = nb / 2;
  int v;
  if (n < max_exact) {
    v = n;
  } else {
    v = max_exact + (int)(logf((float)n / (float)max_exact) / logf((float)max_distance / (float)max_exact) * (float)(nb - max_exact));
    v = v < nb - 1 ? v : nb - 1;
  }
  bias[idx] = DT<T>::rt(table[(size_t)(ret + v) * H + h]);
}

// one workgroup per (query i, head h, batch b): thread j = key j (T <= 256).  scores rt(rt(q.k) + rt(bias + mask)), fp32 softmax -> rt,
// out[d] = rt(sum_j P[j] v[j][d]).  qkv rows [M][3 * inner] = [q | k | v].
template <typename T>
__global__ __launch_bounds__(256) void t5_attn_kernel(const T* __restrict__ qkv, const float* __restrict__ bias, const float* __restrict__ mask,
                                                      T* __restrict__ out, int Tn, int H, int dk, float neg) {
  extern __shared__ float sm[];   // q[dk], p[Tn], red[8]
  float* qs = sm;
  float* ps = sm + dk;
  float* red = ps + Tn;
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, j = threadIdx.x;
  const int inner = H * dk;
  const size_t rowq = ((size_t)b * Tn + i) * 3 * inner + (size_t)h * dk;
  for (int d = j; d < dk; d += 256) qs[d] = DT<T>::ld(qkv + rowq + d);
  __syncthreads();
  float s = -INFINITY;
  if (j < Tn) {
    const T* kr = qkv + ((size_t)b * Tn + j) * 3 * inner + inner + (size_t)h * dk;
    float dot = 0.f;
    for (int d = 0; d < dk; ++d) dot = fmaf(qs[d], DT<T>::ld(kr + d), dot);
    const float ext = (1.0f - mask[(size_t)b * Tn + j]) * neg;
    const float pb = DT<T>::rt(bias[((size_t)h * Tn + i) * Tn + j] + ext);
    s = DT<T>::rt(DT<T>::rt(dot) + pb);
  }
  float mx = s;
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((j & 63) == 0) red[j >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float e = j < Tn ? expf(s - mx) : 0.f;
  float sum = e;
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  if ((j & 63) == 0) red[j >> 6] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  if (j < Tn) ps[j] = DT<T>::rt(e / sum);
  __syncthreads();
  for (int d = j; d < dk; d += 256) {
    float acc = 0.f;
    for (int jj = 0; jj < Tn; ++jj) acc = fmaf(ps[jj], DT<T>::ld(qkv + ((size_t)b * Tn + jj) * 3 * inner + 2 * inner + (size_t)h * dk + d), acc);
    DT<T>::st(out + ((size_t)b * Tn + i) * inner + (size_t)h * dk + d, acc);
  }
}

template <typename T>
__global__ void t5_out_kernel(const T* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = DT<T>::ld(x + i);
}

template <typename T>
const T* W(vlg_t5* h, const std::string& n) {
  return h->w.at(n).buf.as<T>();
}
// fragment-major copy of a Linear weight (gpt_kernels.h: relayout_fragment_major), built on first use after a load; null where the shape
// does not tile
template <typename T>
const T* Wfm(vlg_t5* h, const std::string& n, hipStream_t st) {
  Tensor& t = h->w.at(n);
  if (t.shape.size() != 2 || !fragment_major_ok((int)t.shape[0], (int)t.shape[1], (int)sizeof(T))) return nullptr;
  if (t.fm_stale) {
    if (t.fm.reserve((size_t)t.shape[0] * t.shape[1] * sizeof(T)) != VLG_OK) return nullptr;
    if (relayout_fragment_major<T>(t.buf.as<T>(), t.fm.as<T>(), (int)t.shape[0], (int)t.shape[1], st) != VLG_OK) return nullptr;
    t.fm_stale = false;
  }
  return t.fm.as<T>();
}

template <typename T>
int encode_impl(vlg_t5* h, const int64_t* d_ids, const float* d_mask, int B, int Tn, float* d_out, hipStream_t caller) {
  const vlg_t5_config& c = h->cfg;
  const int D = c.d_model, H = c.num_heads, dk = c.d_kv, inner = H * dk, F = c.d_ff, M = B * Tn;
  const size_t e = h->esz;
  for (auto& kv : h->w) VLG_CHECK(kv.second.loaded, VLG_ERR_STATE, "weight %s was never loaded", kv.first.c_str());
  VLG_TRY(h->x.reserve((size_t)M * D * e));
  VLG_TRY(h->xn.reserve((size_t)M * D * e));
  VLG_TRY(h->qkv.reserve((size_t)M * 3 * inner * e));
  VLG_TRY(h->att.reserve((size_t)M * inner * e));
  VLG_TRY(h->g.reserve((size_t)M * F * e));
  VLG_TRY(h->bias.reserve((size_t)H * Tn * Tn * sizeof(float)));
  size_t wsf = 0;
  for (auto nk : {std::pair<int, int>{3 * inner, D}, {D, inner}, {2 * F, D}, {D, F}}) wsf = std::max(wsf, gemm_ws_floats(M, nk.first, nk.second, (int)e));
  VLG_TRY(h->ws.reserve(wsf * sizeof(float)));
  hipStream_t st = h->st;
  VLG_HIP(hipEventRecord(h->ev_in, caller));
  VLG_HIP(hipStreamWaitEvent(st, h->ev_in, 0));
  T* x = h->x.as<T>();
  T* xn = h->xn.as<T>();
  float* ws = h->ws.as<float>();
  const float eps = c.layer_norm_epsilon;
  const float neg = sizeof(T) == 2 ? -3.3895313892515355e38f : -3.4028234663852886e38f;   // torch.finfo(dtype).min
  VLG_TRY(gather_rows_i64<T>(W<T>(h, "shared.weight"), d_ids, M, 0, x, M, D, c.vocab_size, st));
  t5_bias_kernel<T><<<cdiv(H * Tn * Tn, 256), 256, 0, st>>>(h->bias_tab.as<float>(), h->bias.as<float>(), H, Tn, c.relative_attention_num_buckets,
                                                            c.relative_attention_max_distance);
  VLG_TRY(reduce_residual_rmsnorm<T>(nullptr, 0, x, W<T>(h, "encoder.block.0.layer.0.layer_norm.weight"), xn, M, D, eps, st));
  for (int l = 0; l < c.num_layers; ++l) {
    const std::string p = "encoder.block." + std::to_string(l) + ".layer.";
    int sp = 1;
    VLG_TRY(gemm_slabs<T>(xn, W<T>(h, p + "0.SelfAttention.qkv"), ws, M, 3 * inner, D, &sp, st, Wfm<T>(h, p + "0.SelfAttention.qkv", st)));
    VLG_TRY(reduce_store<T>(ws, sp, h->qkv.as<T>(), nullptr, M, 3 * inner, ACT_NONE, st));
    t5_attn_kernel<T><<<dim3(Tn, H, B), 256, (size_t)(dk + Tn + 8) * sizeof(float), st>>>(h->qkv.as<T>(), h->bias.as<float>(), d_mask, h->att.as<T>(), Tn,
                                                                                         H, dk, neg);
    VLG_TRY(gemm_slabs<T>(h->att.as<T>(), W<T>(h, p + "0.SelfAttention.o.weight"), ws, M, D, inner, &sp, st, Wfm<T>(h, p + "0.SelfAttention.o.weight", st)));
    VLG_TRY(reduce_residual_rmsnorm<T>(ws, sp, x, W<T>(h, p + "1.layer_norm.weight"), xn, M, D, eps, st));
    VLG_TRY(gemm_slabs<T>(xn, W<T>(h, p + "1.DenseReluDense.wi"), ws, M, 2 * F, D, &sp, st, Wfm<T>(h, p + "1.DenseReluDense.wi", st)));
    reduce_gelu_mul_kernel<T><<<(unsigned)cdiv64((long long)M * F, 256), 256, 0, st>>>(ws, sp, h->g.as<T>(), M, F);
    VLG_TRY(gemm_slabs<T>(h->g.as<T>(), W<T>(h, p + "1.DenseReluDense.wo.weight"), ws, M, D, F, &sp, st, Wfm<T>(h, p + "1.DenseReluDense.wo.weight", st)));
    const std::string nxt = l + 1 < c.num_layers ? "encoder.block." + std::to_string(l + 1) + ".layer.0.layer_norm.weight"
                                                 : std::string("encoder.final_layer_norm.weight");
    VLG_TRY(reduce_residual_rmsnorm<T>(ws, sp, x, W<T>(h, nxt), xn, M, D, eps, st));
  }
  t5_out_kernel<T><<<(unsigned)cdiv64((long long)M * D, 256), 256, 0, st>>>(xn, d_out, (long long)M * D);
  VLG_HIP(hipEventRecord(h->ev_out, st));
  VLG_HIP(hipStreamWaitEvent(caller, h->ev_out, 0));
  return VLG_OK;
}

}  // namespace

extern "C" int vlg_t5_create(const vlg_t5_config* cfg, vlg_t5_t** out) {
  VLG_CHECK(cfg && out, VLG_ERR_BAD_ARG, "vlg_t5_create: null argument");
  VLG_CHECK(cfg->dtype == VLG_F32 || cfg->dtype == VLG_BF16, VLG_ERR_UNSUPPORTED, "vlg_t5_create: dtype %d", cfg->dtype);
  VLG_CHECK(cfg->d_model > 0 && cfg->d_kv > 0 && cfg->num_heads > 0 && cfg->d_ff > 0 && cfg->num_layers > 0 && cfg->vocab_size > 0 &&
                cfg->relative_attention_num_buckets >= 4 && cfg->relative_attention_max_distance > 0,
            VLG_ERR_BAD_ARG, "vlg_t5_create: bad configuration");
  VLG_CHECK(cfg->gated_gelu == 1, VLG_ERR_UNSUPPORTED, "only the gated-GELU feed-forward of flan-t5 / t5-v1_1 is built (language/t5.py:16)");
  auto h = std::make_unique<vlg_t5>();
  h->cfg = *cfg;
  h->dtype = cfg->dtype;
  h->esz = dtype_size(cfg->dtype);
  const int D = cfg->d_model, inner = cfg->num_heads * cfg->d_kv, F = cfg->d_ff;
  auto add = [&](const std::string& n, std::vector<int64_t> shape) {
    Tensor& t = h->w[n];
    t.shape = shape;
    return t.buf.reserve((size_t)t.numel() * h->esz);
  };
  VLG_TRY(add("shared.weight", {cfg->vocab_size, D}));
  VLG_TRY(add("encoder.final_layer_norm.weight", {D}));
  for (int l = 0; l < cfg->num_layers; ++l) {
    const std::string p = "encoder.block." + std::to_string(l) + ".layer.";
    VLG_TRY(add(p + "0.SelfAttention.qkv", {3 * inner, D}));          // q, k, v rows merged: one GEMM
    VLG_TRY(add(p + "0.SelfAttention.o.weight", {D, inner}));
    VLG_TRY(add(p + "0.layer_norm.weight", {D}));
    VLG_TRY(add(p + "1.DenseReluDense.wi", {2 * F, D}));              // wi_0, wi_1 merged
    VLG_TRY(add(p + "1.DenseReluDense.wo.weight", {D, F}));
    VLG_TRY(add(p + "1.layer_norm.weight", {D}));
  }
  VLG_TRY(h->bias_tab.reserve((size_t)cfg->relative_attention_num_buckets * cfg->num_heads * sizeof(float)));
  VLG_HIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  VLG_HIP(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
  VLG_HIP(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
  *out = h.release();
  return VLG_OK;
}

extern "C" int vlg_t5_destroy(vlg_t5_t* h) {
  delete h;
  return VLG_OK;
}

extern "C" int vlg_t5_load_tensor(vlg_t5_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t src_dtype,
                                  int32_t src_on_device, int32_t* consumed) {
  VLG_CHECK(h && name && data && shape, VLG_ERR_BAD_ARG, "vlg_t5_load_tensor: null argument");
  if (consumed) *consumed = 0;
  VLG_CHECK(src_dtype == VLG_F32 || src_dtype == VLG_BF16, VLG_ERR_BAD_ARG, "bad src dtype");
  std::string n(name);
  int64_t numel = 1;
  for (int i = 0; i < ndim; ++i) numel *= shape[i];
  const vlg_t5_config& c = h->cfg;
  const int64_t D = c.d_model, inner = (int64_t)c.num_heads * c.d_kv, F = c.d_ff;
  if (n == "encoder.embed_tokens.weight") return VLG_OK;   // tied to shared.weight
  if (n == "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight") {
    VLG_CHECK(ndim == 2 && shape[0] == c.relative_attention_num_buckets && shape[1] == c.num_heads, VLG_ERR_BAD_SHAPE, "size mismatch for %s", name);
    VLG_TRY(upload_convert(h->bias_tab.p, VLG_F32, data, src_dtype, src_on_device, numel, h->st));
    if (consumed) *consumed = 1;
    return VLG_OK;
  }
  std::string target = n;
  int64_t row_off = 0, rows = ndim > 0 ? shape[0] : 1;
  int part = 0, nparts = 1;
  auto ends = [&](const char* suf) {
    const size_t L = strlen(suf);
    return n.size() >= L && n.compare(n.size() - L, L, suf) == 0;
  };
  for (auto qn : {std::pair<const char*, int>{".SelfAttention.q.weight", 0}, {".SelfAttention.k.weight", 1}, {".SelfAttention.v.weight", 2}})
    if (ends(qn.first)) {
      target = n.substr(0, n.size() - strlen("q.weight")) + "qkv";
      row_off = qn.second * inner;
      part = qn.second;
      nparts = 3;
    }
  for (auto wn : {std::pair<const char*, int>{".DenseReluDense.wi_0.weight", 0}, {".DenseReluDense.wi_1.weight", 1}})
    if (ends(wn.first)) {
      target = n.substr(0, n.size() - strlen("wi_0.weight")) + "wi";
      row_off = wn.second * F;
      part = wn.second;
      nparts = 2;
    }
  auto it = h->w.find(target);
  if (it == h->w.end()) return VLG_OK;   // strict=False
  Tensor& t = it->second;
  const int64_t cols = t.shape.size() > 1 ? t.shape[1] : 1;
  const bool merged = target != n;
  bool ok = merged ? (ndim == 2 && shape[1] == cols && rows == (nparts == 3 ? inner : F)) : ((int)t.shape.size() == ndim);
  if (!merged)
    for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == t.shape[i];
  VLG_CHECK(ok, VLG_ERR_BAD_SHAPE, "size mismatch for %s", name);
  (void)D;
  VLG_TRY(upload_convert((char*)t.buf.p + (size_t)row_off * cols * h->esz, h->dtype, data, src_dtype, src_on_device, numel, h->st));
  int& got = h->parts[target];
  got |= 1 << part;
  t.loaded = got == (1 << nparts) - 1;   // a merged tensor is complete once q, k, v (wi_0, wi_1) all came in
  t.fm_stale = true;                     // the fragment-major copy is rebuilt at the next encode
  if (consumed) *consumed = 1;
  return VLG_OK;
}

extern "C" int vlg_t5_encode(vlg_t5_t* h, const int64_t* d_input_ids, const float* d_attention_mask, int32_t B, int32_t T, float* d_out, void* stream) {
  VLG_CHECK(h && d_input_ids && d_attention_mask && d_out, VLG_ERR_BAD_ARG, "vlg_t5_encode: null argument");
  VLG_CHECK(B > 0 && T > 0 && T <= 256, VLG_ERR_BAD_SHAPE, "vlg_t5_encode: B %d, T %d (T <= 256)", B, T);
  if (h->dtype == VLG_BF16) return encode_impl<bf16>(h, d_input_ids, d_attention_mask, B, T, d_out, (hipStream_t)stream);
  return encode_impl<float>(h, d_input_ids, d_attention_mask, B, T, d_out, (hipStream_t)stream);
}
