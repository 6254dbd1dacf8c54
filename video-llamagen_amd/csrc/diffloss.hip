// Elementwise kernels of the DiffLoss per-token diffusion head (SimpleMLPAdaLN + DDPM reverse step).
//
// Reference: autoregressive/models/diffloss.py:55-56 (modulate), :99-148 (ResBlock, FinalLayer), :217-238 (forward);
// diffusion/gaussian_diffusion.py:254-332 (p_mean_variance, LEARNED_RANGE), :334-339, :232-252, :376-420 (p_sample).
// The Linear layers run on the skinny MFMA GEMM of gpt_kernels.hip; this file holds what sits between them.
#include "gpt_kernels.h"

namespace vlg {

__device__ __forceinline__ float silu_d(float x) { return x / (1.0f + expf(-x)); }

// ys[b][w] = rt(silu(rt(temb[w] + cemb[b][w])))      (diffloss.py:229 + the SiLU heading every adaLN_modulation)
template <typename T>
__global__ __launch_bounds__(256) void dl_make_y_kernel(const T* __restrict__ temb, const T* __restrict__ cemb, T* __restrict__ ys, int B, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * W) return;
  const float y = DT<T>::rt(DT<T>::ld(temb + i % W) + DT<T>::ld(cemb + i));
  DT<T>::st(ys + i, silu_d(y));
}
template <typename T>
int dl_make_y(const T* temb, const T* cemb, T* ys, int B, int W, hipStream_t st) {
  dl_make_y_kernel<T><<<cdiv(B * W, 256), 256, 0, st>>>(temb, cemb, ys, B, W);
  return VLG_OK;
}
template int dl_make_y<float>(const float*, const float*, float*, int, int, hipStream_t);
template int dl_make_y<bf16>(const bf16*, const bf16*, bf16*, int, int, hipStream_t);

// all respaced steps at once: ys[(i*B + b)][w] = rt(silu(rt(temb[i][w] + cemb[b][w]))) - the adaLN inputs do not depend on x, so the
// modulation GEMM of a token runs once over S*B rows instead of S times over B rows
template <typename T>
__global__ __launch_bounds__(256) void dl_make_y_all_kernel(const T* __restrict__ temb, const T* __restrict__ cemb, T* __restrict__ ys, int S, int B,
                                                            int W) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)S * B * W) return;
  const int w = (int)(i % W);
  const int b = (int)((i / W) % B);
  const int st = (int)(i / ((long long)W * B));
  const float y = DT<T>::rt(DT<T>::ld(temb + (size_t)st * W + w) + DT<T>::ld(cemb + (size_t)b * W + w));
  DT<T>::st(ys + i, silu_d(y));
}
template <typename T>
int dl_make_y_all(const T* temb, const T* cemb, T* ys, int S, int B, int W, hipStream_t st) {
  dl_make_y_all_kernel<T><<<(unsigned)cdiv64((long long)S * B * W, 256), 256, 0, st>>>(temb, cemb, ys, S, B, W);
  return VLG_OK;
}
template int dl_make_y_all<float>(const float*, const float*, float*, int, int, int, hipStream_t);
template int dl_make_y_all<bf16>(const bf16*, const bf16*, bf16*, int, int, int, hipStream_t);

// g[b] = rt(rt(LayerNorm(h[b]) (* lnw + lnb)) * (1 + scale[b]) + shift[b]);  one workgroup per row, eps 1e-6
template <typename T>
__global__ __launch_bounds__(256) void dl_ln_modulate_kernel(const T* __restrict__ h, const T* __restrict__ lnw, const T* __restrict__ lnb,
                                                             const T* __restrict__ shift, const T* __restrict__ scale, int mod_stride,
                                                             T* __restrict__ g, int W) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const T* hr = h + (size_t)b * W;
  float s = 0.f;
  for (int i = threadIdx.x; i < W; i += 256) s += DT<T>::ld(hr + i);
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)W;
  __syncthreads();
  float q = 0.f;
  for (int i = threadIdx.x; i < W; i += 256) {
    const float d = DT<T>::ld(hr + i) - mu;
    q += d * d;
  }
  for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)W + 1e-6f);
  for (int i = threadIdx.x; i < W; i += 256) {
    float n = (DT<T>::ld(hr + i) - mu) * rstd;
    if (lnw) n = n * DT<T>::ld(lnw + i) + DT<T>::ld(lnb + i);
    n = DT<T>::rt(n);
    const float sc = DT<T>::ld(scale + (size_t)b * mod_stride + i), sh = DT<T>::ld(shift + (size_t)b * mod_stride + i);
    DT<T>::st(g + (size_t)b * W + i, n * (1.0f + sc) + sh);
  }
}
template <typename T>
int dl_ln_modulate(const T* h, const T* lnw, const T* lnb, const T* shift, const T* scale, int mod_stride, T* g, int B, int W, hipStream_t st) {
  dl_ln_modulate_kernel<T><<<B, 256, 0, st>>>(h, lnw, lnb, shift, scale, mod_stride, g, W);
  return VLG_OK;
}
template int dl_ln_modulate<float>(const float*, const float*, const float*, const float*, const float*, int, float*, int, int, hipStream_t);
template int dl_ln_modulate<bf16>(const bf16*, const bf16*, const bf16*, const bf16*, const bf16*, int, bf16*, int, int, hipStream_t);

// h = rt(h + rt(gate * g))          (diffloss.py:128)
template <typename T>
__global__ __launch_bounds__(256) void dl_gated_residual_kernel(T* __restrict__ h, const T* __restrict__ gate, int gate_stride,
                                                                const T* __restrict__ g, int B, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, w = i % W;
  const float v = DT<T>::rt(DT<T>::ld(gate + (size_t)b * gate_stride + w) * DT<T>::ld(g + i));
  DT<T>::st(h + i, DT<T>::ld(h + i) + v);
}
template <typename T>
int dl_gated_residual(T* h, const T* gate, int gate_stride, const T* g, int B, int W, hipStream_t st) {
  dl_gated_residual_kernel<T><<<cdiv(B * W, 256), 256, 0, st>>>(h, gate, gate_stride, g, B, W);
  return VLG_OK;
}
template int dl_gated_residual<float>(float*, const float*, int, const float*, int, int, hipStream_t);
template int dl_gated_residual<bf16>(bf16*, const bf16*, int, const bf16*, int, int, hipStream_t);

__device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint32_t ctr[4] = {a, b, c, d};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1, n3 = (uint32_t)p0;
    ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)(ctr[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(ctr[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);   // Box-Muller
}

// noise layout [N][S+1][B_total][C]; k = 0 is x_T, k >= 1 the draw of the k-th reverse step
template <typename T>
__global__ void dl_init_x_kernel(T* __restrict__ x, const float* __restrict__ noise, const StepState* __restrict__ state, int S, int B, int C,
                                 int b_off, int B_total, uint64_t seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C, step = state->step;
  const float n = noise ? noise[(((size_t)step * (S + 1)) * B_total + b_off + b) * C + c]
                        : philox_normal(seed, (uint32_t)c, (uint32_t)(b_off + b), (uint32_t)step, 0u);
  DT<T>::st(x + i, n);
}
template <typename T>
int dl_init_x(T* x, const float* noise, const StepState* state, int S, int B, int C, int b_off, int B_total, uint64_t seed, hipStream_t st) {
  dl_init_x_kernel<T><<<cdiv(B * C, 256), 256, 0, st>>>(x, noise, state, S, B, C, b_off, B_total, seed);
  return VLG_OK;
}
template int dl_init_x<float>(float*, const float*, const StepState*, int, int, int, int, int, uint64_t, hipStream_t);
template int dl_init_x<bf16>(bf16*, const float*, const StepState*, int, int, int, int, int, uint64_t, hipStream_t);

// one DDPM reverse step (learned-range variance, eps prediction, clip_denoised = False)
template <typename T>
__global__ void dl_ddpm_step_kernel(T* __restrict__ x, const T* __restrict__ out, const float* __restrict__ noise,
                                    const StepState* __restrict__ state, DdpmCoef cf, int k, int S, int B, int C, int b_off, int B_total,
                                    float temperature, uint64_t seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C, step = state->step;
  const float eps = DT<T>::ld(out + (size_t)b * 2 * C + c), v = DT<T>::ld(out + (size_t)b * 2 * C + C + c);
  const float xv = DT<T>::ld(x + i);
  const float frac = (v + 1.0f) / 2.0f;
  const float logvar = frac * cf.max_log + (1.0f - frac) * cf.min_log;
  const float x0 = cf.sqrt_recip * xv - cf.sqrt_recipm1 * eps;
  const float mean = cf.coef1 * x0 + cf.coef2 * xv;
  float r = mean;
  if (cf.nonzero) {
    const float n = noise ? noise[(((size_t)step * (S + 1) + 1 + k) * B_total + b_off + b) * C + c]
                          : philox_normal(seed, (uint32_t)c, (uint32_t)(b_off + b), (uint32_t)step, (uint32_t)(1 + k));
    r = mean + expf(0.5f * logvar) * n * temperature;
  }
  DT<T>::st(x + i, r);
}
template <typename T>
int dl_ddpm_step(T* x, const T* out, const float* noise, const StepState* state, const DdpmCoef& cf, int k, int S, int B, int C, int b_off,
                 int B_total, float temperature, uint64_t seed, hipStream_t st) {
  dl_ddpm_step_kernel<T><<<cdiv(B * C, 256), 256, 0, st>>>(x, out, noise, state, cf, k, S, B, C, b_off, B_total, temperature, seed);
  return VLG_OK;
}
template int dl_ddpm_step<float>(float*, const float*, const float*, const StepState*, const DdpmCoef&, int, int, int, int, int, int, float, uint64_t, hipStream_t);
template int dl_ddpm_step<bf16>(bf16*, const bf16*, const float*, const StepState*, const DdpmCoef&, int, int, int, int, int, int, float, uint64_t, hipStream_t);

// One launch per reverse step: x_out = (k < 0 ? x_T : p_sample(x_in, net_out)), then the next network evaluation's input
// projection hc[b][w] = rt(bias[w] + sum_c x_out[b][c] * Wip[w][c]) (diffloss.py:226, K = C <= 16).  One workgroup per row.
// n_half > 0: guidance inside the sampler (DiffLoss.sample cfg != 1 -> SimpleMLPAdaLN.forward_with_cfg, diffloss.py:37-41,240-248): rows
// [0, n_half) are the conditional half, [n_half, 2 n_half) the unconditional one.  x_T of row b is the draw of row b % n_half; the network
// input of EVERY row is the conditional half's x_t (hc[b] = proj(x[b % n_half])); eps = rt(u + rt(cfg * rt(c - u))) for both rows of a
// pair; variance, draws and the x_t recursion stay per row.
template <typename T, int CMAX>
__global__ __launch_bounds__(256) void dl_step_proj_kernel(const T* __restrict__ x_in, T* __restrict__ x_out, const T* __restrict__ out,
                                                           const float* __restrict__ noise, const StepState* __restrict__ state, DdpmCoef cf, int k,
                                                           int S, int B, int C, int b_off, int B_total, float temperature, uint64_t seed,
                                                           const T* __restrict__ wip, const T* __restrict__ bip, T* __restrict__ hc, int W,
                                                           int n_half, float cfg) {
  __shared__ float xs[CMAX];
  const int b = blockIdx.x, c = threadIdx.x, step = state->step;
  auto reverse_step = [&](int row) -> float {   // x_{t-1}[row][c] (or x_T for k < 0), rounded to T
    const int pr = n_half ? row % n_half : row;
    float r;
    if (k < 0) {
      r = noise ? noise[(((size_t)step * (S + 1)) * B_total + b_off + pr) * C + c]
                : philox_normal(seed, (uint32_t)c, (uint32_t)(b_off + pr), (uint32_t)step, 0u);
    } else {
      float eps = DT<T>::ld(out + (size_t)row * 2 * C + c);
      if (n_half) {
        const float ce = DT<T>::ld(out + (size_t)pr * 2 * C + c), ue = DT<T>::ld(out + (size_t)(pr + n_half) * 2 * C + c);
        eps = DT<T>::rt(ue + DT<T>::rt(cfg * DT<T>::rt(ce - ue)));
      }
      const float v = DT<T>::ld(out + (size_t)row * 2 * C + C + c);
      const float xv = DT<T>::ld(x_in + (size_t)row * C + c);
      const float frac = (v + 1.0f) / 2.0f;
      const float logvar = frac * cf.max_log + (1.0f - frac) * cf.min_log;
      const float x0 = cf.sqrt_recip * xv - cf.sqrt_recipm1 * eps;
      const float mean = cf.coef1 * x0 + cf.coef2 * xv;
      r = mean;
      if (cf.nonzero) {
        const float n = noise ? noise[(((size_t)step * (S + 1) + 1 + k) * B_total + b_off + row) * C + c]
                              : philox_normal(seed, (uint32_t)c, (uint32_t)(b_off + row), (uint32_t)step, (uint32_t)(1 + k));
        r = mean + expf(0.5f * logvar) * n * temperature;
      }
    }
    return DT<T>::rt(r);
  };
  if (c < C) {   // one thread per latent channel draws / steps; the row's values are shared through LDS
    const float r = reverse_step(b);
    DT<T>::st(x_out + (size_t)b * C + c, r);
    xs[c] = (n_half && b >= n_half) ? reverse_step(b - n_half) : r;   // the next evaluation reads the conditional half's x_t
  }
  if (hc == nullptr) return;
  __syncthreads();
  for (int w = threadIdx.x; w < W; w += 256) {
    float acc = 0.f;
#pragma unroll
    for (int cc = 0; cc < CMAX; ++cc)
      if (cc < C) acc = fmaf(xs[cc], DT<T>::ld(wip + (size_t)w * C + cc), acc);
    DT<T>::st(hc + (size_t)b * W + w, acc + DT<T>::ld(bip + w));
  }
}
template <typename T>
int dl_step_proj(const T* x_in, T* x_out, const T* out, const float* noise, const StepState* state, const DdpmCoef& cf, int k, int S, int B, int C,
                 int b_off, int B_total, float temperature, uint64_t seed, const T* wip, const T* bip, T* hc, int W, hipStream_t st, int n_half,
                 float cfg) {
  if (C > 16) {
    set_error("dl_step_proj: vae_embed_dim %d > 16", C);
    return VLG_ERR_UNSUPPORTED;
  }
  dl_step_proj_kernel<T, 16><<<B, 256, 0, st>>>(x_in, x_out, out, noise, state, cf, k, S, B, C, b_off, B_total, temperature, seed,
                                                               wip, bip, hc, W, n_half, cfg);
  return VLG_OK;
}
template int dl_step_proj<float>(const float*, float*, const float*, const float*, const StepState*, const DdpmCoef&, int, int, int, int, int, int,
                                 float, uint64_t, const float*, const float*, float*, int, hipStream_t, int, float);
template int dl_step_proj<bf16>(const bf16*, bf16*, const bf16*, const float*, const StepState*, const DdpmCoef&, int, int, int, int, int, int, float,
                                uint64_t, const bf16*, const bf16*, bf16*, int, hipStream_t, int, float);

// sampled latent T [B,C] -> cur [B,C] fp32 (next step's input), out_lat[b][step], trace
template <typename T>
__global__ void dl_finish_kernel(const T* __restrict__ x, float* __restrict__ cur, float* __restrict__ out_lat, float* __restrict__ trace,
                                 const StepState* __restrict__ state, int B, int C, int N, int b_off, int B_total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C, step = state->step;
  const float v = DT<T>::ld(x + i);
  cur[i] = v;
  out_lat[((size_t)b * N + step) * C + c] = v;
  if (trace) trace[((size_t)step * B_total + b_off + b) * C + c] = v;
}
template <typename T>
int dl_finish(const T* x, float* cur, float* out_lat, float* trace, const StepState* state, int B, int C, int N, int b_off, int B_total,
              hipStream_t st) {
  dl_finish_kernel<T><<<cdiv(B * C, 256), 256, 0, st>>>(x, cur, out_lat, trace, state, B, C, N, b_off, B_total);
  return VLG_OK;
}
template int dl_finish<float>(const float*, float*, float*, float*, const StepState*, int, int, int, int, int, hipStream_t);
template int dl_finish<bf16>(const bf16*, float*, float*, float*, const StepState*, int, int, int, int, int, hipStream_t);

}  // namespace vlg
