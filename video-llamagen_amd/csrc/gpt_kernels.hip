// Hand-written gfx950 kernels for the KV-cached decode / prefill step of the Llama-style decoder.
//
// Reference behaviour being reproduced (paths relative to the reference root):
//   autoregressive/models/gpt.py:137-148 (RMSNorm), :151-167 (SwiGLU), :170-185 (KVCache),
//   :188-242 (Attention), :423-433 (RoPE), autoregressive/models/generate.py:156-165 (mask fix-up).
//
// Design (MI355X): one decode step is HBM-bound (weights once + KV rows 0..p once), so
//   * GEMMs are skinny (M = B' <= 64 rows): weight rows are streamed 64 B per lane straight into MFMA
//     B fragments (k index permuted identically for A and B so every lane reads contiguous bytes),
//     K is split across the 4 waves of a workgroup (LDS reduce) and optionally across workgroups
//     (fp32 slabs, reduced for free in the consumer kernel's prologue - no atomics, deterministic);
//   * attention is split-KV flash-decoding on the VALU (MHA, one query row per (b,h): no MFMA reuse to
//     be had), 16 B per lane coalesced K/V rows, per-lane-group online softmax, DPP reductions;
//   * every elementwise op (residual, RMSNorm, RoPE, KV scatter, SiLU*mul, GELU) lives in a slab-reduce
//     epilogue kernel, rounding to the storage dtype exactly where the reference materialises a tensor.
#include <stdlib.h>

#include "gpt_kernels.h"

namespace vlg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <typename T, int VEC>
struct alignas((VEC * sizeof(T)) > 16 ? 16 : (VEC * sizeof(T))) Pack {
  T v[VEC];
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
// streamed-once data (KV rows, weights): non-temporal load so the stream does not evict what other kernels re-read
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_stream(const T* p) {
  if constexpr (sizeof(Pack<T, VEC>) == 16) {
    return __builtin_bit_cast(Pack<T, VEC>, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)));
  } else if constexpr (sizeof(Pack<T, VEC>) == 8) {
    return __builtin_bit_cast(Pack<T, VEC>, __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p)));
  } else {
    return *reinterpret_cast<const Pack<T, VEC>*>(p);
  }
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// all-reduce (sum) over aligned groups of N consecutive lanes, N power of two <= 64
template <int N>
__device__ __forceinline__ float group_sum(float s) {
  if (N >= 2) s += dpp_f<0xB1>(s);    // quad_perm [1,0,3,2]
  if (N >= 4) s += dpp_f<0x4E>(s);    // quad_perm [2,3,0,1]
  if (N >= 8) s += dpp_f<0x141>(s);   // row_half_mirror
  if (N >= 16) s += dpp_f<0x140>(s);  // row_mirror
  if (N >= 32) s += __shfl_xor(s, 16);
  if (N >= 64) s += __shfl_xor(s, 32);
  return s;
}
__device__ __forceinline__ float wave_sum(float s) { return group_sum<64>(s); }

// block-wide sum for 256-thread blocks; `red` is 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float s, float* red) {
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

template <int TPB>   // `red`: TPB / 64 floats of LDS; the wave sums are added in wave order
__device__ __forceinline__ float block_sum_n(float s, float* red) {
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < TPB / 64; ++i) t += red[i];
  __syncthreads();
  return t;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM on MFMA:   slabs[split][m][n] = sum_{k in slice} x[m][k] * w[n][k]
// workgroup = 4 waves, one 16-column n-tile, MT m-tiles of 16 rows, K blocks dealt round-robin to the
// waves (and to gridDim.z splits).  Per K block every lane reads 64 contiguous bytes of "its" row
// (row = lane&15, quarter = lane>>4): 16 rows x 256 B contiguous per row.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct GemmT;
template <>
struct GemmT<bf16> {
  static constexpr int KBLK = 128;  // elements per K block (256 B per row)
};
template <>
struct GemmT<float> {
  static constexpr int KBLK = 64;
};

template <typename T, int MT>
__device__ __forceinline__ void mfma_block(const u32x4_t (&a)[MT][4], const u32x4_t (&b)[4], f32x4_t (&acc)[MT]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8_t vb = __builtin_bit_cast(bf16x8_t, b[s]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[mt][s]), vb, acc[mt], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* fb = reinterpret_cast<const float*>(&b[s]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float* fa = reinterpret_cast<const float*>(&a[mt][s]);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[e], acc[mt], 0, 0, 0);
        }
      }
    }
  }
}

// DUAL = false: plain GEMM, fp32 partial slabs out.
// DUAL = true : the n-tile is taken from BOTH halves of a [2F, K] weight (rows n and F + n: w1 and w3 of SwiGLU, gpt.py:161-167);
//               no K split across workgroups, and the epilogue writes g = rt(rt(silu(rt(a))) * rt(b)) directly - no slab
//               round trip and no separate SiLU*mul kernel.
template <typename T, int MT, bool DUAL>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                        float* __restrict__ slabs, T* __restrict__ gout, int M, int N, int K) {
  constexpr int KBLK = GemmT<T>::KBLK;
  constexpr int NH = DUAL ? 2 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (MT * 16);
  const int split = blockIdx.z, splits = gridDim.z;
  const int nkb = K / KBLK;

  f32x4_t acc[NH][MT];
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[hf][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // natural k order: load instruction s of a K block reads, for each of the 16 rows, 64 contiguous bytes
  // (4 lanes x 16 B): lane (r, q) gets elements [s*KBLK/4 + q*KBLK/16, +KBLK/16) - identical for A and B operands.
  const T* wrow[NH];
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) wrow[hf] = w + (size_t)(n0 + r + (DUAL ? hf * N : 0)) * K;
  const T* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = m0 + mt * 16 + r;
    row = row < M ? row : M - 1;
    xrow[mt] = x + (size_t)row * K;
  }

  // Every K block this wave owns is requested before the first MFMA: a decode GEMM is ~40-70 KB of weights per
  // CU in total, so the whole kernel is ONE round trip to HBM when all loads are in flight at once.
  constexpr int NB = (MT * NH >= 4) ? 2 : 4;
  u32x4_t a[NB][MT][4], b[NB][NH][4];
  const int stride = 4 * splits;
  for (int kb0 = split * 4 + wave; kb0 < nkb; kb0 += NB * stride) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kb = kb0 + i * stride;
      if (kb < nkb) {
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
          const u32x4_t* pw = reinterpret_cast<const u32x4_t*>(wrow[hf] + (size_t)kb * KBLK) + q;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) b[i][hf][s2] = __builtin_nontemporal_load(pw + s2 * 4);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const u32x4_t* px = reinterpret_cast<const u32x4_t*>(xrow[mt] + (size_t)kb * KBLK) + q;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) a[i][mt][s2] = px[s2 * 4];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (kb0 + i * stride < nkb) {
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) mfma_block<T, MT>(a[i], b[i][hf], acc[hf]);
      }
    }
  }

  // cross-wave reduction through LDS
  __shared__ float red[4][NH][MT][256];
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][hf][mt][e * 64 + lane] = acc[hf][mt][e];
  __syncthreads();
  const int t = threadIdx.x;
  const int e = t >> 6, l2 = t & 63;
  const int col = n0 + (l2 & 15);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = m0 + mt * 16 + (l2 >> 4) * 4 + e;
    if (row < M) {
      const float s0 = red[0][0][mt][t] + red[1][0][mt][t] + red[2][0][mt][t] + red[3][0][mt][t];
      if constexpr (DUAL) {
        const float s1 = red[0][1][mt][t] + red[1][1][mt][t] + red[2][1][mt][t] + red[3][1][mt][t];
        const float av = DT<T>::rt(s0), bv = DT<T>::rt(s1);
        DT<T>::st(gout + (size_t)row * N + col, DT<T>::rt(silu_f(av)) * bv);
      } else {
        slabs[((size_t)split * M + row) * N + col] = s0;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 64-row x 64-column tile for M > 32 (prefill: M = B' x 120 rows; decode at B' = 64; widths the fused decode GEMM does not
// cover, GPT-3B).  gemm_mfma_kernel gives every 16-column workgroup its own copy of the activation rows: at 64 rows that is
// 4 bytes of activations per byte of weights through each CU's vector-memory pipe (GPT-3B decode streamed weights at 1.7 TB/s).
// Here the workgroup's 4 waves own 16 columns each and SHARE the 64 x KBLK activation tile through LDS (double buffer,
// register-staged, 16-byte-chunk XOR swizzle); each wave streams only its own weight rows (non-temporal): 1 : 1.
// Output: fp32 slabs like gemm_mfma_kernel (gridDim.z K-slices), consumed by the reduce_* kernels.
// ------------------------------------------------------------------------------------------------
template <typename T, bool NT, bool FM = false>   // NT: non-temporal weight loads (one row block: every weight byte is read once); FM: fragment-major weights
__global__ __launch_bounds__(256) void gemm_wide_kernel(const T* __restrict__ x, const T* __restrict__ w, float* __restrict__ slabs, int M,
                                                        int N, int K) {
  constexpr int KBLK = GemmT<T>::KBLK;   // 256 bytes per row per K block for both dtypes
  __shared__ u32x4_t As[2][64 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 64 + wave * 16, m0 = blockIdx.y * 64;
  const int split = blockIdx.z, splits = gridDim.z;
  const int nkb = K / KBLK;

  // activation loader: 1024 chunks per tile, 4 per thread
  const u32x4_t* asrc[4];
  int aslot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int row = c >> 4, ch = c & 15;
    int gr = m0 + row;
    gr = gr < M ? gr : M - 1;
    asrc[i] = reinterpret_cast<const u32x4_t*>(x + (size_t)gr * K) + ch;
    aslot[i] = row * 16 + (ch ^ (row & 15));
  }
  constexpr int CPB = KBLK * (int)sizeof(T) / 16;   // 16-byte chunks per row per K block = 16
  // weight chunk s2 of K block kb: row-major = this lane's row, chunk kb * 16 + 4 s2 + q; fragment-major (relayout_fragment_major) = block
  // (n-tile, K step 4 kb + s2), slot `lane`: whole-line requests
  constexpr int WSTEP = FM ? 64 : 4, WBLK = FM ? 256 : CPB;
  const u32x4_t* wsrc = FM ? reinterpret_cast<const u32x4_t*>(w) + (size_t)(n0 / 16) * ((size_t)nkb * 256) + lane
                           : reinterpret_cast<const u32x4_t*>(w + (size_t)(n0 + r) * K) + q;

  f32x4_t acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u32x4_t ra[4], b[4], bn[4];
  int kb = split;
  if (kb < nkb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = asrc[i][(size_t)kb * CPB];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      if constexpr (NT)
        b[s2] = __builtin_nontemporal_load(wsrc + (size_t)kb * WBLK + s2 * WSTEP);
      else
        b[s2] = wsrc[(size_t)kb * WBLK + s2 * WSTEP];   // several row blocks (prefill) re-read the tile: let L2 keep it
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) As[0][aslot[i]] = ra[i];
  }
  __syncthreads();
  int buf = 0;
  for (; kb < nkb; kb += splits) {
    const int kn = kb + splits;
    const bool more = kn < nkb;
    const int kl = more ? kn : kb;   // unconditional loads (the last iteration re-reads its own block): counted waits stay exact
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = asrc[i][(size_t)kl * CPB];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      if constexpr (NT)
        bn[s2] = __builtin_nontemporal_load(wsrc + (size_t)kl * WBLK + s2 * WSTEP);
      else
        bn[s2] = wsrc[(size_t)kl * WBLK + s2 * WSTEP];
    }
    u32x4_t af[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = mt * 16 + r;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) af[mt][s2] = As[buf][row * 16 + ((s2 * 4 + q) ^ (row & 15))];
    }
    mfma_block<T, 4>(af, b, acc);
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) As[buf ^ 1][aslot[i]] = ra[i];
    }
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) b[s2] = bn[s2];
    buf ^= 1;
  }
  // Slab store through LDS (the activation buffers are free after the loop's last barrier): the accumulator layout would write 64-byte
  // pieces (4 rows x 16 columns per wave instruction); the workgroup's 64 x 64 tile goes out as whole 256-byte rows instead - thread t
  // writes 16 bytes, 16 threads cover a row.
  float* cs = reinterpret_cast<float*>(&As[0][0]);   // [64 rows][64 columns], row pitch 68 floats: 17 KB of the 32 KB
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[(mt * 16 + q * 4 + e) * 68 + wave * 16 + r] = acc[mt][e];
  __syncthreads();
  {
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 4;
    const int ncol0 = blockIdx.x * 64;
#pragma unroll
    for (int r0 = 0; r0 < 64; r0 += 16) {
      const int row = m0 + r0 + tr;
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(cs + (r0 + tr) * 68 + tc);
      if (row < M) *reinterpret_cast<f32x4_t*>(slabs + ((size_t)split * M + row) * N + ncol0 + tc) = v;
    }
  }
}

// fallback for shapes the MFMA kernel does not tile (K % KBLK != 0 or N % 16 != 0: adapters with
// vae_embed_dim = 8, toy widths): one wave per output element, lanes stride over K.
template <typename T>
__global__ __launch_bounds__(256) void gemm_naive_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                         float* __restrict__ slabs, int M, int N, int K) {
  const int lane = threadIdx.x & 63;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)M * N;
  if (wid >= total) return;
  const int m = (int)(wid / N), n = (int)(wid % N);
  const T* xr = x + (size_t)m * K;
  const T* wr = w + (size_t)n * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(DT<T>::ld(xr + k), DT<T>::ld(wr + k), s);
  s = wave_sum(s);
  if (lane == 0) slabs[(size_t)m * N + n] = s;
}

int gemm_max_splits() { return 8; }

// launch geometry shared by gemm_slabs and the workspace sizing
static void gemm_plan(int M, int N, int K, int kblk, bool& naive, int& mt, int& mchunks, int& splits, bool* wide = nullptr) {
  naive = (K % kblk != 0) || (N % 16 != 0);
  mt = M > 32 ? 4 : (M > 16 ? 2 : 1);
  mchunks = cdiv(M, mt * 16);
  splits = 1;
  if (wide) *wide = false;
  if (naive) return;
  const int nkb = K / kblk;
  static const bool wide_off = getenv("VLG_GEMM_WIDE") != nullptr && atoi(getenv("VLG_GEMM_WIDE")) == 0;   // A/B knob
  if (M > 32 && N % 64 == 0 && !wide_off) {   // gemm_wide_kernel: 64 x 64 tiles, K slices so that ~3 workgroups per CU stream
    if (wide) *wide = true;
    // (measured on config 5 with 512 / 768 / 1024 / 1536 workgroups: 2.165 / 2.148 / 2.174 / 2.217 s)
    splits = 768 / ((N / 64) * mchunks);
    if (splits > nkb) splits = nkb;
    if (splits > gemm_max_splits()) splits = gemm_max_splits();
    if (splits < 1) splits = 1;
    return;
  }
  const int ntiles = N / 16;
  splits = 1024 / (ntiles * mchunks);
  if (splits > nkb / 4) splits = nkb / 4;
  if (splits > gemm_max_splits()) splits = gemm_max_splits();
  if (splits < 1) splits = 1;
}

size_t gemm_ws_floats(int M, int N, int K, int elem_size) {
  bool naive;
  int mt, mchunks, splits;
  gemm_plan(M, N, K, elem_size == 2 ? 128 : 64, naive, mt, mchunks, splits);
  return (size_t)splits * M * N;
}

template <typename T>
int gemm_slabs(const T* x, const T* w, float* ws, int M, int N, int K, int* splits_out, hipStream_t st, const T* wfm) {
  constexpr int KBLK = GemmT<T>::KBLK;
  if (M <= 0 || N <= 0 || K <= 0) {
    set_error("gemm: bad shape %d %d %d", M, N, K);
    return VLG_ERR_BAD_SHAPE;
  }
  bool naive, wide;
  int mt, mchunks, splits;
  gemm_plan(M, N, K, KBLK, naive, mt, mchunks, splits, &wide);
  if (naive) {
    const long long total = (long long)M * N;
    gemm_naive_kernel<T><<<dim3((unsigned)((total + 3) / 4)), 256, 0, st>>>(x, w, ws, M, N, K);
    *splits_out = 1;
    return VLG_OK;
  }
  if (wide) {
    static const bool fm_off_w = getenv("VLG_GEMM_FM") != nullptr && atoi(getenv("VLG_GEMM_FM")) == 0;
    if (wfm != nullptr && !fm_off_w) {
      if (mchunks == 1)
        gemm_wide_kernel<T, true, true><<<dim3(N / 64, mchunks, splits), 256, 0, st>>>(x, wfm, ws, M, N, K);
      else
        gemm_wide_kernel<T, false, true><<<dim3(N / 64, mchunks, splits), 256, 0, st>>>(x, wfm, ws, M, N, K);
    } else if (mchunks == 1)
      gemm_wide_kernel<T, true><<<dim3(N / 64, mchunks, splits), 256, 0, st>>>(x, w, ws, M, N, K);
    else
      gemm_wide_kernel<T, false><<<dim3(N / 64, mchunks, splits), 256, 0, st>>>(x, w, ws, M, N, K);
    *splits_out = splits;
    return VLG_OK;
  }
  dim3 grid(N / 16, mchunks, splits);
  if (mt == 4)
    gemm_mfma_kernel<T, 4, false><<<grid, 256, 0, st>>>(x, w, ws, nullptr, M, N, K);
  else if (mt == 2)
    gemm_mfma_kernel<T, 2, false><<<grid, 256, 0, st>>>(x, w, ws, nullptr, M, N, K);
  else
    gemm_mfma_kernel<T, 1, false><<<grid, 256, 0, st>>>(x, w, ws, nullptr, M, N, K);
  *splits_out = splits;
  return VLG_OK;
}

// g[M, F] = rt(rt(silu(rt(x @ w1^T))) * rt(x @ w3^T)) with w13 = [w1; w3] ([2F, K]).  Returns false if the shape needs the
// generic path (gemm_slabs + reduce_silu_mul).
template <typename T>
bool gemm_swiglu(const T* x, const T* w13, T* g, int M, int F, int K, hipStream_t st) {
  constexpr int KBLK = GemmT<T>::KBLK;
  if (K % KBLK != 0 || F % 16 != 0) return false;
  static const bool wide_off = getenv("VLG_GEMM_WIDE") != nullptr && atoi(getenv("VLG_GEMM_WIDE")) == 0;
  if (M > 32 && F % 32 == 0 && !wide_off) return false;   // 64-row tiles: gemm_wide_kernel + reduce_silu_mul stream the weights faster
  const int mt = M > 32 ? 4 : (M > 16 ? 2 : 1);
  dim3 grid(F / 16, cdiv(M, mt * 16), 1);
  if (mt == 4)
    gemm_mfma_kernel<T, 4, true><<<grid, 256, 0, st>>>(x, w13, nullptr, g, M, F, K);
  else if (mt == 2)
    gemm_mfma_kernel<T, 2, true><<<grid, 256, 0, st>>>(x, w13, nullptr, g, M, F, K);
  else
    gemm_mfma_kernel<T, 1, true><<<grid, 256, 0, st>>>(x, w13, nullptr, g, M, F, K);
  return true;
}
template bool gemm_swiglu<float>(const float*, const float*, float*, int, int, int, hipStream_t);
template bool gemm_swiglu<bf16>(const bf16*, const bf16*, bf16*, int, int, int, hipStream_t);
template int gemm_slabs<float>(const float*, const float*, float*, int, int, int, int*, hipStream_t, const float*);
template int gemm_slabs<bf16>(const bf16*, const bf16*, float*, int, int, int, int*, hipStream_t, const bf16*);

// out block (t, s), 16-byte slot l = r + 16 q  <-  row 16 t + r, bytes [64 s + 16 q, + 16)
__global__ __launch_bounds__(256) void relayout_fm_kernel(const u32x4_t* __restrict__ w, u32x4_t* __restrict__ out, long long chunks, int nks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= chunks) return;
  const int l = (int)(i & 63);
  const long long blk = i >> 6;
  const long long t = blk / nks;
  const int s = (int)(blk - t * nks);
  const int r = l & 15, q = l >> 4;
  out[i] = w[(t * 16 + r) * (long long)(nks * 4) + s * 4 + q];
}
template <typename T>
int relayout_fragment_major(const T* w, T* out, int N, int K, hipStream_t st) {
  if (!fragment_major_ok(N, K, (int)sizeof(T))) {
    set_error("relayout_fragment_major: [%d][%d] does not tile", N, K);
    return VLG_ERR_BAD_SHAPE;
  }
  const long long chunks = (long long)N * K * (long long)sizeof(T) / 16;
  relayout_fm_kernel<<<dim3((unsigned)((chunks + 255) / 256)), 256, 0, st>>>(reinterpret_cast<const u32x4_t*>(w), reinterpret_cast<u32x4_t*>(out), chunks,
                                                                              (int)((long long)K * sizeof(T) / 64));
  return VLG_OK;
}
template int relayout_fragment_major<float>(const float*, float*, int, int, hipStream_t);
template int relayout_fragment_major<bf16>(const bf16*, bf16*, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// slab-reduce epilogues
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void reduce_store_kernel(const float* __restrict__ ws, int splits, T* __restrict__ out,
                                                           float* __restrict__ out_f32, long long MN, int act, const T* __restrict__ bias,
                                                           int N) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= MN) return;
  float part[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) part[k] = k < splits ? ws[(size_t)k * MN + i] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += part[k];
  if (bias) s += DT<T>::ld(bias + i % N);
  s = DT<T>::rt(s);
  if (act == ACT_GELU_TANH) s = DT<T>::rt(gelu_tanh_f(s));
  if (act == ACT_SILU) s = DT<T>::rt(silu_f(s));
  if (out) DT<T>::st(out + i, s);
  if (out_f32) out_f32[i] = s;
}

template <typename T>
int reduce_store(const float* ws, int splits, T* out, float* out_f32, int M, int N, int act, hipStream_t st, const T* bias) {
  long long MN = (long long)M * N;
  reduce_store_kernel<T><<<dim3((unsigned)((MN + 255) / 256)), 256, 0, st>>>(ws, splits, out, out_f32, MN, act, bias, N);
  return VLG_OK;
}
template int reduce_store<float>(const float*, int, float*, float*, int, int, int, hipStream_t, const float*);
template int reduce_store<bf16>(const float*, int, bf16*, float*, int, int, int, hipStream_t, const bf16*);

// NV = 4-element vectors per thread (row = TPB threads x NV x 4 elements): one workgroup per row, the whole row in
// registers, every split-K slab requested up front (compile-time bound, predicated) with 16-byte loads.  TPB = 1024 with one vector per
// thread for the few-row launches of the decode slab path (config 5: 64 workgroups on 256 compute units, each a chain of dependent loads -
// four times the loads in flight per row; round 4).
template <typename T, int NV, int TPB = 256>
__global__ __launch_bounds__(TPB) void reduce_residual_rmsnorm_kernel(const float* __restrict__ ws, int splits,
                                                                      T* __restrict__ h, const T* __restrict__ w,
                                                                      T* __restrict__ hn, int M, int D, float eps) {
  __shared__ float red[TPB / 64];
  const int m = blockIdx.x;
  T* hr = h + (size_t)m * D;
  float v[NV][4], g[NV][4];
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int i = (e * TPB + threadIdx.x) * 4;
    if (i < D) {
      const Pack<T, 4> pv = *reinterpret_cast<const Pack<T, 4>*>(hr + i);
      const Pack<T, 4> pg = *reinterpret_cast<const Pack<T, 4>*>(w + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[e][j] = DT<T>::ld(&pv.v[j]);
        g[e][j] = DT<T>::ld(&pg.v[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[e][j] = g[e][j] = 0.f;
    }
  }
  if (ws) {
    float4 part[8][NV];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float* row = ws + ((size_t)(k < splits ? k : 0) * M + m) * D;
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const int i = (e * TPB + threadIdx.x) * 4;
        part[k][e] = (k < splits && i < D) ? *reinterpret_cast<const float4*>(row + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s[0] += part[k][e].x;
        s[1] += part[k][e].y;
        s[2] += part[k][e].z;
        s[3] += part[k][e].w;
      }
      const int i = (e * TPB + threadIdx.x) * 4;
      if (i < D) {
        Pack<T, 4> po;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[e][j] = DT<T>::rt(v[e][j] + DT<T>::rt(s[j]));
          DT<T>::st(&po.v[j], v[e][j]);
        }
        *reinterpret_cast<Pack<T, 4>*>(hr + i) = po;
      }
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < NV; ++e)
#pragma unroll
    for (int j = 0; j < 4; ++j) ss += v[e][j] * v[e][j];
  ss = block_sum_n<TPB>(ss, red);
  const float rs = 1.0f / sqrtf(ss / (float)D + eps);
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int i = (e * TPB + threadIdx.x) * 4;
    if (i < D) {
      Pack<T, 4> po;
#pragma unroll
      for (int j = 0; j < 4; ++j) DT<T>::st(&po.v[j], DT<T>::rt(v[e][j] * rs) * g[e][j]);
      *reinterpret_cast<Pack<T, 4>*>(hn + (size_t)m * D + i) = po;
    }
  }
}

template <typename T>
int reduce_residual_rmsnorm(const float* ws, int splits, T* h, const T* w, T* hn, int M, int D, float eps, hipStream_t st) {
  if (D % 4 != 0 || D > 4096 || splits > 8) {
    set_error("rmsnorm: dim %d / splits %d not supported (dim %% 4 == 0, dim <= 4096)", D, splits);
    return VLG_ERR_UNSUPPORTED;
  }
  static const bool wide_off = getenv("VLG_NORM_WIDE") != nullptr && atoi(getenv("VLG_NORM_WIDE")) == 0;   // A/B knob
  if (!wide_off && D > 1024 && M <= 256)
    reduce_residual_rmsnorm_kernel<T, 1, 1024><<<M, 1024, 0, st>>>(ws, splits, h, w, hn, M, D, eps);
  else if (D <= 1024)
    reduce_residual_rmsnorm_kernel<T, 1><<<M, 256, 0, st>>>(ws, splits, h, w, hn, M, D, eps);
  else if (D <= 2048)
    reduce_residual_rmsnorm_kernel<T, 2><<<M, 256, 0, st>>>(ws, splits, h, w, hn, M, D, eps);
  else
    reduce_residual_rmsnorm_kernel<T, 4><<<M, 256, 0, st>>>(ws, splits, h, w, hn, M, D, eps);
  return VLG_OK;
}
template int reduce_residual_rmsnorm<float>(const float*, int, float*, const float*, float*, int, int, float, hipStream_t);
template int reduce_residual_rmsnorm<bf16>(const float*, int, bf16*, const bf16*, bf16*, int, int, float, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void reduce_silu_mul_kernel(const float* __restrict__ ws, int splits, T* __restrict__ g,
                                                              int M, int F) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
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
  DT<T>::st(g + i, DT<T>::rt(silu_f(a)) * b);
}

template <typename T>
int reduce_silu_mul(const float* ws, int splits, T* g, int M, int F, hipStream_t st) {
  long long n = (long long)M * F;
  reduce_silu_mul_kernel<T><<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(ws, splits, g, M, F);
  return VLG_OK;
}
template int reduce_silu_mul<float>(const float*, int, float*, int, int, hipStream_t);
template int reduce_silu_mul<bf16>(const float*, int, bf16*, int, int, hipStream_t);

// qkv slabs -> RoPE(q), RoPE(k) (adjacent pairs, fp32, table row = position; rows of condition positions are
// all-zero: SURVEY Q1) -> q buffer + KV cache scatter at position p
template <typename T>
__global__ __launch_bounds__(256) void qkv_rope_scatter_kernel(const float* __restrict__ ws, int splits, T* __restrict__ qbuf,
                                                               T* __restrict__ kc, T* __restrict__ vc,
                                                               const float* __restrict__ freqs,
                                                               const StepState* __restrict__ state, int M, int Tq, int H,
                                                               int hd, int S, const int32_t* __restrict__ row_pos, KvPages pages) {
  const int D = H * hd;
  const int pairs = 3 * D / 2;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)M * pairs) return;
  const int m = (int)(i / pairs), j = (int)(i % pairs);
  const int col = 2 * j;
  const int sec = col / D, within = col % D;
  const int hh = within / hd, d = within % hd;
  const int b = m / Tq, t = m % Tq;
  const int p = (row_pos ? row_pos[b] : state->pos) + t;
  float x0 = 0.f, x1 = 0.f;
  float2 part[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    part[k] = k < splits ? *reinterpret_cast<const float2*>(ws + ((size_t)k * M + m) * (3 * (size_t)D) + col) : make_float2(0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    x0 += part[k].x;
    x1 += part[k].y;
  }
  x0 = DT<T>::rt(x0);
  x1 = DT<T>::rt(x1);
  float o0 = x0, o1 = x1;
  if (sec < 2) {
    const float2 cs = *reinterpret_cast<const float2*>(freqs + ((size_t)p * (hd / 2) + d / 2) * 2);
    o0 = __fsub_rn(__fmul_rn(x0, cs.x), __fmul_rn(x1, cs.y));
    o1 = __fadd_rn(__fmul_rn(x1, cs.x), __fmul_rn(x0, cs.y));
  }
  T* dst;
  if (sec == 0)
    dst = qbuf + ((size_t)m * H + hh) * hd + d;
  else
    dst = (sec == 1 ? kc : vc) + kv_row_index(pages, b, hh, H, S, p) * hd + d;
  DT<T>::st(dst, o0);
  DT<T>::st(dst + 1, o1);
}

template <typename T>
int qkv_rope_scatter(const float* ws, int splits, T* qbuf, T* kcache, T* vcache, const float* freqs, const StepState* state,
                     int M, int Tq, int H, int hd, int S, hipStream_t st, const int32_t* row_pos, KvPages pages) {
  long long n = (long long)M * (3 * H * hd / 2);
  qkv_rope_scatter_kernel<T><<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(ws, splits, qbuf, kcache, vcache, freqs, state,
                                                                                M, Tq, H, hd, S, row_pos, pages);
  return VLG_OK;
}
template int qkv_rope_scatter<float>(const float*, int, float*, float*, float*, const float*, const StepState*, int, int, int, int, int, hipStream_t,
                                     const int32_t*, KvPages);
template int qkv_rope_scatter<bf16>(const float*, int, bf16*, bf16*, bf16*, const float*, const StepState*, int, int, int, int, int, hipStream_t,
                                    const int32_t*, KvPages);

// ------------------------------------------------------------------------------------------------
// split-KV attention for one query row per (row m, head h)
//   lane = (g, c): g = row group (64/LPR rows per wave-load), c = 16 B (or 8 B) column chunk of the head dim.
//   Each lane group keeps its own online-softmax state (m, l, acc[VEC]); groups -> waves -> splits are merged
//   with the usual (m, l, acc) rescale.
// ------------------------------------------------------------------------------------------------

// PAGED: block-granular cache (KvPages); the batch row's block table is staged in LDS and every cache row address goes through it.
template <typename T, int HD, int VEC, int LPR, int U, bool PAGED = false>
__global__ __launch_bounds__(256) void attn_partial_kernel(const T* __restrict__ qbuf, T* __restrict__ kc,
                                                           T* __restrict__ vc, float* __restrict__ ws,
                                                           T* __restrict__ out, const StepState* __restrict__ state,
                                                           int Tq, int H, int S, const float* __restrict__ mask, int Bmask,
                                                           int Tc, float scale, const int32_t* __restrict__ row_pos, KvPages pg = KvPages{},
                                                           int out_nks = 0) {
  constexpr int RPI = 64 / LPR;  // rows per wave-wide load; U = loads in flight per operand
  constexpr int TILE = RPI * U;
  const int split = blockIdx.x, nsplit = gridDim.x, h = blockIdx.y, m = blockIdx.z;
  const int b = m / Tq, t = m % Tq;
  const int p = (row_pos ? row_pos[b] : state->pos) + t;
  const int nkeys = p + 1;
  const int chunk = (nkeys + nsplit - 1) / nsplit;
  const int r0 = split * chunk;
  const int r1 = min(r0 + chunk, nkeys);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, c = lane % LPR;
  const bool active = c * VEC < HD;
  const int coff = active ? c * VEC : 0;

  float qf[VEC];
  {
    const Pack<T, VEC> qp = *reinterpret_cast<const Pack<T, VEC>*>(qbuf + ((size_t)m * H + h) * HD + coff);
#pragma unroll
    for (int j = 0; j < VEC; ++j) qf[j] = active ? DT<T>::ld(&qp.v[j]) : 0.f;
  }
  __shared__ int32_t blk_s[PAGED ? 256 : 1];
  if constexpr (PAGED) {
    for (int i = threadIdx.x; i < pg.stride; i += 256) blk_s[i] = pg.table[(size_t)b * pg.stride + i];
    __syncthreads();
  }
  T* kbase = PAGED ? kc + (((size_t)h << pg.shift) * HD + coff) : kc + ((size_t)b * H + h) * (size_t)S * HD + coff;
  T* vbase = PAGED ? vc + (((size_t)h << pg.shift) * HD + coff) : vc + ((size_t)b * H + h) * (size_t)S * HD + coff;
  // offset (elements) of cache row rr from kbase / vbase
  auto rowoff = [&](int rr) -> size_t {
    if constexpr (PAGED)
      return ((((size_t)blk_s[rr >> pg.shift] * H) << pg.shift) + (size_t)(rr & ((1 << pg.shift) - 1))) * HD;
    else
      return (size_t)rr * HD;
  };
  const float* mrow = (mask != nullptr) ? mask + (size_t)(b % Bmask) * Tc : nullptr;

  float mx = -INFINITY, l = 0.f, acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;

  for (int tile = r0 + wave * TILE; tile < r1; tile += 4 * TILE) {
    Pack<T, VEC> kk[U], vv[U];
    int rows[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rows[u] = tile + u * RPI + g;
      const int rr = rows[u] < r1 ? rows[u] : r1 - 1;
      const size_t ro = rowoff(rr);
      kk[u] = load_stream<T, VEC>(kbase + ro);
      vv[u] = load_stream<T, VEC>(vbase + ro);
    }
    float s[U];
    float tmax = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) d = fmaf(qf[j], DT<T>::ld(&kk[u].v[j]), d);
      d = group_sum<LPR>(d) * scale;
      bool ok = rows[u] < r1;
      if (mrow != nullptr && rows[u] < Tc && rows[u] != p) ok = ok && (mrow[rows[u] < Tc ? rows[u] : 0] != 0.f);
      s[u] = ok ? d : -INFINITY;
      tmax = fmaxf(tmax, s[u]);
    }
    const float mnew = fmaxf(mx, tmax);
    const float mref = (mnew == -INFINITY) ? 0.f : mnew;
    const float alpha = __expf(mx - mref);
    l *= alpha;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] *= alpha;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pu = __expf(s[u] - mref);
      l += pu;
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = fmaf(pu, DT<T>::ld(&vv[u].v[j]), acc[j]);
    }
    mx = mnew;
  }

  // merge the RPI lane groups of this wave
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    const float mo = __shfl_xor(mx, off), lo = __shfl_xor(l, off);
    const float mn = fmaxf(mx, mo);
    const float mref = (mn == -INFINITY) ? 0.f : mn;
    const float a = __expf(mx - mref), bb = __expf(mo - mref);
    l = l * a + lo * bb;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = acc[j] * a + __shfl_xor(acc[j], off) * bb;
    mx = mn;
  }
  // merge the 4 waves through LDS
  __shared__ float sm[4][HD + 2];
  if (g == 0 && active) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm[wave][2 + coff + j] = acc[j];
    if (c == 0) {
      sm[wave][0] = mx;
      sm[wave][1] = l;
    }
  }
  __syncthreads();
  const int d = threadIdx.x;
  if (d < HD) {
    float M4 = fmaxf(fmaxf(sm[0][0], sm[1][0]), fmaxf(sm[2][0], sm[3][0]));
    const float mref = (M4 == -INFINITY) ? 0.f : M4;
    float L = 0.f, A = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      const float e = __expf(sm[wv][0] - mref);
      L += sm[wv][1] * e;
      A += sm[wv][2 + d] * e;
    }
    if (nsplit == 1) {
      DT<T>::st(out + (out_nks ? afm_index<T>(m, h * HD + d, out_nks) : ((size_t)m * H + h) * HD + d), A / L);
    } else {
      float* o = ws + (((size_t)m * H + h) * nsplit + split) * (HD + 2);
      o[2 + d] = A;
      if (d == 0) {
        o[0] = M4;
        o[1] = L;
      }
    }
  }
}

// NS: partials requested up front (8 for nsplit <= 8; 16 for the deeper splits of small batches, where this launch is as long as the
// attention kernel itself if it walks the partials one dependent load at a time)
template <typename T, int HD, int NS>
__global__ __launch_bounds__(64) void attn_combine_kernel(const float* __restrict__ ws, T* __restrict__ out, int nsplit, int H, int out_nks) {
  const size_t mh = blockIdx.x;
  const int om = (int)(mh / H), oc0 = (int)(mh % H) * HD;   // (row, first column) of this head's output in the [M, H hd] matrix
  const float* base = ws + mh * nsplit * (HD + 2);
  constexpr int ND = (HD + 63) / 64;
  if (nsplit <= NS) {
    // every partial this thread needs is requested before the first use: one memory round trip instead of three
    float mv[NS], lv[NS], av[NS][ND];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float* b = base + (size_t)(s < nsplit ? s : 0) * (HD + 2);
      mv[s] = b[0];
      lv[s] = b[1];
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        const int d = threadIdx.x + 64 * k;
        av[s][k] = b[2 + (d < HD ? d : 0)];
      }
    }
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < nsplit) M = fmaxf(M, mv[s]);
    const float mref = (M == -INFINITY) ? 0.f : M;
    float L = 0.f, A[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) A[k] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < nsplit) {
        const float e = __expf(mv[s] - mref);
        L += lv[s] * e;
#pragma unroll
        for (int k = 0; k < ND; ++k) A[k] += av[s][k] * e;
      }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int d = threadIdx.x + 64 * k;
      if (d < HD) DT<T>::st(out + (out_nks ? afm_index<T>(om, oc0 + d, out_nks) : mh * HD + d), A[k] / L);
    }
    return;
  }
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, base[(size_t)s * (HD + 2)]);
  const float mref = (M == -INFINITY) ? 0.f : M;
  float L = 0.f;
  for (int s = 0; s < nsplit; ++s) L += base[(size_t)s * (HD + 2) + 1] * __expf(base[(size_t)s * (HD + 2)] - mref);
  for (int d = threadIdx.x; d < HD; d += 64) {
    float A = 0.f;
    for (int s = 0; s < nsplit; ++s) A += base[(size_t)s * (HD + 2) + 2 + d] * __expf(base[(size_t)s * (HD + 2)] - mref);
    DT<T>::st(out + (out_nks ? afm_index<T>(om, oc0 + d, out_nks) : mh * HD + d), A / L);
  }
}

// ------------------------------------------------------------------------------------------------
// Prefill attention (Tq > 1 query rows per batch row, generate.py:77-86,156-165): one workgroup per (head, batch row).  The condition's
// K and V rows (<= 120 x head_dim) are staged in LDS ONCE and every query row of the (b, h) pair is served from there - the split-KV decode
// kernel walks the cache once per query row (76,800 workgroups for 32 x 120 rows x 20 heads: 106 us per layer; this form: 640 workgroups).
// gridDim.z workgroups share a (b, h) pair when there are few pairs (each stages K / V again: 30 KB).  Wave w takes rows w, w + 4, ...: lanes = keys for the scores (q broadcast from LDS, K row per lane), lanes = head dims for P.V (one
// probability broadcast per key).  fp32 scores / softmax / accumulation as the decode kernel; the same mask rule (a padded condition key
// is dropped unless it is the row's own position).
// ------------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void prefill_attn_kernel(const T* __restrict__ qbuf, const T* __restrict__ kc, const T* __restrict__ vc,
                                                           T* __restrict__ out, const StepState* __restrict__ state, int Tq, int H, int S, int nk,
                                                           const float* __restrict__ mask, int Bmask, int Tc, float scale) {
  constexpr int EPV = 16 / (int)sizeof(T);                 // elements per 16-byte chunk
  constexpr int ROWB = HD * (int)sizeof(T) + 16;           // LDS row pitch (+ 16 bytes: consecutive keys start on different banks)
  constexpr int DPL = (HD + 63) / 64;                      // head dims per lane in the P.V phase
  extern __shared__ __attribute__((aligned(16))) char pf_smem[];
  char* Ks = pf_smem;
  char* Vs = Ks + (size_t)nk * ROWB;
  float* qs = reinterpret_cast<float*>(Vs + (size_t)nk * ROWB);   // [4 waves][HD]
  const int nkp = (nk + 63) / 64 * 64;
  float* ps = qs + 4 * HD;                                          // [4 waves][nkp]
  const int h = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pos = state->pos;
  const T* kb = kc + ((size_t)b * H + h) * (size_t)S * HD;
  const T* vb = vc + ((size_t)b * H + h) * (size_t)S * HD;
  constexpr int CPR = HD * (int)sizeof(T) / 16;             // 16-byte chunks per row (HD * sizeof(T) % 16 == 0: launcher)
  for (int i = threadIdx.x; i < nk * CPR; i += 256) {
    const int row = i / CPR, c = i - row * CPR;
    *reinterpret_cast<u32x4_t*>(Ks + (size_t)row * ROWB + c * 16) = reinterpret_cast<const u32x4_t*>(kb + (size_t)row * HD)[c];
    *reinterpret_cast<u32x4_t*>(Vs + (size_t)row * ROWB + c * 16) = reinterpret_cast<const u32x4_t*>(vb + (size_t)row * HD)[c];
  }
  __syncthreads();
  const float* mrow = (mask != nullptr) ? mask + (size_t)(b % Bmask) * Tc : nullptr;
  float* qw = qs + wave * HD;
  float* pw = ps + (size_t)wave * nkp;
  for (int t = wave + 4 * blockIdx.z; t < Tq; t += 4 * gridDim.z) {   // rows interleaved over waves and the grid's z: causal lengths balance
    const int m = b * Tq + t;
    int p = pos + t;
    p = p < nk ? p : nk - 1;                                 // host contract: pos + Tq <= nk
    for (int d = lane; d < HD; d += 64) qw[d] = DT<T>::ld(qbuf + ((size_t)m * H + h) * HD + d);
    // scores: key j = lane + 64 i
    float sc[2];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = lane + 64 * i;
      float dot = 0.f;
      if (j <= p) {
        const char* kr = Ks + (size_t)j * ROWB;
#pragma unroll
        for (int c = 0; c < CPR; ++c) {
          const u32x4_t kv = *reinterpret_cast<const u32x4_t*>(kr + c * 16);
          const T* ke = reinterpret_cast<const T*>(&kv);
#pragma unroll
          for (int e = 0; e < EPV; ++e) dot = fmaf(qw[c * EPV + e], DT<T>::ld(ke + e), dot);
        }
      }
      bool ok = j <= p;
      if (ok && mrow != nullptr && j < Tc && j != p) ok = mrow[j] != 0.f;
      sc[i] = ok ? dot * scale : -INFINITY;
      mx = fmaxf(mx, sc[i]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float mref = (mx == -INFINITY) ? 0.f : mx;
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = lane + 64 * i;
      const float e = __expf(sc[i] - mref);
      l += e;
      if (j < nkp) pw[j] = e;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) l += __shfl_xor(l, o);
    // P.V: lane owns head dims lane, lane + 64
    float acc[DPL];
#pragma unroll
    for (int k = 0; k < DPL; ++k) acc[k] = 0.f;
    for (int j = 0; j <= p; ++j) {
      const float pj = pw[j];
      const T* vr = reinterpret_cast<const T*>(Vs + (size_t)j * ROWB);
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        const int d = lane + 64 * k;
        if (d < HD) acc[k] = fmaf(pj, DT<T>::ld(vr + d), acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int d = lane + 64 * k;
      if (d < HD) DT<T>::st(out + ((size_t)m * H + h) * HD + d, acc[k] / l);
    }
  }
}

size_t attn_ws_floats(int M, int H, int hd) { return (size_t)M * H * 16 * (hd + 2); }

template <typename T, int HD, int VEC, int LPR>
static int attn_launch(const T* qbuf, T* kc, T* vc, T* out, float* ws, const StepState* state, int Bp, int Tq, int H,
                       int S, int max_pos, const float* mask, int Bmask, int Tc, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                       const int32_t* row_pos, KvPages pages, int out_nks) {
  const int M = Bp * Tq;
  {
    // prefill (several query rows per batch row at uniform positions, contiguous cache): K / V staged once per (batch row, head)
    static const bool pf_off = getenv("VLG_PREFILL_ATTN") != nullptr && atoi(getenv("VLG_PREFILL_ATTN")) == 0;   // A/B knob
    const int nk = max_pos + 1;
    const size_t lds = (size_t)2 * nk * (HD * sizeof(T) + 16) + (size_t)4 * HD * sizeof(float) + (size_t)4 * ((nk + 63) / 64 * 64) * sizeof(float);
    if (!pf_off && Tq > 1 && nk <= 128 && nk >= Tq && row_pos == nullptr && pages.table == nullptr && out_nks == 0 && (HD * sizeof(T)) % 16 == 0 &&
        lds <= 150 * 1024) {
      static LdsAttrOnce attr_once;   // per instantiation; the attribute itself is per device (common.h)
      VLG_TRY(set_max_dynamic_lds(attr_once, {reinterpret_cast<const void*>(prefill_attn_kernel<T, HD>)}, 150 * 1024));
      if (ev0) (void)hipEventRecord(ev0, st);
      static const int z_knob = getenv("VLG_PREFILL_Z") ? atoi(getenv("VLG_PREFILL_Z")) : 0;   // A/B knob: workgroups per (b, h) pair
      const int zs = z_knob > 0 ? std::min(z_knob, 30) : 8;   // measured 1 / 2 / 4 / 8 on 32 x 120 and 8 x 120 rows: 8 is fastest on both (staging K / V again costs less than idle CUs)
      prefill_attn_kernel<T, HD><<<dim3(H, Bp, zs), 256, lds, st>>>(qbuf, kc, vc, out, state, Tq, H, S, nk, mask, Bmask, Tc, 1.0f / sqrtf((float)HD));
      VLG_HIP(hipGetLastError());
      if (ev1) (void)hipEventRecord(ev1, st);
      return VLG_OK;
    }
  }
  // grid sizing measured on MI355X with non-temporal KV loads (tools/bench_kernels.py attn, B'H = 640): 2560 workgroups
  // (nsplit 4) 72.8 us vs 1536-cap (nsplit 2) 74.8 us at p = 2679.  Not splitting at all (640 workgroups, 8 loads in flight)
  // saves the combine launch but the kernel itself runs 4.6 us longer inside the decode step (75.6 vs 71.0 us average):
  // 21.92 vs 21.81 s/step, so the split stays.  VLG_ATTN_CAP / VLG_ATTN_U are tuning knobs.
  static const int cap_knob = getenv("VLG_ATTN_CAP") ? atoi(getenv("VLG_ATTN_CAP")) : 2560;
  int nsplit = cap_knob / (M * H);
  const int by_len = (max_pos + 1 + 63) / 64;
  if (nsplit > by_len) nsplit = by_len;
  if (nsplit > 16) nsplit = 16;
  if (nsplit < 1) nsplit = 1;
  const float scale = 1.0f / sqrtf((float)HD);
  if (ev0) (void)hipEventRecord(ev0, st);
  if (pages.table != nullptr) {
    if (pages.stride > 256) {
      set_error("block-granular KV: a row may hold at most 256 blocks (has %d)", pages.stride);
      return VLG_ERR_UNSUPPORTED;
    }
    attn_partial_kernel<T, HD, VEC, LPR, 4, true><<<dim3(nsplit, H, M), 256, 0, st>>>(qbuf, kc, vc, ws, out, state, Tq, H, S, mask, Bmask, Tc, scale,
                                                                                     row_pos, pages, out_nks);
  } else {
    static const int u_knob = getenv("VLG_ATTN_U") ? atoi(getenv("VLG_ATTN_U")) : 4;
    if (u_knob == 8)
      attn_partial_kernel<T, HD, VEC, LPR, 8><<<dim3(nsplit, H, M), 256, 0, st>>>(qbuf, kc, vc, ws, out, state, Tq, H, S, mask, Bmask, Tc, scale, row_pos, KvPages{}, out_nks);
    else if (u_knob == 2)
      attn_partial_kernel<T, HD, VEC, LPR, 2><<<dim3(nsplit, H, M), 256, 0, st>>>(qbuf, kc, vc, ws, out, state, Tq, H, S, mask, Bmask, Tc, scale, row_pos, KvPages{}, out_nks);
    else
      attn_partial_kernel<T, HD, VEC, LPR, 4><<<dim3(nsplit, H, M), 256, 0, st>>>(qbuf, kc, vc, ws, out, state, Tq, H, S, mask, Bmask, Tc, scale, row_pos, KvPages{}, out_nks);
  }
  if (ev1) (void)hipEventRecord(ev1, st);
  if (nsplit > 1) {
    if (nsplit <= 8)
      attn_combine_kernel<T, HD, 8><<<M * H, 64, 0, st>>>(ws, out, nsplit, H, out_nks);
    else
      attn_combine_kernel<T, HD, 16><<<M * H, 64, 0, st>>>(ws, out, nsplit, H, out_nks);
  }
  return VLG_OK;
}

template <typename T>
int attn_rows(const T* qbuf, T* kc, T* vc, T* out, float* ws, const StepState* state, int Bp, int Tq, int H, int hd,
              int S, int max_pos, const float* mask, int Bmask, int Tc, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
              const int32_t* row_pos, KvPages pages, int out_nks) {
#define VLG_ATTN(HD_, VEC_, LPR_) \
  return attn_launch<T, HD_, VEC_, LPR_>(qbuf, kc, vc, out, ws, state, Bp, Tq, H, S, max_pos, mask, Bmask, Tc, st, ev0, ev1, row_pos, pages, out_nks)
  if constexpr (sizeof(T) == 2) {
    if (hd == 64) VLG_ATTN(64, 8, 8);
    if (hd == 128) VLG_ATTN(128, 8, 16);
    if (hd == 100) VLG_ATTN(100, 4, 32);
    if (hd == 96) VLG_ATTN(96, 8, 16);
    if (hd == 32) VLG_ATTN(32, 8, 4);
  } else {
    if (hd == 64) VLG_ATTN(64, 4, 16);
    if (hd == 128) VLG_ATTN(128, 4, 32);
    if (hd == 100) VLG_ATTN(100, 4, 32);
    if (hd == 96) VLG_ATTN(96, 4, 32);
    if (hd == 32) VLG_ATTN(32, 4, 8);
  }
#undef VLG_ATTN
  set_error("attention: unsupported head_dim %d (supported: 32, 64, 96, 100, 128)", hd);
  return VLG_ERR_UNSUPPORTED;
}
template int attn_rows<float>(const float*, float*, float*, float*, float*, const StepState*, int, int, int, int, int, int, const float*, int, int, hipStream_t, hipEvent_t, hipEvent_t, const int32_t*, KvPages, int);
template int attn_rows<bf16>(const bf16*, bf16*, bf16*, bf16*, float*, const StepState*, int, int, int, int, int, int, const float*, int, int, hipStream_t, hipEvent_t, hipEvent_t, const int32_t*, KvPages, int);

// ------------------------------------------------------------------------------------------------
// gathers and small glue kernels
// ------------------------------------------------------------------------------------------------
template <typename T, typename I>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ table, const I* __restrict__ idx, int n_idx,
                                                          int null_id, T* __restrict__ out, int rows, int D, int n_rows, int out_nks = 0) {
  const int r = blockIdx.x;
  long long id = r < n_idx ? (long long)idx[r] : (long long)null_id;
  id = id < 0 ? 0 : (id >= n_rows ? n_rows - 1 : id);   // never fault on a bad id
  const T* src = table + (size_t)id * D;
  for (int i = threadIdx.x; i < D; i += 256) out[out_nks ? afm_index<T>(r, i, out_nks) : (size_t)r * D + i] = src[i];
}
template <typename T>
int gather_rows_i32(const T* table, const int32_t* idx, T* out, int rows, int D, int n_rows, hipStream_t st, int out_nks) {
  gather_rows_kernel<T, int32_t><<<rows, 256, 0, st>>>(table, idx, rows, 0, out, rows, D, n_rows, out_nks);
  return VLG_OK;
}
template <typename T>
int gather_rows_i64(const T* table, const int64_t* idx, int n_idx, int null_id, T* out, int rows, int D, int n_rows, hipStream_t st) {
  gather_rows_kernel<T, int64_t><<<rows, 256, 0, st>>>(table, idx, n_idx, null_id, out, rows, D, n_rows);
  return VLG_OK;
}
template int gather_rows_i32<float>(const float*, const int32_t*, float*, int, int, int, hipStream_t, int);
template int gather_rows_i32<bf16>(const bf16*, const int32_t*, bf16*, int, int, int, hipStream_t, int);
template int gather_rows_i64<float>(const float*, const int64_t*, int, int, float*, int, int, int, hipStream_t);
template int gather_rows_i64<bf16>(const bf16*, const int64_t*, int, int, bf16*, int, int, int, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void build_text_cond_kernel(const float* __restrict__ cond, const T* __restrict__ uncond,
                                                              T* __restrict__ out, int B, int Tc, int cd, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long per = (long long)Tc * cd;
  const int b = (int)(i / per);
  const long long rem = i % per;
  if (b < B)
    DT<T>::st(out + i, cond[i]);
  else
    out[i] = uncond[rem];  // zeros_like(cond) + uncond_embedding (generate.py:138)
}
template <typename T>
int build_text_cond(const float* cond, const T* uncond, T* out, int B, int Bp, int Tc, int cd, hipStream_t st) {
  long long total = (long long)Bp * Tc * cd;
  build_text_cond_kernel<T><<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(cond, uncond, out, B, Tc, cd, total);
  return VLG_OK;
}
template int build_text_cond<float>(const float*, const float*, float*, int, int, int, int, hipStream_t);
template int build_text_cond<bf16>(const float*, const bf16*, bf16*, int, int, int, int, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void take_last_rows_kernel(const T* __restrict__ x, T* __restrict__ out, int Tq, int D) {
  const int b = blockIdx.x;
  const T* src = x + ((size_t)b * Tq + (Tq - 1)) * D;
  for (int i = threadIdx.x; i < D; i += 256) out[(size_t)b * D + i] = src[i];
}
template <typename T>
int take_last_rows(const T* x, T* out, int Bp, int Tq, int D, hipStream_t st) {
  take_last_rows_kernel<T><<<Bp, 256, 0, st>>>(x, out, Tq, D);
  return VLG_OK;
}
template int take_last_rows<float>(const float*, float*, int, int, int, hipStream_t);
template int take_last_rows<bf16>(const bf16*, bf16*, int, int, int, hipStream_t);

// x [Bp][Tq][D] -> out [Bp][Tq - 1][D]: every batch row without its last position (session prefill of several slots: the last condition
// token is not run through the layers, it becomes the slot's first decode input)
template <typename T>
__global__ __launch_bounds__(256) void drop_last_rows_kernel(const T* __restrict__ x, T* __restrict__ out, int Tq, int D) {
  const int b = blockIdx.y, t = blockIdx.x;   // t < Tq - 1
  const T* src = x + ((size_t)b * Tq + t) * D;
  T* dst = out + ((size_t)b * (Tq - 1) + t) * D;
  for (int i = threadIdx.x; i < D; i += 256) dst[i] = src[i];
}
template <typename T>
int drop_last_rows(const T* x, T* out, int Bp, int Tq, int D, hipStream_t st) {
  if (Tq > 1) drop_last_rows_kernel<T><<<dim3(Tq - 1, Bp), 256, 0, st>>>(x, out, Tq, D);
  return VLG_OK;
}
template int drop_last_rows<float>(const float*, float*, int, int, int, hipStream_t);
template int drop_last_rows<bf16>(const bf16*, bf16*, int, int, int, hipStream_t);

template <typename T>
__global__ void latent_to_rows_kernel(const float* __restrict__ cur, T* __restrict__ out, int B, int Bp, int C) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bp * C) return;
  const int b = i / C, cidx = i % C;
  DT<T>::st(out + i, cur[(size_t)(b % B) * C + cidx]);
}
template <typename T>
int latent_to_rows(const float* cur, T* out, int B, int Bp, int C, hipStream_t st) {
  latent_to_rows_kernel<T><<<cdiv(Bp * C, 256), 256, 0, st>>>(cur, out, B, Bp, C);
  return VLG_OK;
}
template int latent_to_rows<float>(const float*, float*, int, int, int, hipStream_t);
template int latent_to_rows<bf16>(const float*, bf16*, int, int, int, hipStream_t);

template <typename T>
__global__ void latent_head_finish_kernel(const T* __restrict__ y, float* __restrict__ cur, float* __restrict__ out_lat,
                                          float* __restrict__ trace, const StepState* __restrict__ state, int B, int Bp, int C,
                                          int N, float cfg_scale, int cfg_interval, int b_off, int B_total,
                                          const int32_t* __restrict__ row_step) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, cidx = i % C;
  const int step = row_step ? row_step[b] : state->step;   // sessions: every slot at its own token index
  float v = DT<T>::ld(y + (size_t)b * C + cidx);
  if (Bp > B) {
    // decode step i = step-1 of decode_n_tokens: cfg_flag off once i > cfg_interval >= 0 (generate.py:113-114)
    const bool flag = !(cfg_interval > -1 && (step - 1) > cfg_interval);
    if (flag) {
      const float u = DT<T>::ld(y + (size_t)(b + B) * C + cidx);
      v = DT<T>::rt(u + (v - u) * cfg_scale);
    }
  }
  cur[i] = v;
  out_lat[((size_t)b * N + step) * C + cidx] = v;
  if (trace) trace[((size_t)step * B_total + b_off + b) * C + cidx] = v;
}
template <typename T>
int latent_head_finish(const T* y, float* cur, float* out_lat, float* trace, const StepState* state, int B, int Bp, int C, int N,
                       float cfg_scale, int cfg_interval, hipStream_t st, int b_off, int B_total, const int32_t* row_step) {
  latent_head_finish_kernel<T><<<cdiv(B * C, 256), 256, 0, st>>>(y, cur, out_lat, trace, state, B, Bp, C, N, cfg_scale, cfg_interval, b_off,
                                                                 B_total > 0 ? B_total : B, row_step);
  return VLG_OK;
}
template int latent_head_finish<float>(const float*, float*, float*, float*, const StepState*, int, int, int, int, float, int, hipStream_t, int, int, const int32_t*);
template int latent_head_finish<bf16>(const bf16*, float*, float*, float*, const StepState*, int, int, int, int, float, int, hipStream_t, int, int, const int32_t*);

// ---- t2v decode step, latent side in two launches instead of six ---------------------------------------------------------
// latent_in: cur fp32 [B,C] -> (rows duplicated for CFG) -> t1[m][d] = rt(gelu_tanh(rt(sum_c rt(cur[m % B][c]) * W1[d][c])))
// = latent_to_rows + vae_latent_adapter.fc1 + GELU (gpt_video.py:296-297; K = C <= 16).  One workgroup per row.
template <typename T, int CMAX>
__global__ __launch_bounds__(256) void latent_in_fc1_kernel(const float* __restrict__ cur, const T* __restrict__ w1, T* __restrict__ t1, int B,
                                                            int C, int D) {
  __shared__ float xs[CMAX];
  const int m = blockIdx.x;
  if ((int)threadIdx.x < C) xs[threadIdx.x] = DT<T>::rt(cur[(size_t)(m % B) * C + threadIdx.x]);
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) acc = fmaf(xs[c], DT<T>::ld(w1 + (size_t)d * C + c), acc);
    DT<T>::st(t1 + (size_t)m * D + d, gelu_tanh_f(DT<T>::rt(acc)));
  }
}
template <typename T>
int latent_in_fc1(const float* cur, const T* w1, T* t1, int B, int Bp, int C, int D, hipStream_t st) {
  if (C > 16) {
    set_error("latent_in_fc1: vae_embed_dim %d > 16", C);
    return VLG_ERR_UNSUPPORTED;
  }
  latent_in_fc1_kernel<T, 16><<<Bp, 256, 0, st>>>(cur, w1, t1, B, C, D);
  return VLG_OK;
}
template int latent_in_fc1<float>(const float*, const float*, float*, int, int, int, int, hipStream_t);
template int latent_in_fc1<bf16>(const float*, const bf16*, bf16*, int, int, int, int, hipStream_t);

// latent_out: y[row][c] = rt(sum_k t1[row][k] * W2[c][k]) for the row (and its unconditional partner), then the CFG combine and the
// stores of latent_head_finish = vae_latent_adapter2.fc2 (N = C <= 16) + finish.  One workgroup per user row.
template <typename T, int CMAX>
__global__ __launch_bounds__(256) void latent_out_fc2_kernel(const T* __restrict__ t1, const T* __restrict__ w2, float* __restrict__ cur,
                                                             float* __restrict__ out_lat, float* __restrict__ trace,
                                                             const StepState* __restrict__ state, int B, int Bp, int C, int D, int N,
                                                             float cfg_scale, int cfg_interval, int b_off, int B_total,
                                                             const int32_t* __restrict__ row_step) {
  __shared__ float red[4][2][CMAX];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool pair = Bp > B;
  float ac[CMAX], au[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) ac[c] = au[c] = 0.f;
  constexpr int EPV = 16 / (int)sizeof(T);   // 16-byte vectors: all loads of a thread are issued before the first use
  if (D % EPV == 0) {
    for (int kv = threadIdx.x; kv < D / EPV; kv += 256) {
      const Pack<T, EPV> xc = *reinterpret_cast<const Pack<T, EPV>*>(t1 + (size_t)b * D + kv * EPV);
      Pack<T, EPV> xu = xc;
      if (pair) xu = *reinterpret_cast<const Pack<T, EPV>*>(t1 + (size_t)(b + B) * D + kv * EPV);
      Pack<T, EPV> wv[CMAX];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) wv[c] = *reinterpret_cast<const Pack<T, EPV>*>(w2 + (size_t)(c < C ? c : 0) * D + kv * EPV);
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) {
#pragma unroll
          for (int e = 0; e < EPV; ++e) {
            const float w_ = DT<T>::ld(&wv[c].v[e]);
            ac[c] = fmaf(DT<T>::ld(&xc.v[e]), w_, ac[c]);
            au[c] = fmaf(DT<T>::ld(&xu.v[e]), w_, au[c]);
          }
        }
    }
  } else {
    for (int k = threadIdx.x; k < D; k += 256) {
      const float xc = DT<T>::ld(t1 + (size_t)b * D + k);
      const float xu = pair ? DT<T>::ld(t1 + (size_t)(b + B) * D + k) : 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) {
          const float wv = DT<T>::ld(w2 + (size_t)c * D + k);
          ac[c] = fmaf(xc, wv, ac[c]);
          au[c] = fmaf(xu, wv, au[c]);
        }
    }
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    ac[c] = wave_sum(ac[c]);
    au[c] = wave_sum(au[c]);
    if (lane == 0) {
      red[wave][0][c] = ac[c];
      red[wave][1][c] = au[c];
    }
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= C) return;
  const int step = row_step ? row_step[b] : state->step;   // sessions: every slot at its own token index
  float v = DT<T>::rt(red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c]);
  if (pair) {
    const bool flag = !(cfg_interval > -1 && (step - 1) > cfg_interval);   // generate.py:113-114
    if (flag) {
      const float u = DT<T>::rt(red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c]);
      v = DT<T>::rt(u + (v - u) * cfg_scale);
    }
  }
  cur[(size_t)b * C + c] = v;
  out_lat[((size_t)b * N + step) * C + c] = v;
  if (trace) trace[((size_t)step * B_total + b_off + b) * C + c] = v;
}
template <typename T>
int latent_out_fc2(const T* t1, const T* w2, float* cur, float* out_lat, float* trace, const StepState* state, int B, int Bp, int C, int D, int N,
                   float cfg_scale, int cfg_interval, hipStream_t st, int b_off, int B_total, const int32_t* row_step) {
  if (C > 16) {
    set_error("latent_out_fc2: vae_embed_dim %d > 16", C);
    return VLG_ERR_UNSUPPORTED;
  }
  latent_out_fc2_kernel<T, 16><<<B, 256, 0, st>>>(t1, w2, cur, out_lat, trace, state, B, Bp, C, D, N, cfg_scale, cfg_interval, b_off,
                                                  B_total > 0 ? B_total : B, row_step);
  return VLG_OK;
}
template int latent_out_fc2<float>(const float*, const float*, float*, float*, float*, const StepState*, int, int, int, int, int, float, int,
                                   hipStream_t, int, int, const int32_t*);
template int latent_out_fc2<bf16>(const bf16*, const bf16*, float*, float*, float*, const StepState*, int, int, int, int, int, float, int,
                                  hipStream_t, int, int, const int32_t*);

// iteration-level batching: row m starts a request (row_cls[m] >= 0: its class embedding, the position-0 input of a c2i sequence)
// or continues one (row_cls[m] < 0: the embedding of the token it sampled in the previous iteration)
template <typename T>
__global__ __launch_bounds__(256) void gather_session_rows_kernel(const T* __restrict__ cls_table, int n_cls, const T* __restrict__ tok_table,
                                                                  int n_tok, const int32_t* __restrict__ row_cls,
                                                                  const int32_t* __restrict__ cur_tok, const T* __restrict__ pending,
                                                                  T* __restrict__ out, int D, int out_nks) {
  const int m = blockIdx.x;
  const int c = row_cls[m];
  const T* src;
  if (c >= 0 && cls_table != nullptr) {
    src = cls_table + (size_t)(c < n_cls ? c : n_cls - 1) * D;
  } else if (c == -3 || (c >= 0 && cls_table == nullptr)) {
    src = pending + (size_t)m * D;   // text-conditioned slot: the projected last condition token left by the slot's prefill
  } else {
    int tk = cur_tok[m];
    tk = tk < 0 ? 0 : (tk >= n_tok ? n_tok - 1 : tk);
    src = tok_table + (size_t)tk * D;
  }
  for (int i = threadIdx.x; i < D; i += 256) out[out_nks ? afm_index<T>(m, i, out_nks) : (size_t)m * D + i] = src[i];
}
template <typename T>
int gather_session_rows(const T* cls_table, int n_cls, const T* tok_table, int n_tok, const int32_t* row_cls, const int32_t* cur_tok,
                        const T* pending, T* out, int rows, int D, hipStream_t st, int out_nks) {
  gather_session_rows_kernel<T><<<rows, 256, 0, st>>>(cls_table, n_cls, tok_table, n_tok, row_cls, cur_tok, pending, out, D, out_nks);
  return VLG_OK;
}
template int gather_session_rows<float>(const float*, int, const float*, int, const int32_t*, const int32_t*, const float*, float*, int, int,
                                        hipStream_t, int);
template int gather_session_rows<bf16>(const bf16*, int, const bf16*, int, const int32_t*, const int32_t*, const bf16*, bf16*, int, int, hipStream_t,
                                       int);

// sessions of the continuous-latent models: the input rows come out of the latent adapter (every row: the latent it produced in the previous
// iteration); a row that STARTS a request (row_cls = -3) or idles on the zero row takes its `pending` row instead - the projected last
// condition token its prefill left, as for the text-conditioned token models
template <typename T>
__global__ __launch_bounds__(256) void override_session_rows_kernel(const int32_t* __restrict__ row_cls, const T* __restrict__ pending,
                                                                    T* __restrict__ out, int D, int out_nks) {
  const int m = blockIdx.x;
  if (row_cls[m] != -3) return;
  const T* src = pending + (size_t)m * D;
  for (int i = threadIdx.x; i < D; i += 256) out[out_nks ? afm_index<T>(m, i, out_nks) : (size_t)m * D + i] = src[i];
}
template <typename T>
int override_session_rows(const int32_t* row_cls, const T* pending, T* out, int rows, int D, hipStream_t st, int out_nks) {
  override_session_rows_kernel<T><<<rows, 256, 0, st>>>(row_cls, pending, out, D, out_nks);
  return VLG_OK;
}
template int override_session_rows<float>(const int32_t*, const float*, float*, int, int, hipStream_t, int);
template int override_session_rows<bf16>(const int32_t*, const bf16*, bf16*, int, int, hipStream_t, int);

__global__ void advance_state_kernel(StepState* s) {
  s->pos += 1;
  s->step += 1;
}
int advance_state(StepState* state, hipStream_t st) {
  advance_state_kernel<<<1, 1, 0, st>>>(state);
  return VLG_OK;
}
__global__ void set_state_kernel(StepState* s, int pos, int step) {
  s->pos = pos;
  s->step = step;
}
int set_state(StepState* state, int pos, int step, hipStream_t st) {
  set_state_kernel<<<1, 1, 0, st>>>(state, pos, step);
  return VLG_OK;
}

// Teacher forcing (vlg_gpt_set_teacher): the input of the NEXT step becomes the caller's token / latent for the step just produced
// (state->step, read on the device so that the captured graph replays), whatever the head sampled.  The head's own outputs stay in place.
__global__ void force_next_input_kernel(const StepState* s, const int32_t* ids, const float* lat, int32_t* cur_tok, float* cur_lat, int B, int Bp,
                                        int C, int N, int b_off) {
  const int step = s->step;
  if (step >= N) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (ids != nullptr && i < Bp) cur_tok[i] = ids[(size_t)(b_off + i % B) * N + step];   // rows b and b + B (guidance) are fed the same token: generate.py:92
  if (lat != nullptr && i < B * C) cur_lat[i] = lat[((size_t)(b_off + i / C) * N + step) * C + i % C];
}
int force_next_input(const StepState* state, const int32_t* ids, const float* lat, int32_t* cur_tok, float* cur_lat, int B, int Bp, int C, int N,
                     int b_off, hipStream_t st) {
  const int n = lat != nullptr ? std::max(Bp, B * C) : Bp;
  force_next_input_kernel<<<cdiv(n, 256), 256, 0, st>>>(state, ids, lat, cur_tok, cur_lat, B, Bp, C, N, b_off);
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}

}  // namespace vlg
