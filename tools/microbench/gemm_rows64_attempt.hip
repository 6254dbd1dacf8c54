// LAB ONLY - not part of libvlg.  Round 4 attempt at BASELINE config 5 (judge item 4): measured 41.1 us (w13) + 23.9 us (wqkv) per layer against 63.1 us for the
// slab GEMMs + their reduce launches; config 5 2.17 vs 2.11 s.  Kept as a record (profiles/r04_c5_rows64_attempt_kernel_stats.csv).
// It needs struct Rows64 / reduce_residual_rmsnorm(..., hn_nks) as they were in commit "gemm_rows64" to build.
//
// One-pass GEMMs of a decode step at 33..64 cache rows for the two WIDE Linear layers of a block - wqkv [3D, D] and [w1; w3] [2F, D] - with
// their epilogues inside (RoPE + q / KV-cache scatter, gpt.py:215-227,182-183; silu(a) * b, gpt.py:167).  BASELINE config 5: GPT-3B (D 3200,
// F 8704, head_dim 100), 32 images under guidance = 64 rows.
//
// Before (round 3): gemm_wide_kernel wrote fp32 split-K slabs [splits][64][N] and a second launch (qkv_rope_scatter / reduce_silu_mul) read
// them back - w13: 47 + 7 us per layer for 111 MB of weights, qkv: 25 + 7 us for 61 MB.  These two matrices have enough output columns to
// fill the chip WITHOUT splitting K (9600 / 48 = 200 and 8704 / 32 = 272 workgroups), so a workgroup can own its tile over the whole K range:
//   * 64 rows x NT 16-column tiles per workgroup of 8 waves; K steps of 64 bytes dealt to the waves round-robin, PD steps in flight per wave;
//   * A = the normalised activations xn, A-FRAGMENT-MAJOR (afm_index: the 16-row x 64-byte fragment of an MFMA is 1 KB contiguous - written
//     that way by reduce_residual_rmsnorm), B = the fragment-major weight copy: every wave instruction requests 8 whole cache lines;
//   * fp32 accumulators are summed over the 8 waves through LDS in wave order (fixed order: bit-reproducible), then the epilogue of the
//     launch chain runs on the sums - same rounding points as qkv_rope_scatter / reduce_silu_mul, fp32 summation order differs.
// The two N = D matrices (wo, w2: 200 tiles of 16 columns would make every workgroup pull all 64 x K activations) stay on the split-K slab
// GEMM, whose reduce launch also carries the residual add and the next RMSNorm.
#include <algorithm>

#include "gpt_kernels.h"

namespace vlg {

namespace {

typedef __bf16 r64_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float r64_f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned r64_u32x4_t __attribute__((ext_vector_type(4)));
typedef const r64_u32x4_t __attribute__((address_space(1))) * r64_gptr16;

constexpr int R64_NW = 8;

template <typename T>
__device__ __forceinline__ void r64_mfma(const r64_u32x4_t& a, const r64_u32x4_t& b, r64_f32x4_t& acc) {
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(r64_bf16x8_t, a), __builtin_bit_cast(r64_bf16x8_t, b), acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
  }
}
__device__ __forceinline__ float r64_silu(float x) { return x / (1.0f + expf(-x)); }

// EPI_QKV: N = 3 D, tiles are consecutive output columns.  EPI_SWIGLU: N = F; tiles 0 .. NT/2 - 1 are w1 rows n0 + 16 nt, tiles NT/2 .. NT - 1 the
// SAME columns of w3 (weight rows F + n0 + ...): the workgroup produces 16 NT / 2 columns of g.
// Workgroup bx owns `base` consecutive output n-tiles, the first `extra` workgroups one more (N / 16 tiles dealt over the grid as evenly as
// whole tiles allow: a compute unit streams ~24 GB/s from HBM whatever its neighbours do, so the launch takes as long as its heaviest
// workgroup - 8704 / 32 = 272 equal workgroups on 256 compute units would take two rounds).  NTH = most tiles a workgroup may own.
template <typename T, int NT, int EPI, int PD>
__global__ __launch_bounds__(64 * R64_NW) void gemm_rows64_kernel(const T* __restrict__ xa, const T* __restrict__ wfm, int M, int N, int K, Rows64 fa,
                                                                  int base, int extra) {
  constexpr int ESZ = (int)sizeof(T);
  constexpr int NTH = EPI == EPI_SWIGLU ? NT / 2 : NT;   // output n-tiles per workgroup, at most
  static_assert(EPI == EPI_QKV || (EPI == EPI_SWIGLU && NT % 2 == 0), "epilogues: RoPE + scatter, SwiGLU");
  extern __shared__ __attribute__((aligned(16))) char r64_smem[];
  float* red = reinterpret_cast<float*>(r64_smem);        // [wave pair][mt][nt][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nks = K * ESZ / 64;                            // K steps
  const int mtn = (M + 15) >> 4;                           // m-tiles in use (<= 4)
  const int bx = blockIdx.x;
  const int ntc = base + (bx < extra ? 1 : 0);                                   // output n-tiles of this workgroup (<= NTH)
  const int n0 = 16 * (bx < extra ? bx * (base + 1) : extra * (base + 1) + (bx - extra) * base);
  const int cnt = (nks - wave + R64_NW - 1) / R64_NW;      // K steps of this wave: wave + 8 i

  const char* ap = reinterpret_cast<const char*>(xa) + ((size_t)wave * 1024) + (size_t)lane * 16;        // + (mt * nks + 8 i) KB
  const char* bp[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int wrow = (EPI == EPI_SWIGLU && nt >= NTH) ? N + n0 + 16 * (nt - NTH) : n0 + 16 * nt;          // first weight row of the tile
    bp[nt] = reinterpret_cast<const char*>(wfm) + ((size_t)(wrow >> 4) * (size_t)nks + (size_t)wave) * 1024 + (size_t)lane * 16;
  }
  r64_f32x4_t acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = r64_f32x4_t{0.f, 0.f, 0.f, 0.f};
  r64_u32x4_t a[PD][4], b[PD][NT];
  auto load_step = [&](int slot, int i) __attribute__((always_inline)) {
    const size_t ko = (size_t)i * (R64_NW * 1024);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (mt < mtn) a[slot][mt] = *reinterpret_cast<r64_gptr16>((uintptr_t)(ap + (size_t)mt * (size_t)nks * 1024 + ko));       // L2-resident (every workgroup reads it)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      if (nt % NTH < ntc) b[slot][nt] = __builtin_nontemporal_load((r64_gptr16)(uintptr_t)(bp[nt] + ko));                         // read once
  };
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u < cnt) load_step(u, u);
  for (int i0 = 0; i0 < cnt; i0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int i = i0 + u;
      if (i < cnt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (nt % NTH < ntc) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
              if (mt < mtn) r64_mfma<T>(a[u][mt], b[u][nt], acc[mt][nt]);
          }
        if (i + PD < cnt) load_step(u, i + PD);
      }
    }
  }
  // ---- sum over the waves in a fixed order - (w, w + 4) pairs first, then the four pair sums - and epilogue.  Two stages so that the LDS
  // image is 4 x 4 x NT KB (96 KB at six weight tiles) instead of 8 x: waves 4..7 park their accumulators, waves 0..3 add theirs in place.
  if (wave >= 4) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(((wave - 4) * 4 + mt) * NT + nt) * 256 + e * 64 + lane] = acc[mt][nt][e];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[((wave * 4 + mt) * NT + nt) * 256 + e * 64 + lane] += acc[mt][nt][e];
  }
  __syncthreads();
  auto summed = [&](int mt, int nt, int v) __attribute__((always_inline)) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < R64_NW / 2; ++w) s += red[((w * 4 + mt) * NT + nt) * 256 + v];
    return s;
  };
  // one thread = one pair of adjacent output columns of one row: pair index -> (m-tile, n-tile, e, even lane)
  for (int pi = tid; pi < mtn * ntc * 128; pi += 64 * R64_NW) {
    const int tile = pi >> 7, pv = pi & 127;
    const int mt = tile / ntc, nt = tile - mt * ntc;
    const int e = pv >> 5, l = (pv & 31) * 2;
    const int row = mt * 16 + (l >> 4) * 4 + e;
    if (row >= M) continue;
    const int v = e * 64 + l;
    if constexpr (EPI == EPI_QKV) {
      const int col = n0 + nt * 16 + (l & 15);
      const int D = fa.H * fa.hd;
      const int sec = col / D, within = col - sec * D;
      const int hh = within / fa.hd, d = within - hh * fa.hd;
      const int p = fa.row_pos ? fa.row_pos[row] : fa.state->pos;
      const float x0 = DT<T>::rt(summed(mt, nt, v)), x1 = DT<T>::rt(summed(mt, nt, v + 1));
      float o0 = x0, o1 = x1;
      if (sec < 2) {   // gpt.py:423-433: adjacent pairs, fp32, then cast
        const float2 cs = *reinterpret_cast<const float2*>(fa.freqs + ((size_t)p * (fa.hd / 2) + d / 2) * 2);
        o0 = __fsub_rn(__fmul_rn(x0, cs.x), __fmul_rn(x1, cs.y));
        o1 = __fadd_rn(__fmul_rn(x1, cs.x), __fmul_rn(x0, cs.y));
      }
      T* dst;
      if (sec == 0)
        dst = reinterpret_cast<T*>(fa.qbuf) + ((size_t)row * fa.H + hh) * fa.hd + d;
      else
        dst = reinterpret_cast<T*>(sec == 1 ? fa.kc : fa.vc) + kv_row_index(fa.pages, row, hh, fa.H, fa.S, p) * fa.hd + d;
      DT<T>::st(dst, o0);
      DT<T>::st(dst + 1, o1);
    } else {
      const int col = n0 + nt * 16 + (l & 15);
      T* dst = reinterpret_cast<T*>(fa.out) + (size_t)row * N + col;
#pragma unroll
      for (int j = 0; j < 2; ++j) {   // g = rt(rt(silu(rt(a))) * rt(b))      (reduce_silu_mul)
        const float av = DT<T>::rt(summed(mt, nt, v + j)), bv = DT<T>::rt(summed(mt, nt + NTH, v + j));
        DT<T>::st(dst + j, DT<T>::rt(r64_silu(av)) * bv);
      }
    }
  }
}

constexpr int R64_NTH = 3;     // most output n-tiles per workgroup (48 columns of q | k | v, or 48 columns of g = 3 + 3 weight tiles)

// tiles dealt over the grid: workgroups = min(compute units, tiles) unless that would exceed R64_NTH tiles per workgroup
void r64_deal(int tiles, int cus, int* grid, int* base, int* extra) {
  int g = std::min(cus > 0 ? cus : 256, tiles);
  if ((tiles + g - 1) / g > R64_NTH) g = (tiles + R64_NTH - 1) / R64_NTH;
  *grid = g;
  *base = tiles / g;
  *extra = tiles - *base * g;
}

}  // namespace

template <typename T>
bool gemm_rows64_ok(int M, int D, int F, int H, int hd) {
  constexpr int ESZ = (int)sizeof(T);
  if (M <= 32 || M > 64 || M % 16 != 0) return false;                        // 33..64 rows in whole m-tiles (A-fragment-major rows are padded to 16)
  if (((size_t)D * ESZ) % 64 != 0 || H * hd != D || hd % 2 != 0) return false;
  return (3 * D) % 16 == 0 && F % 16 == 0;                                   // a column pair never straddles a head or a q | k | v section (hd even)
}
template bool gemm_rows64_ok<float>(int, int, int, int, int);
template bool gemm_rows64_ok<bf16>(int, int, int, int, int);

template <typename T>
int gemm_rows64(const T* xa, const T* wfm, int M, int N, int K, int epi, const Rows64& fa, hipStream_t st) {
  VLG_CHECK(xa && wfm && (epi == EPI_QKV || epi == EPI_SWIGLU), VLG_ERR_BAD_ARG, "gemm_rows64: bad argument");
  static int cus_cache[64] = {};
  int dev = 0, cus = 0;
  VLG_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && cus_cache[dev] > 0)
    cus = cus_cache[dev];
  else {
    VLG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cus_cache[dev] = cus;
  }
  int grid = 0, base = 0, extra = 0;
  r64_deal(N / 16, cus, &grid, &base, &extra);
  if (epi == EPI_QKV) {
    auto kern = gemm_rows64_kernel<T, R64_NTH, EPI_QKV, 4>;
    const int lds = (R64_NW / 2) * 4 * R64_NTH * 1024;
    static LdsAttrOnce once;
    VLG_TRY(set_max_dynamic_lds(once, {reinterpret_cast<const void*>(kern)}, lds));
    kern<<<grid, 64 * R64_NW, lds, st>>>(xa, wfm, M, N, K, fa, base, extra);
  } else {
    auto kern = gemm_rows64_kernel<T, 2 * R64_NTH, EPI_SWIGLU, 3>;
    const int lds = (R64_NW / 2) * 4 * 2 * R64_NTH * 1024;
    static LdsAttrOnce once;
    VLG_TRY(set_max_dynamic_lds(once, {reinterpret_cast<const void*>(kern)}, lds));
    kern<<<grid, 64 * R64_NW, lds, st>>>(xa, wfm, M, N, K, fa, base, extra);   // N = F here
  }
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}
template int gemm_rows64<float>(const float*, const float*, int, int, int, int, const Rows64&, hipStream_t);
template int gemm_rows64<bf16>(const bf16*, const bf16*, int, int, int, int, const Rows64&, hipStream_t);

}  // namespace vlg
