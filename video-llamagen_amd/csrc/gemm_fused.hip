// Fused skinny GEMM for the decode step: y = epilogue( prologue(x) @ w^T ), one workgroup per 16-column n-tile, no inter-workgroup
// split-K (so the epilogue can finish the op in place) - removes the RMSNorm, RoPE/KV-scatter, SiLU*mul and residual kernels
// and all fp32 slab traffic from the decode layer (9 -> 6 launches per layer).
//
//   prologue  PRO_NORM : x is the residual stream h [M,K]; every workgroup holds the full K range of its rows in registers
//                        (it needs them as the A operand anyway), so it computes sum(h^2) itself (wave partials -> LDS), then
//                        normalises its fragments in registers: rt(rt(h * rsqrt(mean + eps)) * g)   (gpt.py:143-148, exact
//                        rounding order of the reference).  No extra memory traffic.
//   epilogues EPI_RESID  h[row][col] = rt(h + rt(acc))                                    (gpt.py:257-258)
//             EPI_QKV    RoPE on adjacent pairs (partner column sits in the neighbouring thread's LDS slot) + scatter to the
//                        q buffer / KV cache at position p                                    (gpt.py:215-227,182-183)
//             EPI_SWIGLU n-tile taken from both halves of [w1; w3]: g = rt(rt(silu(rt(a))) * rt(b))   (gpt.py:167)
//             EPI_STORE  out = rt(act(rt(acc + bias))) (T and/or fp32)                        (heads, adapters)
//             EPI_GATED  h[row][col] = rt(h + rt(gate[row][col] * rt(acc + bias)))           (DiffLoss ResBlock, diffloss.py:128)
#include "gpt_kernels.h"

#ifdef VLG_KTRACE   // tools/microbench: in-kernel timestamps (100 MHz), thread 0 of every workgroup
#define VLG_KT(i)                                                                                                        \
  do {                                                                                                                   \
    if (threadIdx.x == 0 && fa.trace) {                                                                                  \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                        \
      fa.trace[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (i)] = wall_clock64();    \
    }                                                                                                                    \
  } while (0)
#else
#define VLG_KT(i) \
  do {            \
  } while (0)
#endif

namespace vlg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float silu_g(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_g(float x) {
  const float k = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}

__device__ __forceinline__ uint32_t pack2(float a, float b) {   // two RNE bf16 in one v_cvt_pk_bf16_f32
  const bf16x2_t v = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo16(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

template <typename T>
struct KB;
template <>
struct KB<bf16> { static constexpr int KBLK = 128; };
template <>
struct KB<float> { static constexpr int KBLK = 64; };

template <typename T>
__device__ __forceinline__ float sumsq(const u32x4_t& v) {
  float s = 0.f;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = lo16(v[j]), b = hi16(v[j]);
      s += a * a + b * b;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = __uint_as_float(v[j]);
      s += a * a;
    }
  }
  return s;
}

// v <- rt(rt(v * rs) * g)
template <typename T>
__device__ __forceinline__ void norm_frag(u32x4_t& v, float rs, const u32x4_t& g) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t n = pack2(lo16(v[j]) * rs, hi16(v[j]) * rs);
      v[j] = pack2(lo16(n) * lo16(g[j]), hi16(n) * hi16(g[j]));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __float_as_uint(__uint_as_float(v[j]) * rs * __uint_as_float(g[j]));
  }
}

template <typename T, int MT>
__device__ __forceinline__ void mfma_blk(const u32x4_t (&a)[MT][4], const u32x4_t (&b)[4], f32x4_t (&acc)[MT]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8_t vb = __builtin_bit_cast(bf16x8_t, b[s]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[mt][s]), vb, acc[mt], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[mt][s][e]), __uint_as_float(b[s][e]), acc[mt], 0, 0, 0);
  }
}

// one 16-byte fragment pair: bf16 = one 16x16x32 step; fp32 = four 16x16x4 steps
template <typename T>
__device__ __forceinline__ void mfma_one(const u32x4_t& a, const u32x4_t& b, f32x4_t& acc) {
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
  }
}

// NC: output columns per workgroup.  8 (EPI_RESID at M <= 16 only): lanes 8..15 of a 16-lane group mirror lanes 0..7 (same weight row,
// same column, results dropped) - half of the MFMA tile is wasted, but twice the workgroups stream the two N = D matrices, whose 80
// 16-column tiles leave 2/3 of the CUs idle while each busy one is bound by its own miss-handling rate.
// NT: 16-column n-tiles per workgroup.  2 (with MT = 1 and the two 16-row halves of the batch as grid rows) for the kernels with the RMSNorm
// prologue at 17..32 rows: every workgroup normalises ALL activation elements of its rows (it needs the whole K range), and that VALU work
// is 1.5 us of a QKV / w13 launch - a workgroup of 16 rows x 32 columns does half of it and reads half the
// activation bytes for the same number of workgroups (the second reader of a weight tile hits its XCD's L2, as for wo / w2).
template <typename T, int MT, int NW, bool PRO, int EPI, int NC = 16, int NT = 1>
__global__ __launch_bounds__(64 * NW) void gemm_fused_kernel(const T* __restrict__ x, const T* __restrict__ w, int M, int N, int K, FusedGemm fa) {
  static_assert(NC == 16 || (NC == 8 && EPI == EPI_RESID && !PRO), "8-column tiles: residual epilogue only");
  static_assert(NT == 1 || (NT == 2 && NC == 16 && MT == 1), "two n-tiles: 16-row workgroups of 16-column tiles");
  constexpr int KBLK = KB<T>::KBLK;
  constexpr int NHH = (EPI == EPI_SWIGLU) ? 2 : 1;   // weight tiles per output tile (w1 and w3 rows of the same columns)
  constexpr int NH = NHH * NT;                        // weight tiles per workgroup: index nt * NHH + half
  constexpr int NB = (MT * NH >= 4) ? 2 : 4;
  constexpr int EPV = 16 / sizeof(T);   // elements per 16-byte vector
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.y == 2 && (gridDim.x & 7) == 0) {
    // the two row halves of one n-tile stream the same weight rows: put them on the same XCD (workgroups are dealt to the 8 XCDs
    // round-robin in linear order) so the second reader hits that XCD's L2.  Speed only.
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    bx = (id & 7) + 8 * (id >> 4);
    by = (id >> 3) & 1;
  }
  const int n0 = bx * (NC * NT), m0 = by * (MT * 16);
  const int nkb = K / KBLK;
  VLG_KT(0);

  __shared__ float red[NW][NH][MT][256];
  __shared__ float ssq_sm[NW][MT * 16];
  __shared__ u32x4_t gsm[PRO ? 256 : 1];   // norm weight, K * sizeof(T) <= 4 KiB

  f32x4_t acc[NH][MT];
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[hf][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Weight fragment (K block kb, chunk s2) of this lane: wbase + kb * w_kstep + s2 * w_sstep (16-byte units).  Row-major [N][K]: the lane's
  // row, chunk kb * 16 + 4 s2 + q.  Fragment-major copy (fa.wfm, gpt_kernels.h): block (tile, K step 4 kb + s2), slot row-in-tile + 16 q -
  // a wave instruction then reads 1 KB of whole cache lines instead of 16 half lines (1.5 x the cold stream rate).
  const u32x4_t* wbase[NH];
  const bool fm = fa.wfm != nullptr;
  const int w_sstep = fm ? 64 : 4, w_kstep = fm ? 256 : (int)(KBLK * sizeof(T) / 16);
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
    const int ncol = n0 + (hf / NHH) * 16 + (hf % NHH) * N;
    if (fm)
      wbase[hf] = reinterpret_cast<const u32x4_t*>(fa.wfm) + (size_t)(ncol >> 4) * ((size_t)nkb * 256) + ((ncol & 15) + (r & (NC - 1))) + 16 * q;
    else
      wbase[hf] = reinterpret_cast<const u32x4_t*>(w + (size_t)(ncol + (r & (NC - 1))) * K) + q;
  }
  // A fragment (K block kb, chunk s2) of m-tile mt: abase[mt] + kb * a_kstep + s2 * a_sstep (16-byte units); row-major rows, or the
  // A-fragment-major matrix (fa.a_fm, gpt_kernels.h afm_index): block (m-tile, K step 4 kb + s2), slot = lane
  const u32x4_t* abase[MT];
  const bool afm = fa.a_fm != 0, ofm = fa.o_fm != 0;
  const int a_sstep = afm ? 64 : 4, a_kstep = afm ? 256 : (int)(KBLK * sizeof(T) / 16);
  const int o_nks = N * (int)sizeof(T) / 64;   // K steps per row of an A-fragment-major [M, N] result
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = m0 + mt * 16 + r;
    row = row < M ? row : M - 1;
    if (afm)
      abase[mt] = reinterpret_cast<const u32x4_t*>(x) + (size_t)((m0 >> 4) + mt) * ((size_t)nkb * 256) + lane;
    else
      abase[mt] = reinterpret_cast<const u32x4_t*>(x + (size_t)row * K) + q;
  }
  auto oidx = [&](int row, int c) -> size_t { return ofm ? afm_index<T>(row, c, o_nks) : (size_t)row * N + c; };
  // Request order = arrival order (vmcnt retires in order): activations (L2) first, then the norm weight, then the weight stream
  // (HBM), then the epilogue operands that do not depend on the GEMM (residual rows; RoPE pair of the current position).  The
  // norm prologue then runs on the activations while the weights are still in flight, and the epilogue never waits on memory.
  const int et = threadIdx.x, ee = et >> 6, el = et & 63;
  const int ecol0 = n0 + (el & (NC - 1));   // output column of n-tile 0; n-tile nt: + 16 nt
  float eres[NT][MT], ecx[NT][MT], ecy[NT][MT];   // ecx doubles as the gate value for EPI_GATED
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) eres[nt][mt] = 0.f, ecx[nt][mt] = 1.f, ecy[nt][mt] = 0.f;
  int epos = 0;
  const int32_t* erow_pos = nullptr;   // iteration-level batching: per-row positions (a kernel argument, no dependent load)
  if constexpr (EPI == EPI_QKV) {
    epos = fa.state->pos;   // scalar load
    erow_pos = fa.row_pos;
  }
  auto epilogue_operands = [&]() {
    if constexpr (EPI == EPI_RESID || EPI == EPI_GATED) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          int row = m0 + mt * 16 + (el >> 4) * 4 + ee;
          row = row < M ? row : M - 1;
          eres[nt][mt] = DT<T>::ld(reinterpret_cast<const T*>(fa.h) + oidx(row, ecol0 + nt * 16));
          if constexpr (EPI == EPI_GATED) ecx[nt][mt] = DT<T>::ld(reinterpret_cast<const T*>(fa.gate) + (size_t)row * fa.gate_stride + ecol0 + nt * 16);
        }
    }
    if constexpr (EPI == EPI_QKV) {
      const int D = fa.H * fa.hd;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int d = ((ecol0 + nt * 16) % D) % fa.hd;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          int row = m0 + mt * 16 + (el >> 4) * 4 + ee;
          row = row < M ? row : M - 1;
          // uniform position: sessions (per-row positions) reload their pair in the epilogue - a per-row load here would put a
          // branch and a dependent load in front of the weight stream
          const float* cp = fa.freqs + ((size_t)(epos + row % fa.Tq) * (fa.hd / 2) + d / 2) * 2;
          ecx[nt][mt] = cp[0];
          ecy[nt][mt] = cp[1];
        }
      }
    }
  };

  u32x4_t a[NB][MT][4], b[NB][NH][4];
  u32x4_t gld = u32x4_t{0u, 0u, 0u, 0u};
  bool first = true;
  const int kstride = NW;
  for (int kb0 = wave; kb0 < nkb; kb0 += NB * kstride) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kb = kb0 + i * kstride;
      if (kb < nkb) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const u32x4_t* px = abase[mt] + (size_t)kb * a_kstep;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) a[i][mt][s2] = px[s2 * a_sstep];
        }
      }
    }
    if constexpr (PRO) {
      const int gi = (int)threadIdx.x < K / EPV ? (int)threadIdx.x : 0;
      gld = reinterpret_cast<const u32x4_t*>(fa.norm_w)[gi];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kb = kb0 + i * kstride;
      if (kb < nkb) {
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
          const u32x4_t* pw = wbase[hf] + (size_t)kb * w_kstep;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) b[i][hf][s2] = __builtin_nontemporal_load(pw + s2 * w_sstep);
        }
      }
    }
    if (first) {
      epilogue_operands();
      first = false;
    }
    if constexpr (PRO) {
      if ((int)threadIdx.x < K / EPV) gsm[threadIdx.x] = gld;   // owners sit in waves < K / (4 * KBLK) <= nkb: always in the loop
    }
    if constexpr (PRO) {
      // host guarantees nkb <= NB * NW: this loop body runs once per wave and holds all of the rows' K range
      float part[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i)
          if (kb0 + i * kstride < nkb) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) p += sumsq<T>(a[i][mt][s2]);
          }
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        part[mt] = p;
      }
      if (q == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ssq_sm[wave][mt * 16 + r] = part[mt];
      }
    }
    if constexpr (PRO) {
      // waves without K blocks never enter this loop: they publish zeros below, outside it
    }
    if constexpr (!PRO) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (kb0 + i * kstride < nkb) {
#pragma unroll
          for (int hf = 0; hf < NH; ++hf) mfma_blk<T, MT>(a[i], b[i][hf], acc[hf]);
        }
    }
  }
  if (first) epilogue_operands();
  if constexpr (PRO) {
    if (wave >= nkb && q == 0) {   // this wave owns no K block
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) ssq_sm[wave][mt * 16 + r] = 0.f;
    }
    __syncthreads();
    if (wave < nkb) {
      float rs[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float tot = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) tot += ssq_sm[wv][mt * 16 + r];
        rs[mt] = 1.0f / sqrtf(tot / (float)K + fa.eps);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int kb = wave + i * NW;
        if (kb < nkb) {
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const u32x4_t g = gsm[kb * (KBLK / EPV) + s2 * 4 + q];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) norm_frag<T>(a[i][mt][s2], rs[mt], g);
          }
#pragma unroll
          for (int hf = 0; hf < NH; ++hf) mfma_blk<T, MT>(a[i], b[i][hf], acc[hf]);
        }
      }
    }
  }

  VLG_KT(1);
  // ---- cross-wave reduction ----
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][hf][mt][e * 64 + lane] = acc[hf][mt][e];
  __syncthreads();
  VLG_KT(2);
  const int t = threadIdx.x;
  const int e = t >> 6, l2 = t & 63;
  if (t >= 256) return;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int col = n0 + nt * 16 + (l2 & (NC - 1));
    const int row = m0 + mt * 16 + (l2 >> 4) * 4 + e;
    float s0 = 0.f, s1 = 0.f, sp = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) {
      s0 += red[wv][nt * NHH][mt][t];
      if constexpr (NHH == 2) s1 += red[wv][nt * NHH + 1][mt][t];
      if constexpr (EPI == EPI_QKV) sp += red[wv][nt * NHH][mt][t ^ 1];
    }
    if (row >= M) continue;
    if (NC == 8 && (l2 & 8)) continue;   // mirror lanes
    if constexpr (EPI == EPI_RESID) {
      DT<T>::st(reinterpret_cast<T*>(fa.h) + oidx(row, col), eres[nt][mt] + DT<T>::rt(s0));
    } else if constexpr (EPI == EPI_GATED) {
      float v = s0;
      if (fa.bias) v += DT<T>::ld(reinterpret_cast<const T*>(fa.bias) + col);
      DT<T>::st(reinterpret_cast<T*>(fa.h) + oidx(row, col), eres[nt][mt] + DT<T>::rt(ecx[nt][mt] * DT<T>::rt(v)));
    } else if constexpr (EPI == EPI_SWIGLU) {
      const float av = DT<T>::rt(s0), bv = DT<T>::rt(s1);
      DT<T>::st(reinterpret_cast<T*>(fa.out) + oidx(row, col), DT<T>::rt(silu_g(av)) * bv);
    } else if constexpr (EPI == EPI_STORE) {
      float v = s0;
      if (fa.bias) v += DT<T>::ld(reinterpret_cast<const T*>(fa.bias) + col);
      v = DT<T>::rt(v);
      if (fa.act == ACT_GELU_TANH) v = DT<T>::rt(gelu_g(v));
      if (fa.act == ACT_SILU) v = DT<T>::rt(silu_g(v));
      if (fa.out) DT<T>::st(reinterpret_cast<T*>(fa.out) + oidx(row, col), v);
      if (fa.out_f32) fa.out_f32[(size_t)row * N + col] = v;
    } else {  // EPI_QKV
      const int D = fa.H * fa.hd;
      const int sec = col / D, within = col - sec * D;
      const int hh = within / fa.hd, d = within - hh * fa.hd;
      const int bq = row / fa.Tq, tq = row - bq * fa.Tq;
      const int p = (erow_pos ? erow_pos[bq] : epos) + tq;
      const float xs = DT<T>::rt(s0), xp = DT<T>::rt(sp);
      float o = xs;
      if (sec < 2) {
        float cx = ecx[nt][mt], cy = ecy[nt][mt];
        if (erow_pos) {
          const float* cp = fa.freqs + ((size_t)p * (fa.hd / 2) + d / 2) * 2;
          cx = cp[0];
          cy = cp[1];
        }
        o = (d & 1) ? __fadd_rn(__fmul_rn(xs, cx), __fmul_rn(xp, cy))    // x1*c + x0*s
                    : __fsub_rn(__fmul_rn(xs, cx), __fmul_rn(xp, cy));   // x0*c - x1*s
      }
      T* dst;
      if (sec == 0)
        dst = reinterpret_cast<T*>(fa.qbuf) + ((size_t)row * fa.H + hh) * fa.hd + d;
      else
        dst = reinterpret_cast<T*>(sec == 1 ? fa.kc : fa.vc) + kv_row_index(fa.pages, bq, hh, fa.H, fa.S, p) * fa.hd + d;
      DT<T>::st(dst, o);
    }
  }
  VLG_KT(3);
}

// ---- LayerNorm + modulate prologue, store epilogue: 16 rows x 16 columns per workgroup, 4 waves, <= 2 K blocks per wave ----------
template <typename T>
__device__ __forceinline__ float frag_sum(const u32x4_t& v) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (sizeof(T) == 2)
      s += lo16(v[j]) + hi16(v[j]);
    else
      s += __uint_as_float(v[j]);
  }
  return s;
}
template <typename T>
__device__ __forceinline__ float frag_sqdev(const u32x4_t& v, float mu) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (sizeof(T) == 2) {
      const float a = lo16(v[j]) - mu, b = hi16(v[j]) - mu;
      s += a * a + b * b;
    } else {
      const float a = __uint_as_float(v[j]) - mu;
      s += a * a;
    }
  }
  return s;
}
// v <- rt(rt(((v - mu) * rs) [* gw + gb]) * (1 + sc) + sh), element order as dl_ln_modulate_kernel
template <typename T>
__device__ __forceinline__ void ln_frag(u32x4_t& v, float mu, float rs, bool affine, const u32x4_t& gw, const u32x4_t& gb, const u32x4_t& sc,
                                        const u32x4_t& sh) {
  auto one = [&](float x, float w_, float b_, float c_, float h_) {
    float n = (x - mu) * rs;
    if (affine) n = n * w_ + b_;
    n = DT<T>::rt(n);
    return n * (1.0f + c_) + h_;
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (sizeof(T) == 2)
      v[j] = pack2(one(lo16(v[j]), lo16(gw[j]), lo16(gb[j]), lo16(sc[j]), lo16(sh[j])),
                   one(hi16(v[j]), hi16(gw[j]), hi16(gb[j]), hi16(sc[j]), hi16(sh[j])));
    else
      v[j] = __float_as_uint(one(__uint_as_float(v[j]), __uint_as_float(gw[j]), __uint_as_float(gb[j]), __uint_as_float(sc[j]),
                                 __uint_as_float(sh[j])));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_ln_kernel(const T* __restrict__ x, const T* __restrict__ w, int M, int N, int K, LnGemm fa) {
  constexpr int KBLK = KB<T>::KBLK, NW = 4, NB = 2;
  constexpr int EPV = 16 / sizeof(T);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y;
  if (gridDim.y == 2 && (gridDim.x & 7) == 0) {   // row halves of one n-tile on one XCD (see gemm_fused_kernel)
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    bx = (id & 7) + 8 * (id >> 4);
    by = (id >> 3) & 1;
  }
  const int n0 = bx * 16, m0 = by * 16;
  const int nkb = K / KBLK;
  const bool affine = fa.ln_w != nullptr;

  __shared__ float red[NW][256];
  __shared__ float stat[2][NW][16];
  __shared__ u32x4_t gsm[2][256];   // ln weight / bias, K * sizeof(T) <= 4 KiB each

  int row = m0 + r;
  row = row < M ? row : M - 1;
  const T* xrow = x + (size_t)row * K;
  const T* shrow = reinterpret_cast<const T*>(fa.shift) + (size_t)row * fa.mod_stride;
  const T* scrow = reinterpret_cast<const T*>(fa.scale) + (size_t)row * fa.mod_stride;
  const T* wrow = w + (size_t)(n0 + r) * K;

  // request order = arrival order: activations, modulation rows, LN affine, weight stream, bias
  u32x4_t a[NB][1][4], sh[NB][4], sc[NB][4], b[NB][4];
  u32x4_t gw = u32x4_t{0u, 0u, 0u, 0u}, gb = gw;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int kb = wave + i * NW;
    if (kb < nkb) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) a[i][0][s2] = (reinterpret_cast<const u32x4_t*>(xrow + (size_t)kb * KBLK) + q)[s2 * 4];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) sc[i][s2] = (reinterpret_cast<const u32x4_t*>(scrow + (size_t)kb * KBLK) + q)[s2 * 4];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) sh[i][s2] = (reinterpret_cast<const u32x4_t*>(shrow + (size_t)kb * KBLK) + q)[s2 * 4];
    }
  }
  const int gi = (int)threadIdx.x < K / EPV ? (int)threadIdx.x : 0;
  if (affine) {
    gw = reinterpret_cast<const u32x4_t*>(fa.ln_w)[gi];
    gb = reinterpret_cast<const u32x4_t*>(fa.ln_b)[gi];
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int kb = wave + i * NW;
    if (kb < nkb) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) b[i][s2] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wrow + (size_t)kb * KBLK) + q + s2 * 4);
    }
  }
  const int t = threadIdx.x, e = t >> 6, l2 = t & 63;
  const int col = n0 + (l2 & 15);
  const float bv = fa.bias ? DT<T>::ld(reinterpret_cast<const T*>(fa.bias) + col) : 0.f;

  // LayerNorm statistics, two passes as the reference kernel (mean, then the mean of squared deviations)
  float p = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (wave + i * NW < nkb) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) p += frag_sum<T>(a[i][0][s2]);
    }
  p += __shfl_xor(p, 16);
  p += __shfl_xor(p, 32);
  if (q == 0) stat[0][wave][r] = p;
  if ((int)threadIdx.x < K / EPV) {
    gsm[0][threadIdx.x] = gw;
    gsm[1][threadIdx.x] = gb;
  }
  __syncthreads();
  const float mu = (stat[0][0][r] + stat[0][1][r] + stat[0][2][r] + stat[0][3][r]) / (float)K;
  float d2 = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (wave + i * NW < nkb) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) d2 += frag_sqdev<T>(a[i][0][s2], mu);
    }
  d2 += __shfl_xor(d2, 16);
  d2 += __shfl_xor(d2, 32);
  if (q == 0) stat[1][wave][r] = d2;
  __syncthreads();
  const float rs = 1.0f / sqrtf((stat[1][0][r] + stat[1][1][r] + stat[1][2][r] + stat[1][3][r]) / (float)K + fa.eps);

  f32x4_t acc[1] = {f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int kb = wave + i * NW;
    if (kb < nkb) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int gidx = kb * (KBLK / EPV) + s2 * 4 + q;
        ln_frag<T>(a[i][0][s2], mu, rs, affine, gsm[0][gidx], gsm[1][gidx], sc[i][s2], sh[i][s2]);
      }
      mfma_blk<T, 1>(a[i], b[i], acc);
    }
  }
#pragma unroll
  for (int ee = 0; ee < 4; ++ee) red[wave][ee * 64 + lane] = acc[0][ee];
  __syncthreads();
  const int orow = m0 + (l2 >> 4) * 4 + e;
  if (orow >= M) return;
  float v = red[0][t] + red[1][t] + red[2][t] + red[3][t] + bv;
  v = DT<T>::rt(v);
  if (fa.act == ACT_GELU_TANH) v = DT<T>::rt(gelu_g(v));
  if (fa.act == ACT_SILU) v = DT<T>::rt(silu_g(v));
  DT<T>::st(reinterpret_cast<T*>(fa.out) + (size_t)orow * N + col, v);
}

template <typename T, int MT, int NW, bool PRO>
void launch_epi(int epi, dim3 grid, hipStream_t st, const T* x, const T* w, int M, int N, int K, const FusedGemm& fa) {
  switch (epi) {
    case EPI_RESID: gemm_fused_kernel<T, MT, NW, PRO, EPI_RESID><<<grid, 64 * NW, 0, st>>>(x, w, M, N, K, fa); break;
    case EPI_QKV: gemm_fused_kernel<T, MT, NW, PRO, EPI_QKV><<<grid, 64 * NW, 0, st>>>(x, w, M, N, K, fa); break;
    case EPI_SWIGLU: gemm_fused_kernel<T, MT, NW, PRO, EPI_SWIGLU><<<grid, 64 * NW, 0, st>>>(x, w, M, N, K, fa); break;
    case EPI_GATED: gemm_fused_kernel<T, MT, NW, PRO, EPI_GATED><<<grid, 64 * NW, 0, st>>>(x, w, M, N, K, fa); break;
    default: gemm_fused_kernel<T, MT, NW, PRO, EPI_STORE><<<grid, 64 * NW, 0, st>>>(x, w, M, N, K, fa); break;
  }
}

int lds_knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

// true if the fused kernel covers this shape (otherwise the caller uses the slab GEMM + separate epilogue kernels)
template <typename T>
bool gemm_fused_ok(int M, int N, int K, bool pro, int epi) {
  constexpr int KBLK = KB<T>::KBLK;
  if (M < 1 || K % KBLK != 0 || N % 16 != 0) return false;
  if (!pro) return true;
  const int mt = M > 32 ? 4 : (M > 16 ? 2 : 1);
  const int nh = epi == EPI_SWIGLU ? 2 : 1;
  const int nb = (mt * nh >= 4) ? 2 : 4;
  const int nw = (nb == 2) ? 8 : 4;
  return K / KBLK <= nb * nw && (size_t)K * sizeof(T) <= 4096;
}
template bool gemm_fused_ok<float>(int, int, int, bool, int);
template bool gemm_fused_ok<bf16>(int, int, int, bool, int);

template <typename T>
int gemm_fused(const T* x, const T* w, int M, int N, int K, bool pro, int epi, const FusedGemm& fa, hipStream_t st) {
  if (!gemm_fused_ok<T>(M, N, K, pro, epi)) {
    set_error("gemm_fused: shape M=%d N=%d K=%d pro=%d epi=%d not covered", M, N, K, (int)pro, epi);
    return VLG_ERR_UNSUPPORTED;
  }
  int mt = M > 32 ? 4 : (M > 16 ? 2 : 1);
  const int nh = epi == EPI_SWIGLU ? 2 : 1;
  const int nkb_all = K / KB<T>::KBLK;
  // few n-tiles (N = D: wo, w2): 16-row workgroups, so twice the CUs stream and each pulls half the activations through its
  // vector-memory pipe (the row halves of a tile share an XCD, see the kernel); 8 waves when K needs more than 16 K-block slots
  const bool rows16 = !pro && (epi == EPI_RESID || epi == EPI_GATED || epi == EPI_STORE) && mt == 2 && N / 16 <= 128 && (N / 16) % 8 == 0;
  if (rows16) mt = 1;
  const bool wide = (mt * nh >= 4) || (rows16 && nkb_all > 16);   // 8 waves so one pass of the K loop covers the slice
  dim3 grid(N / 16, cdiv(M, mt * 16));
  // small batches (M <= 16: the per-GPU shards of a batch split over GPUs): the two N = D GEMMs on 8-column tiles, twice the workgroups
  static const int nc8_knob = lds_knob("VLG_GEMM_NC8", 1);
  if (nc8_knob && !pro && epi == EPI_RESID && M <= 16 && N / 16 <= 128 && N % 8 == 0) {
    grid = dim3(N / 8, 1, 1);
    if (nkb_all > 16)
      gemm_fused_kernel<T, 1, 8, false, EPI_RESID, 8><<<grid, 512, 0, st>>>(x, w, M, N, K, fa);
    else
      gemm_fused_kernel<T, 1, 4, false, EPI_RESID, 8><<<grid, 256, 0, st>>>(x, w, M, N, K, fa);
    return VLG_OK;
  }
  // RMSNorm-prologue kernels at 17..32 rows: 16 rows x two n-tiles per workgroup (NT = 2), the two row halves as grid rows paired on an XCD:
  // half the normalisation work and half the activation bytes per workgroup, the same number of workgroups
  static const int nt2_knob = lds_knob("VLG_GEMM_NT2", 1);
  if (nt2_knob && pro && mt == 2 && (epi == EPI_QKV || epi == EPI_SWIGLU || epi == EPI_STORE) && N % 32 == 0 && (N / 32) % 8 == 0) {
    const dim3 g2(N / 32, cdiv(M, 16));
    if (epi == EPI_SWIGLU)
      gemm_fused_kernel<T, 1, 8, true, EPI_SWIGLU, 16, 2><<<g2, 512, 0, st>>>(x, w, M, N, K, fa);
    else if (epi == EPI_QKV)   // (round 4: 8 waves per QKV workgroup - half the normalisation arithmetic per wave - measured 1824 vs 1826 ms per 1024-token step: not kept)
      gemm_fused_kernel<T, 1, 4, true, EPI_QKV, 16, 2><<<g2, 256, 0, st>>>(x, w, M, N, K, fa);
    else
      gemm_fused_kernel<T, 1, 4, true, EPI_STORE, 16, 2><<<g2, 256, 0, st>>>(x, w, M, N, K, fa);
    return VLG_OK;
  }
#define VLG_GF(MT_, NW_)                                                            \
  do {                                                                              \
    if (pro)                                                                        \
      launch_epi<T, MT_, NW_, true>(epi, grid, st, x, w, M, N, K, fa);              \
    else                                                                            \
      launch_epi<T, MT_, NW_, false>(epi, grid, st, x, w, M, N, K, fa);             \
  } while (0)
  if (mt == 4)
    VLG_GF(4, 8);
  else if (mt == 2) {
    if (wide)
      VLG_GF(2, 8);
    else
      VLG_GF(2, 4);
  } else {
    if (wide)
      VLG_GF(1, 8);
    else
      VLG_GF(1, 4);
  }
#undef VLG_GF
  return VLG_OK;
}
template <typename T>
bool gemm_ln_fused_ok(int M, int N, int K) {
  constexpr int KBLK = KB<T>::KBLK;
  return M >= 1 && N % 16 == 0 && K % KBLK == 0 && K / KBLK <= 8 && (size_t)K * sizeof(T) <= 4096;
}
template bool gemm_ln_fused_ok<float>(int, int, int);
template bool gemm_ln_fused_ok<bf16>(int, int, int);

template <typename T>
int gemm_ln_fused(const T* x, const T* w, int M, int N, int K, const LnGemm& fa, hipStream_t st) {
  if (!gemm_ln_fused_ok<T>(M, N, K)) {
    set_error("gemm_ln_fused: shape M=%d N=%d K=%d not covered", M, N, K);
    return VLG_ERR_UNSUPPORTED;
  }
  gemm_ln_kernel<T><<<dim3(N / 16, cdiv(M, 16)), 256, 0, st>>>(x, w, M, N, K, fa);
  return VLG_OK;
}
template int gemm_ln_fused<float>(const float*, const float*, int, int, int, const LnGemm&, hipStream_t);
template int gemm_ln_fused<bf16>(const bf16*, const bf16*, int, int, int, const LnGemm&, hipStream_t);

template int gemm_fused<float>(const float*, const float*, int, int, int, bool, int, const FusedGemm&, hipStream_t);
template int gemm_fused<bf16>(const bf16*, const bf16*, int, int, int, bool, int, const FusedGemm&, hipStream_t);

}  // namespace vlg
