// LAB ONLY (tools/microbench/pd_lab.hip includes this file; it is not part of libvlg): the six-hand-off form of the persistent decode step,
// built and measured in round 4 and NOT adopted - parity-green (reference goldens bit-exact in fp32, launch chain within bf16 summation
// noise at GPT-XL width, 1...16 rows) but slower than pdecode.hip at every shape measured: GPT-XL 4 rows at position 2680 37.6 vs 34.9 us per
// layer, 8 rows at 600 38.8 vs 31.4, 1 row 31.1 vs 28.5, GPT-L 16 rows 37.9 vs 30.8 (profiles/r04_pd_lab_stamps.txt).  Two hand-offs fewer
// per layer saved ~6 us of waiting; the fp32 partial rows that replace them cost ~9: 224 F slices x M x D granules published (3.3 us) and
// summed back (5.5 us) per layer, 20 heads x M x D on the attention side (3.7 us).  DESIGN.md section 5, round 4.
//
// Persistent decode step, second form (round 4): the L transformer layers of ONE decode step (Tq = 1, gpt.py:255-259 x n_layer) in ONE launch,
// with SIX dependent hand-offs per layer instead of the eight of pdecode.hip.
//
// A hand-off between workgroups costs 2.2-4 us inside a launch (store -> fabric -> poll, queued behind the polling compute unit's own weight
// requests: DESIGN.md section 5, MI355X_MICROARCH.md "handoff" / "allgather" rows) and the eight of pdecode.hip were 22 of its 36 us per
// layer.  Here the two GEMM pairs of a layer are paired the way a tensor-parallel layer pairs them - column-parallel producer, row-parallel
// consumer in the SAME workgroup - so that the activation between them never leaves the compute unit:
//
//   QKV   WG t owns the 16-column tile t of wqkv: sweep x (all-gather), RMSNorm, MFMA, RoPE, publish q | k | v, append k / v to the cache
//                                                                                                              (gpt.py:215-227,182-183)
//   ATT   WG (row group, head h, share s): split-KV online-softmax attention of its rows over its KV range; the `gs` shares of a (row
//         group, head) exchange their (m, l, acc) partials and EACH merges all of them in share order (identical bits everywhere); the
//         merged attention rows (rounded to T as the chain rounds them) stay in LDS and are multiplied with the head's K-slice of wo
//         right there - share s computes the output columns [s D / gs, (s + 1) D / gs) - publishing fp32 partial rows [m][h][D]
//                                                                                                              (gpt.py:230-238)
//   HRED  unit (row m, column chunk): h = rt(x + rt(sum over heads, in head order)), publish h (all-gather source); the residual stream
//         never leaves the unit's LDS between layers                                                          (gpt.py:257)
//   MLP   WG j owns the 16-wide F slice j: sweep h (all-gather), RMSNorm, its w1 / w3 tiles, silu(a) * b -> 16 columns of g in LDS,
//         times its K-slice of w2 (all D output columns) -> fp32 partial rows [j][m][D]                        (gpt.py:166-167)
//   XRED  unit (m, chunk): x' = rt(h + rt(sum over the F / 16 slices, in a fixed order)), publish x'            (gpt.py:258)
//
// Hand-offs per layer: x all-gather, q | k | v, share partials, wo partials, h all-gather, w2 partials.  The attention output and the
// SwiGLU output - two all-gathers of pdecode.hip - are gone, and so are the K-slice merges of its wo / w2 tiles (the reducers that
// replace them ARE the producers of the next all-gather).  The price: fp32 partial rows instead of T-typed rows on two edges (M D (H + F / 16)
// granules per layer: 9.9 MB at 4 rows of GPT-XL, which is why this form serves small row counts and pdecode.hip the rest), and each wo
// K-slice is read by the M / rg * gs ... workgroups of its head (from L2 after the first).
//
// Everything else is pdecode.hip's: flag-in-data granules (4 payload bytes + tag per 8-byte write-through store, 16-byte sc1 polls),
// regions double-buffered by layer parity, tags = step * 8 L + 8 layer + edge + 1, every wait bounded (handle fault, VLG_ERR_STATE),
// fragment-major weights straight into the MFMA B registers one phase ahead, the chain's rounding points (results agree with the launch
// chain up to fp32 summation order: per-head / per-slice partial sums are added in a fixed order, never atomically).
#include <algorithm>
#include <type_traits>

#include "gpt_kernels.h"
#include "pd_common.h"

namespace vlg {

// ---- persistent decode step, second form (pdecode2.hip): six hand-offs per layer, small row counts ------------------------------------
// work split of a launch, shared by host and device
struct Pd2Geom {
  int rg = 0, gs = 0, nitems = 0;   // attention items (row group of rg rows, head, share s of gs): n_rg * H * gs <= workgroups
  int cw = 0, upr = 0, nunits = 0;  // reducer units (row, chunk of cw columns): M * upr <= workgroups
  bool ok = false;
  __host__ __device__ Pd2Geom(int M, int D, int H, int hd, int esz, int G) {
    const int ksh = hd * esz / 64, ntd = D / 16;
    if (ksh < 1 || G < 1) return;
    const int tpw_max = 10 / ksh;                       // output tiles of the wo slice one wave can hold (PD2_NF fragments)
    // row-group size: the least K / V work per item (rg rows x 1 / gs of the range); ties -> fewer rows per item
    for (int c = 1; c <= 16; c *= 2) {
      const int n = (M + c - 1) / c;
      if (n * H > G) continue;
      int g = G / (n * H);
      if (g > 8) g = 8;
      if (((ntd + g - 1) / g + 7) / 8 > tpw_max) continue;
      if (rg == 0 || c * gs < rg * g) {
        rg = c;
        gs = g;
      }
    }
    if (rg == 0) return;
    nitems = ((M + rg - 1) / rg) * H * gs;
    cw = ((M * D + G - 1) / G + 3) / 4 * 4;
    if (cw < 4) cw = 4;
    while (M * ((D + cw - 1) / cw) > G) cw += 4;
    upr = (D + cw - 1) / cw;
    nunits = M * upr;
    ok = cw <= 512;
  }
};
// granule offsets (8-byte units) of the exchange regions, both parities
struct Pd2Xbuf {
  unsigned nx, nq, nap, nwp, nw2p, per_parity;
  __host__ __device__ Pd2Xbuf(int M, int D, int F, int H, int hd, int esz, const Pd2Geom& g) {
    nx = (unsigned)M * D * esz / 4;
    nq = 3 * nx;
    nap = (unsigned)g.nitems * g.rg * (hd + 2);
    nwp = (unsigned)M * H * D;
    nw2p = (unsigned)(F / 16) * M * D;
    per_parity = 2 * nx + nq + nap + nwp + nw2p;
  }
  __host__ __device__ unsigned X(int par) const { return par * per_parity; }            // layer input rows (T)
  __host__ __device__ unsigned Q(int par) const { return X(par) + nx; }                 // q | k | v rows of the current position, RoPE applied (T)
  __host__ __device__ unsigned HH(int par) const { return Q(par) + nq; }                // residual stream after attention (T)
  __host__ __device__ unsigned AP(int par) const { return HH(par) + nx; }               // attention partials [item][row][hd + 2] (fp32)
  __host__ __device__ unsigned WP(int par) const { return AP(par) + nap; }              // wo partial rows [m][head][D] (fp32)
  __host__ __device__ unsigned W2P(int par) const { return WP(par) + nwp; }             // w2 partial rows [F slice][m][D] (fp32)
  __host__ __device__ size_t bytes() const { return (size_t)2 * per_parity * 8; }
};
template <typename T>
bool pd2_ok(int M, int D, int H, int hd, int F, int cus);
size_t pd2_xbuf_bytes(int M, int D, int H, int hd, int F, int esz, int cus);
template <typename T>
int pd2_layers(PdArgs a, hipStream_t st);   // needs a.fm (fragment-major weight copies)


namespace {

// LDS carve-up, shared by kernel and launcher
struct Pd2Lds {
  unsigned a_stride, as, red, gt, resid, part, qkv, sm, pall, flags, total;
  __host__ __device__ Pd2Lds(int D, int esz, int hd, int rg, int gs, int cw) {
    auto al = [](unsigned v) { return (v + 15u) & ~15u; };
    a_stride = (unsigned)D * esz + 32;                       // + 32 bytes: the 16 rows of a ds_read_b128 A fragment fall on 16 different slots
    unsigned o = 0;
    as = o; o += 16u * a_stride;                             // activation rows (x / h; then the merged attention rows of the item's head)
    red = o; o += 8u * 2 * 256 * 4;                          // per-wave partial accumulators [wave][tile][256]
    gt = o; o += 16u * 80;                                   // the workgroup's 16 columns of g as ONE K step of A: [16 rows][64 + 16 bytes]
    resid = o; o = al(o + (unsigned)cw * 4);                 // this unit's chunk of the residual stream (lives across layers)
    part = o; o += 1024u * 4;                                // reducer partial sums [group][chunk columns]: floor(512 / (cw / 2)) * cw <= 1024
    qkv = o; o = al(o + (unsigned)rg * 3 * hd * esz);        // q | k | v of the item's rows
    sm = o; o = al(o + 8u * (hd + 2) * 4);                   // per-wave (m, l, acc)
    pall = o; o = al(o + (unsigned)gs * rg * (hd + 2) * 4);  // every share's partial of every row [share][row][hd + 2]
    flags = o;
    total = o + 64;
  }
};

#ifdef VLG_PD_PROF   // in-kernel time stamps of layer 1, kept in LDS until the end (tools/microbench/pd_lab.hip)
#define PD2_STAMP(id)                                       \
  do {                                                      \
    if (tid == 0 && l == 1) prof_s[id] = wall_clock64();    \
  } while (0)
#else
#define PD2_STAMP(id) \
  do {                \
  } while (0)
#endif

constexpr int PD2_NW = 8;
constexpr int PD2_NTHR = PD2_NW * 64;
constexpr int PD2_NF = 10;          // weight fragments per register set
constexpr int PD2_UB = 8;           // 16-byte polls in flight per thread

template <typename T, int HD, int VEC, int LPR>
__global__ __launch_bounds__(PD2_NTHR) void pd2_layers_kernel(PdArgs a) {
  constexpr int ESZ = (int)sizeof(T);
  constexpr int EPV = 16 / ESZ;        // elements per 16-byte chunk
  constexpr int KS = 64 / ESZ;         // K elements per MFMA step of one fragment (32 bf16 / 16 fp32)
  constexpr int EPG = 4 / ESZ;         // elements per granule
  constexpr int KSH = HD * ESZ / 64;   // K steps of one head
  static_assert(HD * ESZ % 64 == 0, "a head is a whole number of K steps");
  extern __shared__ __attribute__((aligned(16))) char pd_smem[];
  const int M = a.M, D = a.D, F = a.F, H = a.H, L = a.L;
  const Pd2Geom geo(M, D, H, HD, ESZ, (int)gridDim.x);
  const int RG = geo.rg, GS = geo.gs, CW = geo.cw;
  const Pd2Lds lds(D, ESZ, HD, RG, GS, CW);
  char* As = pd_smem + lds.as;
  float* red = reinterpret_cast<float*>(pd_smem + lds.red);
  char* gt = pd_smem + lds.gt;
  float* resid = reinterpret_cast<float*>(pd_smem + lds.resid);
  float* part = reinterpret_cast<float*>(pd_smem + lds.part);
  char* qkv_s = pd_smem + lds.qkv;                                                  // [rg][3][HD] T
  float* sm = reinterpret_cast<float*>(pd_smem + lds.sm);                           // [NW][HD + 2]
  float* pall = reinterpret_cast<float*>(pd_smem + lds.pall);                       // [gs][rg][HD + 2]
  int* flags = reinterpret_cast<int*>(pd_smem + lds.flags);                         // [0] alive
  const unsigned a_stride = lds.a_stride;
#ifdef VLG_PD_PROF
  __shared__ unsigned long long prof_s[32];
  if (threadIdx.x < 32) prof_s[threadIdx.x] = 0;
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wg = blockIdx.x;
  const int pos = a.state->pos;
  const unsigned eb = (unsigned)a.state->step * (unsigned)(8 * L);
  const Pd2Xbuf xb(M, D, F, H, HD, ESZ, geo);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, (int)a.xbuf_bytes, 0x00020000);
  const T* xg = reinterpret_cast<const T*>(a.x);
  typedef const PdLayer __attribute__((address_space(4))) * layer_cptr;
  const layer_cptr layers_c = (layer_cptr)(uintptr_t)a.layers;

  // a fault word that is already set: leave at once (the call is lost either way; pdecode.hip)
  if (tid == 0) flags[0] = (a.fault == nullptr || __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) ? 1 : 0;
  for (unsigned i = tid; i < 16u * a_stride / 16; i += PD2_NTHR) reinterpret_cast<pd_u32x4_t*>(As)[i] = pd_u32x4_t{0u, 0u, 0u, 0u};
  for (unsigned i = tid; i < 16u * 80 / 16; i += PD2_NTHR) reinterpret_cast<pd_u32x4_t*>(gt)[i] = pd_u32x4_t{0u, 0u, 0u, 0u};
  pd_barrier();
  if (flags[0] == 0) return;

  // ---- roles of this workgroup (one per phase) ---------------------------------------------------------------------------------------
  const int NTD = D / 16;                                       // output tiles of a D-wide GEMM
  const bool has_qkv = wg < 3 * NTD, has_f = wg < F / 16;
  const bool has_it = wg < geo.nitems, has_u = wg < geo.nunits;
  const int it_s = wg % GS, it_h = (wg / GS) % H, it_g = wg / (GS * H);          // share, head, row group of the attention item
  const int it_m0 = it_g * RG, it_rows = has_it ? min(RG, M - it_m0) : 0;
  const int ct_lo = NTD * it_s / GS, ct_hi = NTD * (it_s + 1) / GS;               // wo output tiles of this share
  const int u_m = wg / geo.upr, u_c0 = (wg - u_m * geo.upr) * CW;                // reducer unit: row, first column
  const int u_cw = has_u ? min(CW, D - u_c0) : 0;
  const int nks_d = D / KS;
  const int ksw_d = (nks_d - wave + PD2_NW - 1) / PD2_NW;        // K steps of this wave in a full-depth GEMM over D

  // ---- hand-off primitives (pdecode.hip) ------------------------------------------------------------------------------------------------
  auto sweep = [&](char* dst, unsigned dstride, unsigned gbase, int N, int c0, int c1, unsigned tag) __attribute__((always_inline)) {
    const int ppr = (c1 - c0) * ESZ / 8;
    const int npair = M * ppr;
    for (int i0 = tid; i0 < npair; i0 += PD2_NTHR * PD2_UB) {
      pd_u32x4_t v[PD2_UB];
      int goff[PD2_UB];
      bool need[PD2_UB];
#pragma unroll
      for (int u = 0; u < PD2_UB; ++u) {
        const int i = i0 + u * PD2_NTHR;
        const int ii = i < npair ? i : npair - 1;
        const int row = ii / ppr, wi = ii - row * ppr;
        goff[u] = (int)((gbase + (unsigned)((row * N + c0) * ESZ / 4)) * 8u) + wi * 16;
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[u], 0, 16);
      }
      for (int spin = 0;; ++spin) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < PD2_UB; ++u) {
          need[u] = (i0 + u * PD2_NTHR < npair) && (v[u][1] != tag || v[u][3] != tag);
          any = any || need[u];
        }
        if (!any) break;
        if (spin >= a.spin_max) {
          flags[0] = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int u = 0; u < PD2_UB; ++u)
          if (need[u]) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[u], 0, 16);
      }
#pragma unroll
      for (int u = 0; u < PD2_UB; ++u) {
        const int i = i0 + u * PD2_NTHR;
        if (i < npair) {
          const int row = i / ppr, wi = i - row * ppr;
          *reinterpret_cast<pd_u32x2_t*>(dst + (size_t)row * dstride + wi * 8) = pd_u32x2_t{v[u][0], v[u][2]};
        }
      }
    }
  };
  // one 16-byte poll: two granules at byte offset goff until both carry `tag` (bounded)
  auto poll2 = [&](int goff, unsigned tag) __attribute__((always_inline)) -> pd_u32x4_t {
    pd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 16);
    for (int spin = 0; v[1] != tag || v[3] != tag; ++spin) {
      if (spin >= a.spin_max) {
        flags[0] = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 16);
    }
    return v;
  };
  // one value per thread -> granule of element (row, col) of a tagged [.][N] T matrix; bf16: the even-column lane stores the pair
  auto publish = [&](unsigned gbase, int N, int row, int col, float v, bool valid, unsigned tag) __attribute__((always_inline)) {
    if constexpr (ESZ == 2) {
      const float vn = __shfl_down(v, 1);
      if (valid && !(col & 1))
        __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{pd_pack2(v, vn), tag}, rs, (int)((gbase + (unsigned)(row * N + col) / 2) * 8u), 0, 16);
    } else {
      if (valid) __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{__float_as_uint(v), tag}, rs, (int)((gbase + (unsigned)(row * N + col)) * 8u), 0, 16);
    }
  };
  auto put_f32 = [&](unsigned gidx, float v, unsigned tag) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{__float_as_uint(v), tag}, rs, (int)(gidx * 8u), 0, 16);
  };
  // Reducer: sum over `nsl` fp32 partial rows of this unit's columns, slice sl at granule gbase + sl * sl_stride (+ column).  Thread (g, cp):
  // column pair cp of slices g, g + NG, ... added in that order; the NG group sums are then added in group order (fixed order, no atomics).
  // Returns the sum of column `tid` to threads tid < u_cw (0 elsewhere); ends with the workgroup in sync.
  auto reduce_unit = [&](unsigned gbase, unsigned sl_stride, int nsl, unsigned tag) __attribute__((always_inline)) -> float {
    const int CP = u_cw / 2, NG = PD2_NTHR / CP;
    const int g = tid / CP, cp = tid - g * CP;
    float a0 = 0.f, a1 = 0.f;
    if (g < NG) {
      for (int sl0 = g; sl0 < nsl; sl0 += NG * PD2_UB) {
        pd_u32x4_t v[PD2_UB];
        int goff[PD2_UB];
        bool need[PD2_UB];
#pragma unroll
        for (int u = 0; u < PD2_UB; ++u) {
          const int sl = sl0 + u * NG;
          goff[u] = (int)((gbase + (unsigned)(sl < nsl ? sl : sl0) * sl_stride + (unsigned)(2 * cp)) * 8u);
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[u], 0, 16);
        }
        for (int spin = 0;; ++spin) {
          bool any = false;
#pragma unroll
          for (int u = 0; u < PD2_UB; ++u) {
            need[u] = (sl0 + u * NG < nsl) && (v[u][1] != tag || v[u][3] != tag);
            any = any || need[u];
          }
          if (!any) break;
          if (spin >= a.spin_max) {
            flags[0] = 0;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int u = 0; u < PD2_UB; ++u)
            if (need[u]) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[u], 0, 16);
        }
#pragma unroll
        for (int u = 0; u < PD2_UB; ++u)
          if (sl0 + u * NG < nsl) {
            a0 += __uint_as_float(v[u][0]);
            a1 += __uint_as_float(v[u][2]);
          }
      }
      part[g * u_cw + 2 * cp] = a0;
      part[g * u_cw + 2 * cp + 1] = a1;
    }
    pd_barrier();
    float s = 0.f;
    if (tid < u_cw) {
      const int ng = min(NG, nsl);
      for (int g2 = 0; g2 < ng; ++g2) s += part[g2 * u_cw + tid];
    }
    return s;
  };

  // ---- weights (fragment-major copies, gpt_kernels.h relayout_fragment_major): block (tile, K step) = 1 KB, this lane's 16 bytes at lane * 16
  pd_u32x4_t R[PD2_NF];
  const unsigned lane16 = (unsigned)lane * 16u;
  // full-depth GEMM over D: fragment j * NTW + t = K step wave + 8 j of weight tile (row0 + t * row_step) / 16
  auto load_set = [&](const void* wv, int row0, int row_step, auto ntw_c) __attribute__((always_inline)) {
    constexpr int NTW = decltype(ntw_c)::value;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const char* tb = reinterpret_cast<const char*>(wv) + ((size_t)((row0 + t * row_step) >> 4) * (size_t)nks_d + (size_t)wave) * 1024;   // wave-uniform
#pragma unroll
      for (int j = 0; j < PD2_NF / NTW; ++j)
        if (j < ksw_d) R[j * NTW + t] = __builtin_nontemporal_load((pd_gptr16)(uintptr_t)(tb + lane16 + j * (PD2_NW * 1024)));
    }
  };
  auto gemm_d = [&](pd_f32x4_t (&acc)[2], auto ntw_c) __attribute__((always_inline)) {
    constexpr int NTW = decltype(ntw_c)::value;
    unsigned abase = (unsigned)r * a_stride + (unsigned)q * 16u + (unsigned)wave * (unsigned)(KS * ESZ);
    asm volatile("" : "+v"(abase));   // opaque per call: otherwise hipcc hoists one address per (phase, fragment) out of the layer loop and spills
#pragma unroll
    for (int j = 0; j < PD2_NF / NTW; ++j)
      if (j < ksw_d) {
        const pd_u32x4_t af = *reinterpret_cast<const pd_u32x4_t*>(As + abase + (unsigned)(j * PD2_NW * KS * ESZ));
#pragma unroll
        for (int t = 0; t < NTW; ++t) pd_mfma<T>(af, R[j * NTW + t], acc[t]);
      }
  };
  auto kill_set = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PD2_NF; ++i) R[i] = pd_u32x4_t{0u, 0u, 0u, 0u};
  };
  auto store_red = [&](const pd_f32x4_t (&acc)[2], int ntw) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < ntw) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave * 2 + t) * 256 + e * 64 + lane] = acc[t][e];
      }
  };
  auto reduced = [&](int t, int idx) __attribute__((always_inline)) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < PD2_NW; ++wv) s += red[(wv * 2 + t) * 256 + idx];
    return s;
  };
  auto load_norm_w = [&](pd_u32x4_t (&gv)[4], const void* gw) __attribute__((always_inline)) {
    const int nch = D / EPV;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = lane + 64 * it;
      gv[it] = ((pd_gptr16)(uintptr_t)gw)[c < nch ? c : 0];
    }
  };
  // in-place RMSNorm of the M rows of As (K = D): rt(rt(x * rsqrt(mean(x^2) + eps)) * g)      (gpt.py:143-148)
  auto rmsnorm = [&](const pd_u32x4_t (&gv)[4]) __attribute__((always_inline)) {
    const int nch = D / EPV;
    for (int row0 = wave; row0 < M; row0 += 2 * PD2_NW) {
      pd_u32x4_t xv[2][4];
      float ss[2] = {0.f, 0.f};
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = row0 + rr * PD2_NW < M ? row0 + rr * PD2_NW : row0;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c = lane + 64 * it;
          if (c < nch) xv[rr][it] = *reinterpret_cast<const pd_u32x4_t*>(As + (size_t)row * a_stride + (size_t)c * 16);
        }
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int it = 0; it < 4; ++it)
          if (lane + 64 * it < nch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if constexpr (ESZ == 2) {
                const float lo = pd_lo(xv[rr][it][j]), hi = pd_hi(xv[rr][it][j]);
                ss[rr] += lo * lo + hi * hi;
              } else {
                const float f = __uint_as_float(xv[rr][it][j]);
                ss[rr] += f * f;
              }
            }
          }
      ss[0] = pd_wave_sum(ss[0]);
      ss[1] = pd_wave_sum(ss[1]);
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = row0 + rr * PD2_NW;
        const float rsq = 1.0f / sqrtf(ss[rr] / (float)D + a.eps);
        if (row < M) {
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int c = lane + 64 * it;
            if (c < nch) {
              pd_u32x4_t o;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if constexpr (ESZ == 2) {
                  const unsigned n = pd_pack2(pd_lo(xv[rr][it][j]) * rsq, pd_hi(xv[rr][it][j]) * rsq);
                  o[j] = pd_pack2(pd_lo(n) * pd_lo(gv[it][j]), pd_hi(n) * pd_hi(gv[it][j]));
                } else {
                  o[j] = __float_as_uint(__uint_as_float(xv[rr][it][j]) * rsq * __uint_as_float(gv[it][j]));
                }
              }
              *reinterpret_cast<pd_u32x4_t*>(As + (size_t)row * a_stride + (size_t)c * 16) = o;
            }
          }
        }
      }
    }
  };

  bool alive = true;
#define PD2_SYNC_ALIVE()       \
  do {                         \
    pd_barrier();              \
    alive = flags[0] != 0;     \
  } while (0)

  // the residual stream of this unit: layer 0's input
  if (tid < u_cw) resid[tid] = DT<T>::ld(xg + (size_t)u_m * D + u_c0 + tid);
  // first weights: qkv of layer 0
  if (has_qkv) load_set(layers_c[0].wqkv, wg * 16, 0, std::integral_constant<int, 1>{});

  const float att_scale = 1.0f / sqrtf((float)HD);
  const int tid_k = tid;
  for (int l = 0; l < L && alive; ++l) {
    // Thread indices re-derived from an opaque copy once per layer: everything computed from them below (tile numbers, publish offsets,
    // LDS addresses, predicates) is then loop-VARIANT to hipcc.  Otherwise it hoists dozens of such values out of the layer loop, runs out
    // of registers and spills them - and each scratch reload is a vector-memory operation whose s_waitcnt vmcnt(0) also waits for every
    // hand-off store still in flight (measured: the ten publish steps of the w2 slice serialised on it, 7 us instead of 1).
    int tid_l = tid_k;
    asm volatile("" : "+v"(tid_l));
    const int tid = tid_l, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const unsigned lane16 = (unsigned)lane * 16u;
    struct {
      const void *wo, *w13, *w2, *norm1, *norm2;
    } ly = {layers_c[l].wo, layers_c[l].w13, layers_c[l].w2, layers_c[l].norm1, layers_c[l].norm2};
    const int par = l & 1;
    const unsigned tag0 = eb + (unsigned)(8 * l) + 1u;   // + edge: 0 x, 1 qkv, 2 share partials, 3 wo partials, 4 h, 5 w2 partials
    pd_f32x4_t acc[2];

    // =========================== QKV ===========================
    PD2_STAMP(0);
    if (has_qkv) {
      const int ecol = wg * 16 + (tid & 15);
      const int esec = ecol / D, ewithin = ecol - esec * D;
      const int ehh = ewithin / HD, ed = ewithin - ehh * HD;
      const float* cp = a.freqs + ((size_t)pos * (HD / 2) + ed / 2) * 2;
      const float cx = cp[0], cy = cp[1];
      pd_u32x4_t gv[4];
      load_norm_w(gv, ly.norm1);
      if (l == 0) {
        const int nch = D / EPV;
        for (int i = tid; i < M * nch; i += PD2_NTHR) {
          const int row = i / nch, c = i - row * nch;
          *reinterpret_cast<pd_u32x4_t*>(As + (size_t)row * a_stride + (size_t)c * 16) = reinterpret_cast<const pd_u32x4_t*>(xg + (size_t)row * D)[c];
        }
      } else {
        sweep(As, a_stride, xb.X(par), D, 0, D, tag0 + 0u);
      }
      PD2_SYNC_ALIVE();
      if (!alive) break;
      PD2_STAMP(1);
      rmsnorm(gv);
      pd_barrier();
      acc[0] = pd_f32x4_t{0.f, 0.f, 0.f, 0.f};
      gemm_d(acc, std::integral_constant<int, 1>{});
      kill_set();
      store_red(acc, 1);
      PD2_STAMP(2);
      pd_barrier();
      // epilogue: RoPE on adjacent pairs, publish q | k | v, append k / v to the cache (gemm_fused.hip EPI_QKV)
      if (tid < 256) {
        const int e = tid >> 6, l2 = tid & 63;
        const int row = (l2 >> 4) * 4 + e;
        const float xs = DT<T>::rt(reduced(0, tid)), xp = DT<T>::rt(reduced(0, tid ^ 1));
        float o = xs;
        if (esec < 2) o = (ed & 1) ? __fadd_rn(__fmul_rn(xs, cx), __fmul_rn(xp, cy)) : __fsub_rn(__fmul_rn(xs, cx), __fmul_rn(xp, cy));
        o = DT<T>::rt(o);
        const bool valid = row < M;
        publish(xb.Q(par), 3 * D, row, ecol, o, valid, tag0 + 1u);
        if (valid && esec >= 1) {
          T* cache = reinterpret_cast<T*>(esec == 1 ? a.kc : a.vc) + (size_t)l * a.kv_lstride;
          DT<T>::st(cache + (((size_t)row * H + ehh) * a.S + pos) * HD + ed, o);
        }
      }
    }
    PD2_STAMP(3);

    // =========================== ATT + wo slice ===========================
    if (has_it) {
      const int nkeys = pos + 1;
      constexpr int RPI = 64 / LPR, U = 8, TILE = RPI * U;
      const int g = lane / LPR, c = lane % LPR;
      const bool active = c * VEC < HD;
      const int coff = active ? c * VEC : 0;
      const int chunk = (nkeys + GS - 1) / GS;
      const int r0 = it_s * chunk, r1 = min(r0 + chunk, nkeys);
      const int r1c = min(r1, pos);     // rows of the cache (all < pos: row `pos` is appended by this launch for later steps and never read here)
      // q | k | v of the item's rows from the QKV hand-off -> LDS
      {
        constexpr int PPR = HD * ESZ / 8;
        for (int i = tid; i < it_rows * 3 * PPR; i += PD2_NTHR) {
          const int rr = i / (3 * PPR), rem = i - rr * 3 * PPR;
          const int sec = rem / PPR, wi = rem - sec * PPR;
          const int goff = (int)((xb.Q(par) + (unsigned)(((it_m0 + rr) * 3 * D + sec * D + it_h * HD) * ESZ / 4)) * 8u) + wi * 16;
          const pd_u32x4_t v = poll2(goff, tag0 + 1u);
          *reinterpret_cast<pd_u32x2_t*>(qkv_s + ((size_t)(rr * 3 + sec) * HD) * ESZ + wi * 8) = pd_u32x2_t{v[0], v[2]};
        }
      }
      PD2_SYNC_ALIVE();
      if (!alive) break;
      PD2_STAMP(4);
      for (int rr = 0; rr < it_rows; ++rr) {
        const int m = it_m0 + rr;
        const char* kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.kc) + (size_t)l * a.kv_lstride + ((size_t)m * H + it_h) * (size_t)a.S * HD);
        const char* vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.vc) + (size_t)l * a.kv_lstride + ((size_t)m * H + it_h) * (size_t)a.S * HD);
        float qf[VEC];
        PdPack<T, VEC> knew, vnew;
        {
          const T* qs = reinterpret_cast<const T*>(qkv_s) + (size_t)(rr * 3 + 0) * HD;
          const T* ks = reinterpret_cast<const T*>(qkv_s) + (size_t)(rr * 3 + 1) * HD;
          const T* vs = reinterpret_cast<const T*>(qkv_s) + (size_t)(rr * 3 + 2) * HD;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            qf[j] = active ? DT<T>::ld(qs + coff + j) : 0.f;
            knew.v[j] = ks[coff + j];
            vnew.v[j] = vs[coff + j];
          }
        }
        const float* mrow = (a.mask != nullptr) ? a.mask + (size_t)(m % a.Bmask) * a.Tc : nullptr;
        float mx = -INFINITY, lsum = 0.f, av[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) av[j] = 0.f;
        for (int tile = r0 + wave * TILE; tile < r1c; tile += PD2_NW * TILE) {
          PdPack<T, VEC> kk[U], vv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = tile + u * RPI + g;
            const int rw = row < r1c ? row : r1c - 1;
            const unsigned lo = (unsigned)(rw * HD + coff) * ESZ;
            kk[u] = pd_load_stream<T, VEC>(kbase + lo);
            vv[u] = pd_load_stream<T, VEC>(vbase + lo);
          }
          float s[U];
          float tmax = -INFINITY;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = tile + u * RPI + g;
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) d = fmaf(qf[j], DT<T>::ld(&kk[u].v[j]), d);
            d = pd_group_sum<LPR>(d) * att_scale;
            bool ok = row < r1c;
            if (mrow != nullptr && row < a.Tc) ok = ok && (mrow[row < a.Tc ? row : 0] != 0.f);
            s[u] = ok ? d : -INFINITY;
            tmax = fmaxf(tmax, s[u]);
          }
          const float mnew = fmaxf(mx, tmax);
          const float mref = (mnew == -INFINITY) ? 0.f : mnew;
          const float alpha = __expf(mx - mref);
          lsum *= alpha;
#pragma unroll
          for (int j = 0; j < VEC; ++j) av[j] *= alpha;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float pu = __expf(s[u] - mref);
            lsum += pu;
#pragma unroll
            for (int j = 0; j < VEC; ++j) av[j] = fmaf(pu, DT<T>::ld(&vv[u].v[j]), av[j]);
          }
          mx = mnew;
        }
        // the key of the current position (never masked: generate.py:156-165 leaves the diagonal on), by lane group 0 of wave 0 of the share
        // whose range holds it
        {
          float d = 0.f;
#pragma unroll
          for (int j = 0; j < VEC; ++j) d = fmaf(qf[j], DT<T>::ld(&knew.v[j]), d);
          d = pd_group_sum<LPR>(d) * att_scale;
          const bool mine = (r0 <= pos && pos < r1) && wave == 0 && g == 0;
          const float sn = mine ? d : -INFINITY;
          const float mnew = fmaxf(mx, sn);
          const float mref = (mnew == -INFINITY) ? 0.f : mnew;
          const float alpha = __expf(mx - mref), pu = __expf(sn - mref);
          lsum = lsum * alpha + pu;
#pragma unroll
          for (int j = 0; j < VEC; ++j) av[j] = fmaf(pu, DT<T>::ld(&vnew.v[j]), av[j] * alpha);
          mx = mnew;
        }
        PD2_STAMP(13);
        // merge the lane groups of the wave, then the waves (same arithmetic as attn_partial_kernel)
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
          const float mo = __shfl_xor(mx, off), lo = __shfl_xor(lsum, off);
          const float mn = fmaxf(mx, mo);
          const float mref = (mn == -INFINITY) ? 0.f : mn;
          const float ea = __expf(mx - mref), ebb = __expf(mo - mref);
          lsum = lsum * ea + lo * ebb;
#pragma unroll
          for (int j = 0; j < VEC; ++j) av[j] = av[j] * ea + __shfl_xor(av[j], off) * ebb;
          mx = mn;
        }
        if (g == 0 && active) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) sm[wave * (HD + 2) + 2 + coff + j] = av[j];
          if (c == 0) {
            sm[wave * (HD + 2) + 0] = mx;
            sm[wave * (HD + 2) + 1] = lsum;
          }
        }
        pd_barrier();
        PD2_STAMP(14);
        // this share's (m, l, acc[HD]) of the row: kept, and handed to the other shares
        if (tid < HD + 2) {
          float M8 = -INFINITY;
#pragma unroll
          for (int wv = 0; wv < PD2_NW; ++wv) M8 = fmaxf(M8, sm[wv * (HD + 2)]);
          const float mref = (M8 == -INFINITY) ? 0.f : M8;
          float val;
          if (tid == 0) {
            val = M8;
          } else {
            val = 0.f;
#pragma unroll
            for (int wv = 0; wv < PD2_NW; ++wv) val += sm[wv * (HD + 2) + tid] * __expf(sm[wv * (HD + 2)] - mref);
          }
          pall[(it_s * RG + rr) * (HD + 2) + tid] = val;
          if (GS > 1) put_f32(xb.AP(par) + (unsigned)((wg * RG + rr) * (HD + 2) + tid), val, tag0 + 2u);
        }
        pd_barrier();   // sm is reused by the next row
      }
      PD2_STAMP(5);
      // The head's K-slice of wo for this share's output columns: requested here, behind the K / V stream (its registers are free now), and
      // landing under the exchange below.  Fragment j * KSH + k = K step it_h * KSH + k of output tile ct_lo + wave + 8 j.
      {
        // one running pointer, opaque to the optimiser: a run-time tile stride otherwise makes hipcc keep one address per fragment across the
        // layer loop, spill them, and put a scratch reload (which queues behind the weight stream) in front of every weight request
        const char* wp = reinterpret_cast<const char*>(ly.wo) + ((size_t)(ct_lo + wave) * (size_t)nks_d + (size_t)it_h * KSH) * 1024 + lane16;
        size_t tstride = (size_t)PD2_NW * (size_t)nks_d * 1024;
        asm volatile("" : "+s"(tstride));
#pragma unroll
        for (int j = 0; j < PD2_NF / KSH; ++j) {
          if (ct_lo + wave + PD2_NW * j < ct_hi) {
#pragma unroll
            for (int k = 0; k < KSH; ++k) R[j * KSH + k] = __builtin_nontemporal_load((pd_gptr16)(uintptr_t)(wp + k * 1024));
          }
          wp += tstride;
        }
      }
      // the other shares' partials
      if (GS > 1) {
        const int per = it_rows * (HD + 2);                      // granules per share (HD + 2 is even)
        for (int i = tid; i < (GS - 1) * per / 2; i += PD2_NTHR) {
          const int o = i / (per / 2), wi = i - o * (per / 2);
          const int sp = o + (o >= it_s ? 1 : 0);
          const int goff = (int)((xb.AP(par) + (unsigned)(((wg - it_s + sp) * RG) * (HD + 2))) * 8u) + wi * 16;
          const pd_u32x4_t v = poll2(goff, tag0 + 2u);
          pall[(sp * RG) * (HD + 2) + 2 * wi] = __uint_as_float(v[0]);
          pall[(sp * RG) * (HD + 2) + 2 * wi + 1] = __uint_as_float(v[2]);
        }
      }
      PD2_SYNC_ALIVE();
      if (!alive) break;
      PD2_STAMP(6);
      // merge in share order (attn_combine_kernel), round to T, -> rows 0 .. it_rows - 1 of As, K columns 0 .. HD - 1
      for (int i = tid; i < it_rows * (HD / EPG); i += PD2_NTHR) {
        const int rr = i / (HD / EPG), t = i - rr * (HD / EPG);
        float Mx = -INFINITY;
        for (int sp = 0; sp < GS; ++sp) Mx = fmaxf(Mx, pall[(sp * RG + rr) * (HD + 2)]);
        const float mref = (Mx == -INFINITY) ? 0.f : Mx;
        float Ls = 0.f, A0 = 0.f, A1 = 0.f;
        for (int sp = 0; sp < GS; ++sp) {
          const float* pp = pall + (sp * RG + rr) * (HD + 2);
          const float ee = __expf(pp[0] - mref);
          Ls += pp[1] * ee;
          A0 += pp[2 + t * EPG] * ee;
          if constexpr (EPG == 2) A1 += pp[2 + t * EPG + 1] * ee;
        }
        if constexpr (EPG == 2)
          *reinterpret_cast<unsigned*>(As + (size_t)rr * a_stride + (size_t)t * 4) = pd_pack2(A0 / Ls, A1 / Ls);
        else
          *reinterpret_cast<float*>(As + (size_t)rr * a_stride + (size_t)t * 4) = A0 / Ls;
      }
      pd_barrier();
      // partial[m][h][col] = attention row (K = this head) . wo[col][head's K range]: every wave finishes its own tiles (no cross-wave sum)
      {
        pd_u32x4_t af[KSH];
#pragma unroll
        for (int k = 0; k < KSH; ++k) af[k] = *reinterpret_cast<const pd_u32x4_t*>(As + (unsigned)r * a_stride + (unsigned)q * 16u + (unsigned)(k * KS * ESZ));
#pragma unroll
        for (int j = 0; j < PD2_NF / KSH; ++j) {
          const int ct = ct_lo + wave + PD2_NW * j;
          if (ct < ct_hi) {
            pd_f32x4_t o = pd_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KSH; ++k) pd_mfma<T>(af[k], R[j * KSH + k], o);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int rr = q * 4 + e;
              if (rr < it_rows) put_f32(xb.WP(par) + (unsigned)(((it_m0 + rr) * H + it_h) * D + ct * 16 + r), o[e], tag0 + 3u);
            }
          }
        }
        kill_set();
      }
    }
    PD2_STAMP(7);
    // the set is free: w13 of this layer
    if (has_f) load_set(ly.w13, wg * 16, F, std::integral_constant<int, 2>{});

    // =========================== HRED ===========================
    if (has_u) {
      const float s = reduce_unit(xb.WP(par) + (unsigned)(u_m * H * D + u_c0), (unsigned)D, H, tag0 + 3u);
      alive = flags[0] != 0;
      if (!alive) break;
      float v = 0.f;
      if (tid < u_cw) {
        v = DT<T>::rt(resid[tid] + DT<T>::rt(s));       // gemm_fused.hip EPI_RESID
        resid[tid] = v;
      }
      publish(xb.HH(par), D, u_m, u_c0 + tid, v, tid < u_cw, tag0 + 4u);
      pd_barrier();   // part is rewritten by the next reduce
    }
    PD2_STAMP(8);

    // =========================== MLP ===========================
    if (has_f) {
      pd_u32x4_t gv[4];
      load_norm_w(gv, ly.norm2);
      pd_barrier();                                      // every wave has read its attention rows from As
      sweep(As, a_stride, xb.HH(par), D, 0, D, tag0 + 4u);
      PD2_SYNC_ALIVE();
      if (!alive) break;
      PD2_STAMP(9);
      rmsnorm(gv);
      pd_barrier();
      acc[0] = pd_f32x4_t{0.f, 0.f, 0.f, 0.f};
      acc[1] = pd_f32x4_t{0.f, 0.f, 0.f, 0.f};
      gemm_d(acc, std::integral_constant<int, 2>{});
      kill_set();
      store_red(acc, 2);
      // this workgroup's K-slice of w2 (the 16 F columns it is about to produce) for ALL output tiles: tile wave + 8 j.  bf16: the slice is half
      // of a 64-byte K step - lanes of the other half stay zero, and so does their half of the A operand.
      {
        constexpr int SPS = 64 / (16 * ESZ);                                      // F slices per K step (2 bf16, 1 fp32)
        const int kk = wg / SPS, half = wg % SPS;
        const bool mine = SPS == 1 || (q >> 1) == half;
        const char* wp = reinterpret_cast<const char*>(ly.w2) + ((size_t)wave * (size_t)(F / KS) + (size_t)kk) * 1024 + lane16;
        size_t tstride = (size_t)PD2_NW * (size_t)(F / KS) * 1024;
        asm volatile("" : "+s"(tstride));   // running pointer, as for the wo slice
#pragma unroll
        for (int j = 0; j < PD2_NF; ++j) {
          if (wave + PD2_NW * j < NTD && mine) R[j] = __builtin_nontemporal_load((pd_gptr16)(uintptr_t)wp);
          wp += tstride;
        }
      }
      PD2_STAMP(10);
      pd_barrier();
      if (tid < 256) {   // g = rt(rt(silu(rt(a))) * rt(b))      (gemm_fused.hip EPI_SWIGLU) -> the A tile of the w2 slice
        constexpr int SPS = 64 / (16 * ESZ);
        const int half = wg % SPS;
        const int e = tid >> 6, l2 = tid & 63;
        const int row = (l2 >> 4) * 4 + e, col = l2 & 15;
        const float av2 = DT<T>::rt(reduced(0, tid)), bv = DT<T>::rt(reduced(1, tid));
        const float gval = DT<T>::rt(DT<T>::rt(pd_silu(av2)) * bv);
        if (row < M) DT<T>::st(reinterpret_cast<T*>(gt + (size_t)row * 80) + half * 16 + col, gval);
      }
      pd_barrier();
      {
        const pd_u32x4_t af = *reinterpret_cast<const pd_u32x4_t*>(gt + (unsigned)r * 80u + (unsigned)q * 16u);
#pragma unroll
        for (int j = 0; j < PD2_NF; ++j) {
          const int ct = wave + PD2_NW * j;
          if (ct < NTD) {
            pd_f32x4_t o = pd_f32x4_t{0.f, 0.f, 0.f, 0.f};
            pd_mfma<T>(af, R[j], o);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = q * 4 + e;
              if (row < M) put_f32(xb.W2P(par) + (unsigned)((wg * M + row) * D + ct * 16 + r), o[e], tag0 + 5u);
            }
          }
        }
        kill_set();
      }
    }
    PD2_STAMP(11);
    // the set is free: qkv of the next layer
    if (l + 1 < L && has_qkv) load_set(layers_c[l + 1].wqkv, wg * 16, 0, std::integral_constant<int, 1>{});

    // =========================== XRED ===========================
    if (has_u) {
      const float s = reduce_unit(xb.W2P(par) + (unsigned)(u_m * D + u_c0), (unsigned)(M * D), F / 16, tag0 + 5u);
      alive = flags[0] != 0;
      if (!alive) break;
      float v = 0.f;
      if (tid < u_cw) {
        v = DT<T>::rt(resid[tid] + DT<T>::rt(s));
        resid[tid] = v;
      }
      if (l + 1 < L)
        publish(xb.X(1 - par), D, u_m, u_c0 + tid, v, tid < u_cw, tag0 + 8u);   // = edge 0 of layer l + 1
      else if (tid < u_cw)
        DT<T>::st(reinterpret_cast<T*>(a.x) + (size_t)u_m * D + u_c0 + tid, v);
      pd_barrier();   // part is rewritten by the next reduce
    }
    PD2_STAMP(12);
  }
#undef PD2_SYNC_ALIVE
#ifdef VLG_PD_PROF
  pd_barrier();
  if (a.prof && threadIdx.x < 32) a.prof[(size_t)blockIdx.x * 32 + threadIdx.x] = prof_s[threadIdx.x];
#endif
  // a wait ran out: tell the host (vlg_gpt_status); the word lives in pinned host memory
  if (!alive && tid == 0 && a.fault) __hip_atomic_store(a.fault, kFaultDecode | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int pd2_cu_count() {
  static int cache[64] = {};
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev < 64 && cache[dev] > 0) return cache[dev];
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (dev < 64) cache[dev] = cus;
  return cus;
}

}  // namespace

size_t pd2_xbuf_bytes(int M, int D, int H, int hd, int F, int esz, int cus) {
  const Pd2Geom geo(M, D, H, hd, esz, cus);
  return Pd2Xbuf(M, D, F, H, hd, esz, geo).bytes();
}

template <typename T>
bool pd2_ok(int M, int D, int H, int hd, int F, int cus) {
  constexpr int ESZ = (int)sizeof(T), KS = 64 / ESZ;
  if (cus < 64 || M < 1 || M > 16) return false;
  if (!(hd == 32 || hd == 64 || hd == 96 || hd == 128) || H * hd != D || (hd * ESZ) % 64 != 0) return false;
  if (D % 16 != 0 || F % 16 != 0 || D % KS != 0 || F % KS != 0 || (size_t)D * ESZ > 4096) return false;
  if (3 * D / 16 > cus || F / 16 > cus) return false;                            // one tile / slice per workgroup and phase
  if (2 * cdiv(D / KS, PD2_NW) > PD2_NF || cdiv(D / 16, PD2_NW) > PD2_NF) return false;   // the register set holds a phase's fragments
  const Pd2Geom geo(M, D, H, hd, ESZ, cus);
  if (!geo.ok) return false;
  const Pd2Xbuf xb(M, D, F, H, hd, ESZ, geo);
  if (xb.bytes() >= ((size_t)1 << 31)) return false;                             // 32-bit buffer offsets
  return Pd2Lds(D, ESZ, hd, geo.rg, geo.gs, geo.cw).total <= 159 * 1024;
}
template bool pd2_ok<float>(int, int, int, int, int, int);
template bool pd2_ok<bf16>(int, int, int, int, int, int);

namespace {
template <typename T, int HD, int VEC, int LPR>
int pd2_launch(PdArgs a, int G, hipStream_t st) {
  auto kern = pd2_layers_kernel<T, HD, VEC, LPR>;
  static LdsAttrOnce attr_once;
  VLG_TRY(set_max_dynamic_lds(attr_once, {reinterpret_cast<const void*>(kern)}, 159 * 1024));
  const Pd2Geom geo(a.M, a.D, a.H, HD, (int)sizeof(T), G);
  size_t ldsb = Pd2Lds(a.D, (int)sizeof(T), HD, geo.rg, geo.gs, geo.cw).total;
  if (ldsb < 84 * 1024) ldsb = 84 * 1024;   // one workgroup per compute unit (the hand-off forms are measured for that; correctness does not depend on it)
  kern<<<G, PD2_NTHR, ldsb, st>>>(a);
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}
}  // namespace

template <typename T>
int pd2_layers(PdArgs a, hipStream_t st) {
  const int G = pd2_cu_count();
  if (!a.fm || !pd2_ok<T>(a.M, a.D, a.H, a.hd, a.F, G)) {
    set_error("pd2_layers: shape M=%d D=%d H=%d hd=%d F=%d not covered (fragment-major weights: %d)", a.M, a.D, a.H, a.hd, a.F, a.fm);
    return VLG_ERR_UNSUPPORTED;
  }
  a.xbuf_bytes = (unsigned)pd2_xbuf_bytes(a.M, a.D, a.H, a.hd, a.F, (int)sizeof(T), G);
  if (a.spin_max <= 0) a.spin_max = 1 << 20;
#define PD2_GO(HD_, VEC_, LPR_) return pd2_launch<T, HD_, VEC_, LPR_>(a, G, st)
  if constexpr (sizeof(T) == 2) {
    if (a.hd == 64) PD2_GO(64, 8, 8);
    if (a.hd == 128) PD2_GO(128, 8, 16);
    if (a.hd == 96) PD2_GO(96, 8, 16);
    if (a.hd == 32) PD2_GO(32, 8, 4);
  } else {
    if (a.hd == 64) PD2_GO(64, 4, 16);
    if (a.hd == 128) PD2_GO(128, 4, 32);
    if (a.hd == 96) PD2_GO(96, 4, 32);
    if (a.hd == 32) PD2_GO(32, 4, 8);
  }
#undef PD2_GO
  set_error("pd2_layers: head_dim %d", a.hd);
  return VLG_ERR_UNSUPPORTED;
}
template int pd2_layers<float>(PdArgs, hipStream_t);
template int pd2_layers<bf16>(PdArgs, hipStream_t);

}  // namespace vlg
