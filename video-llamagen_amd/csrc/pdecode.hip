// Persistent decode step (round 3): the L transformer layers of ONE decode step (Tq = 1, gpt.py:255-259 x n_layer) in ONE launch.
//
// Replaces, for small row counts, the 6-launches-per-layer chain of gpt.hip::layers_fused (QKV GEMM, split-KV attention, combine,
// wo GEMM, w13 GEMM, w2 GEMM): at <= 32 rows those launches are bound by their boundaries and cold starts, not by HBM (DESIGN.md
// section 5).  One workgroup per compute unit stays resident for the whole step; a layer is five PHASES and the workgroups hand
// activations to each other through global memory INSIDE the launch:
//
//   QKV   WG t owns the 16-column tile t of wqkv: RMSNorm(x) (every WG holds the full rows: it needs them as the A operand anyway),
//         MFMA, RoPE, publishes its q | k | v columns and appends k / v to the cache for later steps        (gpt.py:215-227,182-183)
//   ATT   WG i owns work item i = (row, head, KV split): split-KV online-softmax attention over cache rows 0..p-1 (written by EARLIER
//         launches) + the row of the current position taken from the QKV hand-off; the owner split merges the others (gpt.py:230-237)
//   WO    tile t of wo on the attention rows, + residual                                                      (gpt.py:257)
//   W13   f-tile t of [w1; w3]: RMSNorm(h), two MFMA tiles, silu(a) * b                                        (gpt.py:166-167)
//   W2    tile t of w2 (K = F, staged through LDS in chunks), + residual -> next layer's x                     (gpt.py:258)
//
// Hand-off = flag-in-data granules (cdna_hip_programming.md Guideline 16 R2, as diffloss_persist.hip): every 4 payload bytes travel
// with the phase's tag in ONE 8-byte write-through (sc1) store; consumers sweep the granules with 16-byte sc1 loads (past their L1)
// and repeat a load until both tags match.  No flags, no fences, no store drain; placement-independent.  Tag = step * 8L + 8 layer +
// edge + 1 with `step` from StepState (it grows with every decode step of a generate() call; the buffer is zeroed when a call starts),
// regions are double-buffered by layer parity: a workgroup can run at most one layer ahead of any workgroup whose data it needs.
// Every wait is bounded; a wait that runs out sets the handle's fault word (vlg_gpt_status) and the workgroup leaves.
//
// Weights go straight from HBM to the MFMA B-operand registers (a decode weight byte is used once: no LDS round trip): the moment a
// GEMM phase has consumed its fragments, the NEXT GEMM phase's are requested into the same registers, so the stream runs under the
// epilogue, the publish and the hand-off wait in between (and under the whole attention phase for wo).  Activations
// are staged in LDS (rows padded by 32 bytes: conflict-free ds_read_b128 A fragments).  Rounding points are those of gemm_fused.hip /
// gpt_kernels.hip (the launch chain), so both paths agree up to fp32 summation order.
#include <algorithm>
#include <type_traits>

#include "gpt_kernels.h"
#include "pd_common.h"

namespace vlg {

namespace {

// LDS carve-up, shared by kernel and launcher
struct PdLds {
  unsigned a_stride, as, red, resid, wpart, att, flags, total;
  __host__ __device__ PdLds(int MT, int D, int kchunk, int esz, int hd) {
    const int kc = D > kchunk ? D : kchunk;
    a_stride = (unsigned)kc * esz + 32;                     // + 32 bytes: the 16 rows of a ds_read_b128 A fragment fall on 16 different slots
    unsigned o = 0;
    as = o; o += 16u * MT * a_stride;
    red = o; o += (unsigned)PD_NW * MT * 2 * 256 * 4;        // per-wave partial accumulators [wave][mt][tile][256]
    resid = o; o += (unsigned)MT * 256 * 4;                  // residual columns of the tile this workgroup finishes [mt][256]
    wpart = o; o += (unsigned)(PD_MAXKS - 1) * MT * 256 * 4; // the other K slices' partial tiles (owner of a wo / w2 tile)
    att = o; o += 3u * 128 * 4 + (unsigned)(PD_NW + PD_MAXNS) * (hd + 2) * 4 + 64;
    flags = o; o += 64;
    total = (o + 15u) & ~15u;
  }
};

#ifdef VLG_PD_PROF   // in-kernel time stamps of layer 1, kept in LDS until the end (tools/microbench/pd_lab.hip)
#define PD_STAMP(id)                                        \
  do {                                                      \
    if (tid == 0 && l == 1) prof_s[id] = wall_clock64();    \
  } while (0)
#else
#define PD_STAMP(id) \
  do {               \
  } while (0)
#endif

template <typename T, int MT, int HD, int VEC, int LPR>
__global__ __launch_bounds__(PD_NTHR) void pd_layers_kernel(PdArgs a) {
  constexpr int ESZ = (int)sizeof(T);
  constexpr int EPV = 16 / ESZ;        // elements per 16-byte chunk
  constexpr int KS = 64 / ESZ;         // K elements per MFMA step of one fragment (32 bf16 / 16 fp32)
  constexpr int EPG = 4 / ESZ;         // elements per granule
  extern __shared__ __attribute__((aligned(16))) char pd_smem[];
  const int M = a.M, D = a.D, F = a.F, H = a.H, L = a.L;
  const PdLds lds(MT, D, a.kchunk, ESZ, HD);
  char* As = pd_smem + lds.as;
  float* red = reinterpret_cast<float*>(pd_smem + lds.red);
  float* resid = reinterpret_cast<float*>(pd_smem + lds.resid);
  float* wpart = reinterpret_cast<float*>(pd_smem + lds.wpart);
  T* qkv_s = reinterpret_cast<T*>(pd_smem + lds.att);                               // [3][128] q | k | v of the current item
  float* sm = reinterpret_cast<float*>(pd_smem + lds.att + 3 * 128 * 4);            // [NW][HD + 2] per-wave (m, l, acc)
  float* part = sm + PD_NW * (HD + 2);                                              // [MAXNS][HD + 2] split partials (owner)
  int* flags = reinterpret_cast<int*>(pd_smem + lds.flags);                         // [0] alive
  const unsigned a_stride = lds.a_stride;
#ifdef VLG_PD_PROF
  __shared__ unsigned long long prof_s[32];
  if (threadIdx.x < 32) prof_s[threadIdx.x] = 0;
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int G = gridDim.x, wg = blockIdx.x;
  const int pos = a.state->pos;
  const unsigned eb = (unsigned)a.state->step * (unsigned)(8 * L);
  const PdXbuf xb(M, D, F, H, HD, ESZ, a.ns_max, a.ksplit);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, (int)a.xbuf_bytes, 0x00020000);
  const T* xg = reinterpret_cast<const T*>(a.x);
  const pd_layer_cptr layers_c = (pd_layer_cptr)(uintptr_t)a.layers;

  // A fault word that is already set (an earlier step of this call timed out, and the host has enqueued the remaining steps long ago): leave
  // at once instead of spinning every wait of every remaining step to its bound - the call is lost either way (vlg_gpt_status).
  if (tid == 0) flags[0] = (a.fault == nullptr || __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) ? 1 : 0;
  for (unsigned i = tid; i < 16u * MT * a_stride / 16; i += PD_NTHR) reinterpret_cast<pd_u32x4_t*>(As)[i] = pd_u32x4_t{0u, 0u, 0u, 0u};
  pd_barrier();
  if (flags[0] == 0) return;

  // ---- work assignment (one tile or unit per workgroup and phase: pd_ok) -------------------------------------------------------------
  // QKV: 16-column tile wg of wqkv.  W13: f-tile wg of [w1; w3].  WO / W2: the D / 16 output tiles are cut into `ksplit` K slices so that
  // (almost) every workgroup streams a share of wo / w2: unit wg = (tile wg / ksplit, slice wg % ksplit); slice 0 owns the tile - it adds
  // the other slices' fp32 partials in slice order (deterministic), then the residual, and publishes.
  const int KSP = a.ksplit;
  const bool has_qkv = wg < 3 * D / 16, has_f = wg < F / 16, has_d = wg < (D / 16) * KSP;
  const int dtile = wg / KSP, dslice = wg - dtile * KSP;
  const bool d_owner = has_d && dslice == 0;
  const int nks_d = D / KS, nks_f = F / KS;                      // K steps of the D- and F-deep GEMMs
  const int ksw_d = (nks_d - wave + PD_NW - 1) / PD_NW;          // K steps of this wave in a full-depth GEMM over D
  const int wo_lo = nks_d * dslice / KSP, wo_hi = nks_d * (dslice + 1) / KSP;     // K steps of this workgroup's wo slice
  const int w2_lo = nks_f * dslice / KSP, w2_hi = nks_f * (dslice + 1) / KSP;     // ... and of its w2 slice
  const int ksw_wo = (wo_hi - wo_lo - wave + PD_NW - 1) / PD_NW, ksw_w2 = (w2_hi - w2_lo - wave + PD_NW - 1) / PD_NW;

  // ---- hand-off primitives ---------------------------------------------------------------------------------------------------------
  // rows [0, M) x columns [c0, c1) of the tagged [M][N] matrix at granule offset gbase -> dst (row stride dstride bytes, column c0 at byte 0)
  auto sweep = [&](char* dst, unsigned dstride, unsigned gbase, int N, int c0, int c1, unsigned tag) __attribute__((always_inline)) {
    const int ppr = (c1 - c0) * ESZ / 8;                          // granule pairs (16 bytes of granules = 8 payload bytes) per row
    const int npair = M * ppr;
    for (int i0 = tid; i0 < npair; i0 += PD_NTHR * PD_UB) {
      pd_u32x4_t v[PD_UB];
      int goff[PD_UB];
      bool need[PD_UB];
#pragma unroll
      for (int u = 0; u < PD_UB; ++u) {
        const int i = i0 + u * PD_NTHR;
        const int ii = i < npair ? i : npair - 1;
        const int row = ii / ppr, wi = ii - row * ppr;
        goff[u] = (int)((gbase + (unsigned)((row * N + c0) * ESZ / 4)) * 8u) + wi * 16;
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[u], 0, 16);
      }
      for (int spin = 0;; ++spin) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < PD_UB; ++u) {
          need[u] = (i0 + u * PD_NTHR < npair) && (v[u][1] != tag || v[u][3] != tag);
          any = any || need[u];
        }
        if (!any) break;
        if (spin >= a.spin_max) {
          flags[0] = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int u = 0; u < PD_UB; ++u)
          if (need[u]) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[u], 0, 16);
      }
#pragma unroll
      for (int u = 0; u < PD_UB; ++u) {
        const int i = i0 + u * PD_NTHR;
        if (i < npair) {
          const int row = i / ppr, wi = i - row * ppr;
          *reinterpret_cast<pd_u32x2_t*>(dst + (size_t)row * dstride + wi * 8) = pd_u32x2_t{v[u][0], v[u][2]};
        }
      }
    }
  };
  // `n` consecutive fp32 granules (n even) at granule offset gbase -> dst[0..n)
  auto sweep_f32 = [&](float* dst, unsigned gbase, int n, unsigned tag) __attribute__((always_inline)) {
    for (int i = tid; i < n / 2; i += PD_NTHR) {
      const int goff = (int)(gbase * 8u) + i * 16;
      pd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 16);
      for (int spin = 0; v[1] != tag || v[3] != tag; ++spin) {
        if (spin >= a.spin_max) {
          flags[0] = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 16);
      }
      dst[2 * i] = __uint_as_float(v[0]);
      dst[2 * i + 1] = __uint_as_float(v[2]);
    }
  };
  // one value per thread -> granule of element (row, col) of a tagged [M][N] matrix; bf16: the even-column lane stores the pair
  // (its odd neighbour sits in the next lane)
  auto publish = [&](unsigned gbase, int N, int row, int col, float v, bool valid, unsigned tag) __attribute__((always_inline)) {
    if constexpr (ESZ == 2) {
      const float vn = __shfl_down(v, 1);
      if (valid && !(col & 1))
        __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{pd_pack2(v, vn), tag}, rs, (int)((gbase + (unsigned)(row * N + col) / 2) * 8u), 0, 16);
    } else {
      if (valid) __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{__float_as_uint(v), tag}, rs, (int)((gbase + (unsigned)(row * N + col)) * 8u), 0, 16);
    }
  };
  // the 16 columns [c0, c0 + 16) of the rows of a tagged [M][D] matrix (or of the plain layer-0 input) -> resid[mt][256] in accumulator order
  auto fetch_resid = [&](unsigned gbase, int c0, unsigned tag, bool plain) __attribute__((always_inline)) {
    if (plain) {
      if (tid < 256) {
        const int e = tid >> 6, l2 = tid & 63;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = mt * 16 + (l2 >> 4) * 4 + e;
          resid[mt * 256 + tid] = row < M ? DT<T>::ld(xg + (size_t)row * D + c0 + (l2 & 15)) : 0.f;
        }
      }
    } else {
      // 16 columns = 16 * ESZ / 8 granule pairs per row; pair i of row -> elements [i * 8 / ESZ, ...)
      constexpr int PPR = 16 * ESZ / 8, EPP = 8 / ESZ;
      for (int i = tid; i < M * PPR; i += PD_NTHR) {
        const int row = i / PPR, wi = i - row * PPR;
        const int goff = (int)((gbase + (unsigned)((row * D + c0) * ESZ / 4)) * 8u) + wi * 16;
        pd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 16);
        for (int spin = 0; v[1] != tag || v[3] != tag; ++spin) {
          if (spin >= a.spin_max) {
            flags[0] = 0;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 16);
        }
        const int mt = row >> 4, rr = row & 15;
        float* dstp = resid + mt * 256 + (rr & 3) * 64 + (rr >> 2) * 16 + wi * EPP;   // accumulator order: idx = e * 64 + (row / 4) * 16 + col, e = row % 4
        if constexpr (ESZ == 2) {
          dstp[0] = pd_lo(v[0]); dstp[1] = pd_hi(v[0]); dstp[2] = pd_lo(v[2]); dstp[3] = pd_hi(v[2]);
        } else {
          dstp[0] = __uint_as_float(v[0]); dstp[1] = __uint_as_float(v[2]);
        }
      }
    }
  };

  // ---- weights: one register set, refilled for the next GEMM phase as soon as the current one has consumed it -----------------------------
  // fragment i = j * NTW + t: K step kk_lo + wave + 8 j of weight tile t (first weight row row0 + t * row_step).  Address = uniform tile
  // base (scalar) + ONE per-lane byte offset per K (voff_d / voff_f) + a compile-time j * 512.
  pd_u32x4_t R[PD_NF];
  const unsigned voff_d = (unsigned)((r * D + wave * KS) * ESZ + q * 16), voff_f = (unsigned)((r * F + wave * KS) * ESZ + q * 16);
  // Fragment-major weights (a.fm, gpt_kernels.h: relayout_fragment_major): the fragment of (tile, K step kk) is the 1 KB block
  // (tile * K / KS + kk), this lane's 16 bytes at lane * 16 - a wave instruction reads 8 whole cache lines instead of 16 half lines.
  const bool fm = a.fm != 0;
  const unsigned voff_fm = (unsigned)wave * 1024u + (unsigned)lane * 16u;
  auto load_set = [&](const void* wv, int K, int row0, int row_step, int kk_lo, int ksw, auto ntw_c) __attribute__((always_inline)) {
    constexpr int NTW = decltype(ntw_c)::value;
    // two copies of the loop so that the per-fragment offsets stay compile-time constants in both layouts (a run-time stride made hipcc
    // keep per-fragment addresses and spill them to scratch - and a scratch reload queues behind the weight stream)
    if (fm) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const char* tb = reinterpret_cast<const char*>(wv) + ((size_t)((row0 + t * row_step) >> 4) * (size_t)(K / KS) + (size_t)kk_lo) * 1024;   // wave-uniform
#pragma unroll
        for (int j = 0; j < PD_NF / NTW; ++j)
          if (j < ksw) R[j * NTW + t] = __builtin_nontemporal_load((pd_gptr16)(uintptr_t)(tb + voff_fm + j * (PD_NW * 1024)));
      }
    } else {
      const unsigned voff = (K == D) ? voff_d : voff_f;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const char* tb = reinterpret_cast<const char*>(wv) + ((size_t)(row0 + t * row_step) * K + (size_t)kk_lo * KS) * ESZ;   // wave-uniform
#pragma unroll
        for (int j = 0; j < PD_NF / NTW; ++j)
          if (j < ksw) R[j * NTW + t] = __builtin_nontemporal_load((pd_gptr16)(uintptr_t)(tb + voff + j * (PD_NW * 64)));
      }
    }
  };
  // acc[mt][t] += A[rows of mt][this wave's K steps in [kk0, kk1)] . R      (A column of K step kk: (kk - kk0) * KS; kk = kk_lo + wave + 8 j)
  auto gemm = [&](pd_f32x4_t (&acc)[MT][2], int kk_lo, int ksw, int kk0, int kk1, auto ntw_c) __attribute__((always_inline)) {
    constexpr int NTW = decltype(ntw_c)::value;
    // per-lane LDS offset of the A fragments, made opaque per call: otherwise hipcc hoists one address per (phase, fragment) out of the
    // layer loop, runs out of registers and parks them in scratch - and a scratch reload queues behind the weight stream
    unsigned abase = (unsigned)r * a_stride + (unsigned)q * 16u + (unsigned)(wave + kk_lo - kk0) * (unsigned)(KS * ESZ);
    asm volatile("" : "+v"(abase));
#pragma unroll
    for (int j = 0; j < PD_NF / NTW; ++j) {
      const int kk = kk_lo + wave + PD_NW * j;
      if (j < ksw && kk >= kk0 && kk < kk1) {
        pd_u32x4_t af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          af[mt] = *reinterpret_cast<const pd_u32x4_t*>(As + abase + (unsigned)(mt * 16) * a_stride + (unsigned)(j * PD_NW * KS * ESZ));
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pd_mfma<T>(af[mt], R[j * NTW + t], acc[mt][t]);
      }
    }
  };
  // After a phase's last MFMA the fragments are dead, but the next (conditional, guarded) prefetch only overwrites part of the set: without
  // an explicit kill the compiler keeps the old values live across the attention phase and spills them - and a spill of a just-requested
  // fragment is a wait for the HBM stream.  Zeros are rematerialisable, so this costs no registers.
  auto kill_set = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PD_NF; ++i) R[i] = pd_u32x4_t{0u, 0u, 0u, 0u};
  };
  auto zero_acc = [&](pd_f32x4_t (&acc)[MT][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[mt][t] = pd_f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  auto store_red = [&](const pd_f32x4_t (&acc)[MT][2], int ntw) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (t < ntw) {
#pragma unroll
          for (int e = 0; e < 4; ++e) red[((wave * MT + mt) * 2 + t) * 256 + e * 64 + lane] = acc[mt][t][e];
        }
  };
  // sum over the 8 waves of element idx (0..255: e * 64 + lane of the accumulator layout) of (mt, tile t)
  auto reduced = [&](int mt, int t, int idx) __attribute__((always_inline)) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < PD_NW; ++wv) s += red[((wv * MT + mt) * 2 + t) * 256 + idx];
    return s;
  };
  // RMSNorm weight chunks of this lane (requested BEFORE the sweep that precedes the norm: the fetch latency hides under the hand-off)
  auto load_norm_w = [&](pd_u32x4_t (&gv)[4], const void* gw) __attribute__((always_inline)) {
    const int nch = D / EPV;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = lane + 64 * it;
      gv[it] = ((pd_gptr16)(uintptr_t)gw)[c < nch ? c : 0];
    }
  };
  // in-place RMSNorm of the M rows of As (K = D): rt(rt(x * rsqrt(mean(x^2) + eps)) * g)      (gpt.py:143-148).  A wave takes rows w, w + 8
  // two at a time (independent chains interleave), statistics by a DPP wave sum in a fixed order.
  auto rmsnorm = [&](const pd_u32x4_t (&gv)[4]) __attribute__((always_inline)) {
    const int nch = D / EPV;
    for (int row0 = wave; row0 < M; row0 += 2 * PD_NW) {
      pd_u32x4_t xv[2][4];
      float ss[2] = {0.f, 0.f};
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = row0 + rr * PD_NW < M ? row0 + rr * PD_NW : row0;
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
        const int row = row0 + rr * PD_NW;
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
#define PD_SYNC_ALIVE()        \
  do {                         \
    pd_barrier();           \
    alive = flags[0] != 0;     \
  } while (0)

  // Finish a wo / w2 tile: `red` holds this workgroup's K-slice partial.  Other slices publish theirs (fp32), the owner adds them in
  // slice order, then the residual: v = rt(resid + rt(sum))  (gemm_fused.hip EPI_RESID); returns false when a wait ran out.
  // out(mt, t256, row, col, v) consumes the owner's values (called by threads < 256 for every mt).
  auto finish_d_tile = [&](unsigned preg, unsigned ptag, auto&& out) __attribute__((always_inline)) -> bool {
    const int t256 = tid & 255;
    if (!d_owner) {
      if (tid < 256) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{__float_as_uint(reduced(mt, 0, t256)), ptag}, rs,
                                                (int)((preg + (unsigned)(((dtile * KSP + dslice) * MT + mt) * 256 + t256)) * 8u), 0, 16);
      }
      return true;
    }
    if (KSP > 1) sweep_f32(wpart, preg + (unsigned)((dtile * KSP + 1) * MT * 256), (KSP - 1) * MT * 256, ptag);
    pd_barrier();
    if (flags[0] == 0) return false;
    if (tid < 256) {
      const int e = t256 >> 6, l2 = t256 & 63;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float sacc = reduced(mt, 0, t256);
        for (int sl = 1; sl < KSP; ++sl) sacc += wpart[((sl - 1) * MT + mt) * 256 + t256];
        const int row = mt * 16 + (l2 >> 4) * 4 + e, col = dtile * 16 + (l2 & 15);
        out(mt, t256, row, col, DT<T>::rt(resid[mt * 256 + t256] + DT<T>::rt(sacc)));
      }
    }
    return true;
  };

  // first weights: qkv of layer 0
  if (has_qkv) load_set(layers_c[0].wqkv, D, wg * 16, 0, 0, ksw_d, std::integral_constant<int, 1>{});

  const float att_scale = 1.0f / sqrtf((float)HD);
  for (int l = 0; l < L && alive; ++l) {
    struct {
      const void *wo, *w13, *w2, *norm1, *norm2;
    } ly = {layers_c[l].wo, layers_c[l].w13, layers_c[l].w2, layers_c[l].norm1, layers_c[l].norm2};
    const int par = l & 1;
    const unsigned tag0 = eb + (unsigned)(8 * l) + 1u;   // + edge: 0 x, 1 qkv, 2 ao, 3 h, 4 g, 5 attention partials, 6 wo partials, 7 w2 partials
    pd_f32x4_t acc[MT][2];

    // =========================== QKV ===========================
    PD_STAMP(0);
    if (has_qkv) {
      // RoPE pair of this thread's output column at the step's position (epilogue operand, requested early)
      const int ecol = wg * 16 + (tid & 15);
      const int esec = ecol / D, ewithin = ecol - esec * D;
      const int ehh = ewithin / HD, ed = ewithin - ehh * HD;
      const float* cp = a.freqs + ((size_t)pos * (HD / 2) + ed / 2) * 2;
      const float cx = cp[0], cy = cp[1];
      pd_u32x4_t gv[4];
      load_norm_w(gv, ly.norm1);
      if (l == 0) {
        const int nch = D / EPV;
        for (int i = tid; i < M * nch; i += PD_NTHR) {
          const int row = i / nch, c = i - row * nch;
          *reinterpret_cast<pd_u32x4_t*>(As + (size_t)row * a_stride + (size_t)c * 16) = reinterpret_cast<const pd_u32x4_t*>(xg + (size_t)row * D)[c];
        }
      } else {
        sweep(As, a_stride, xb.X(par), D, 0, D, tag0 + 0u);
      }
      PD_SYNC_ALIVE();
      if (!alive) break;
      PD_STAMP(1);
      rmsnorm(gv);
      PD_STAMP(18);
      pd_barrier();
      PD_STAMP(19);
      zero_acc(acc);
      gemm(acc, 0, ksw_d, 0, nks_d, std::integral_constant<int, 1>{});
      kill_set();
      store_red(acc, 1);
      PD_STAMP(2);
      pd_barrier();
      // epilogue: RoPE on adjacent pairs, publish q | k | v, append k / v to the cache (gemm_fused.hip EPI_QKV)
      if (tid < 256) {
        const int e = tid >> 6, l2 = tid & 63;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = mt * 16 + (l2 >> 4) * 4 + e;
          const float xs = DT<T>::rt(reduced(mt, 0, tid)), xp = DT<T>::rt(reduced(mt, 0, tid ^ 1));
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
    }
    PD_STAMP(3);

    // =========================== ATT ===========================
    {
      const int nkeys = pos + 1;
      // KV splits: fewest rounds x (fixed cost per item + keys per item), the fixed cost priced at 384 keys
      int ns = 1;
      {
        int best = 0x7fffffff;
        const int cap = min(a.ns_max, (nkeys + 63) / 64);
        for (int c2 = 1; c2 <= cap; ++c2) {
          const int cost = ((M * H * c2 + G - 1) / G) * (384 + (nkeys + c2 - 1) / c2);
          if (cost < best) {
            best = cost;
            ns = c2;
          }
        }
      }
      const int nitems = M * H * ns;
      constexpr int RPI = 64 / LPR, U = 8, TILE = RPI * U;
      const int g = lane / LPR, c = lane % LPR;
      const bool active = c * VEC < HD;
      const int coff = active ? c * VEC : 0;
      for (int it = wg; it < nitems && alive; it += G) {
        const int split = it % ns, mh = it / ns;
        const int m = mh / H, h = mh - m * H;
        const int chunk = (nkeys + ns - 1) / ns;
        const int r0 = split * chunk;
        const int r1 = min(r0 + chunk, nkeys);
        // rows of the cache (all < pos: row `pos` is appended by this launch with plain stores, for later steps, and never read here)
        const int r1c = min(r1, pos);
        // wave-uniform row base + a 32-bit per-lane byte offset: one address register per load
        const char* kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.kc) + (size_t)l * a.kv_lstride + ((size_t)m * H + h) * (size_t)a.S * HD);
        const char* vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.vc) + (size_t)l * a.kv_lstride + ((size_t)m * H + h) * (size_t)a.S * HD);
        // (Requesting the first tile BEFORE the wait for the QKV hand-off would overlap the two latencies, but keeps 64 registers live across
        // the sweep: hipcc then spills, and a scratch reload queues behind the weight stream - measured slower.)
        // q | k | v of (m, h) from the QKV hand-off -> LDS
        {
          const int ppr = HD * ESZ / 8, npair = 3 * ppr;
          for (int i = tid; i < npair; i += PD_NTHR) {
            const int sec = i / ppr, wi = i - sec * ppr;
            const int goff = (int)((xb.Q(par) + (unsigned)((m * 3 * D + sec * D + h * HD) * ESZ / 4)) * 8u) + wi * 16;
            pd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 16);
            for (int spin = 0; v[1] != tag0 + 1u || v[3] != tag0 + 1u; ++spin) {
              if (spin >= a.spin_max) {
                flags[0] = 0;
                break;
              }
              __builtin_amdgcn_s_sleep(1);
              v = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 16);
            }
            *reinterpret_cast<pd_u32x2_t*>(reinterpret_cast<char*>(qkv_s) + sec * 128 * 4 + wi * 8) = pd_u32x2_t{v[0], v[2]};
          }
        }
        PD_SYNC_ALIVE();
        if (!alive) break;
        PD_STAMP(4);
        float qf[VEC];
        PdPack<T, VEC> knew, vnew;
        {
          const T* qs = qkv_s;
          const T* ks = reinterpret_cast<const T*>(reinterpret_cast<const char*>(qkv_s) + 128 * 4);
          const T* vs = reinterpret_cast<const T*>(reinterpret_cast<const char*>(qkv_s) + 2 * 128 * 4);
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
        for (int tile = r0 + wave * TILE; tile < r1c; tile += PD_NW * TILE) {
          PdPack<T, VEC> kk[U], vv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = tile + u * RPI + g;
            const int rr = row < r1c ? row : r1c - 1;
            const unsigned lo = (unsigned)(rr * HD + coff) * ESZ;
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
        // the key of the current position (never masked: generate.py:156-165 leaves the diagonal on), taken from the QKV hand-off by lane
        // group 0 of wave 0 of the split whose range holds it
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
        PD_STAMP(5);
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
        if (ns == 1) {
          // one split: merge the 8 waves and publish this thread's granule of the output row right away
          if (tid < HD / EPG) {
            float M8 = -INFINITY;
#pragma unroll
            for (int wv = 0; wv < PD_NW; ++wv) M8 = fmaxf(M8, sm[wv * (HD + 2)]);
            const float mref = (M8 == -INFINITY) ? 0.f : M8;
            float Ls = 0.f, A0 = 0.f, A1 = 0.f;
#pragma unroll
            for (int wv = 0; wv < PD_NW; ++wv) {
              const float ee = __expf(sm[wv * (HD + 2)] - mref);
              Ls += sm[wv * (HD + 2) + 1] * ee;
              A0 += sm[wv * (HD + 2) + 2 + tid * EPG] * ee;
              if constexpr (EPG == 2) A1 += sm[wv * (HD + 2) + 2 + tid * EPG + 1] * ee;
            }
            const unsigned gidx = xb.AO(par) + (unsigned)((m * D + h * HD) / EPG + tid);
            if constexpr (EPG == 2)
              __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{pd_pack2(A0 / Ls, A1 / Ls), tag0 + 2u}, rs, (int)(gidx * 8u), 0, 16);
            else
              __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{__float_as_uint(A0 / Ls), tag0 + 2u}, rs, (int)(gidx * 8u), 0, 16);
          }
          if (it + G < nitems) pd_barrier();   // sm / qkv_s are reused by the next item
          continue;
        }
        // this split's (m, l, acc[HD]) -> part[0] (owner) or the partial hand-off (other splits)
        if (tid < HD + 2) {
          float M8 = -INFINITY;
#pragma unroll
          for (int wv = 0; wv < PD_NW; ++wv) M8 = fmaxf(M8, sm[wv * (HD + 2)]);
          const float mref = (M8 == -INFINITY) ? 0.f : M8;
          float val;
          if (tid == 0) {
            val = M8;
          } else {
            val = 0.f;
#pragma unroll
            for (int wv = 0; wv < PD_NW; ++wv) val += sm[wv * (HD + 2) + tid] * __expf(sm[wv * (HD + 2)] - mref);
          }
          if (split == 0)
            part[tid] = val;
          else
            __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{__float_as_uint(val), tag0 + 5u}, rs,
                                                  (int)((xb.AP(par) + (unsigned)((mh * a.ns_max + split) * (HD + 2) + tid)) * 8u), 0, 16);
        }
        if (split == 0) {
          if (ns > 1) sweep_f32(part + (HD + 2), xb.AP(par) + (unsigned)((mh * a.ns_max + 1) * (HD + 2)), (ns - 1) * (HD + 2), tag0 + 5u);
          PD_SYNC_ALIVE();
          if (!alive) break;
          if (tid < HD / EPG) {   // merge in split order (attn_combine_kernel) and publish this thread's granule of the output row
            float Mx = -INFINITY;
            for (int sp = 0; sp < ns; ++sp) Mx = fmaxf(Mx, part[sp * (HD + 2)]);
            const float mref = (Mx == -INFINITY) ? 0.f : Mx;
            float Ls = 0.f, A0 = 0.f, A1 = 0.f;
            for (int sp = 0; sp < ns; ++sp) {
              const float ee = __expf(part[sp * (HD + 2)] - mref);
              Ls += part[sp * (HD + 2) + 1] * ee;
              A0 += part[sp * (HD + 2) + 2 + tid * EPG] * ee;
              if constexpr (EPG == 2) A1 += part[sp * (HD + 2) + 2 + tid * EPG + 1] * ee;
            }
            const unsigned gidx = xb.AO(par) + (unsigned)((m * D + h * HD) / EPG + tid);
            if constexpr (EPG == 2)
              __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{pd_pack2(A0 / Ls, A1 / Ls), tag0 + 2u}, rs, (int)(gidx * 8u), 0, 16);
            else
              __builtin_amdgcn_raw_buffer_store_b64(pd_u32x2_t{__float_as_uint(A0 / Ls), tag0 + 2u}, rs, (int)(gidx * 8u), 0, 16);
          }
        }
        pd_barrier();   // sm / part / qkv_s are reused by the next item
      }
      if (!alive) break;
    }
    PD_STAMP(6);

    // =========================== WO ===========================
    if (has_d) {
      // this workgroup's wo slice (a few KB): requested here, not before the attention phase, so that no weight register is live across
      // it (the K / V tiles need them); the latency hides under the wait for the attention rows.  A request can stall the issuing waves
      // for ~1 us (a compute unit holds only ~64 KB of outstanding misses), which is why the other prefetches sit AFTER their phase's publish.
      load_set(ly.wo, D, dtile * 16, 0, wo_lo, ksw_wo, std::integral_constant<int, 1>{});
      if (d_owner) fetch_resid(xb.X(par), dtile * 16, tag0 + 0u, l == 0);
      sweep(As, a_stride, xb.AO(par), D, wo_lo * KS, wo_hi * KS, tag0 + 2u);
      PD_SYNC_ALIVE();
      if (!alive) break;
      PD_STAMP(7);
      zero_acc(acc);
      gemm(acc, wo_lo, ksw_wo, wo_lo, wo_hi, std::integral_constant<int, 1>{});
      kill_set();
      PD_STAMP(15);
      store_red(acc, 1);
    }
    PD_STAMP(16);
    if (has_d) {
      pd_barrier();
      PD_STAMP(17);
      const bool okf = finish_d_tile(xb.WP(par), tag0 + 6u, [&](int mt, int t256, int row, int col, float v) {
        publish(xb.HH(par), D, row, col, v, row < M, tag0 + 3u);
      });
      if (!okf) {
        alive = false;
        break;
      }
    }
    // the set is free: w13 of this layer
    if (has_f) load_set(ly.w13, D, wg * 16, F, 0, ksw_d, std::integral_constant<int, 2>{});
    PD_STAMP(8);

    // =========================== W13 ===========================
    if (has_f) {
      pd_u32x4_t gv[4];
      load_norm_w(gv, ly.norm2);
      sweep(As, a_stride, xb.HH(par), D, 0, D, tag0 + 3u);
      PD_SYNC_ALIVE();
      if (!alive) break;
      PD_STAMP(9);
      rmsnorm(gv);
      pd_barrier();
      zero_acc(acc);
      gemm(acc, 0, ksw_d, 0, nks_d, std::integral_constant<int, 2>{});
      kill_set();
      store_red(acc, 2);
      PD_STAMP(10);
    }
    if (has_f) {
      pd_barrier();
      if (tid < 256) {   // g = rt(rt(silu(rt(a))) * rt(b))      (gemm_fused.hip EPI_SWIGLU)
        const int e = tid >> 6, l2 = tid & 63;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = mt * 16 + (l2 >> 4) * 4 + e, col = wg * 16 + (l2 & 15);
          const float av2 = DT<T>::rt(reduced(mt, 0, tid)), bv = DT<T>::rt(reduced(mt, 1, tid));
          publish(xb.G(par), F, row, col, DT<T>::rt(DT<T>::rt(pd_silu(av2)) * bv), row < M, tag0 + 4u);
        }
      }
    }
    // the set is free: this workgroup's w2 slice
    if (has_d) load_set(ly.w2, F, dtile * 16, 0, w2_lo, ksw_w2, std::integral_constant<int, 1>{});
    PD_STAMP(11);

    // =========================== W2 ===========================
    if (has_d) {
      if (d_owner) {
        pd_barrier();   // resid / wpart of the wo tile have been consumed
        fetch_resid(xb.HH(par), dtile * 16, tag0 + 3u, false);
      }
      zero_acc(acc);
      const int nkc = a.kchunk / KS;                   // K steps per staged chunk (a multiple of 8)
      for (int kk0 = w2_lo; kk0 < w2_hi; kk0 += nkc) {
        const int kk1 = min(kk0 + nkc, w2_hi);
        pd_barrier();                               // earlier A fragments (W13 / the previous chunk) have been read
        sweep(As, a_stride, xb.G(par), F, kk0 * KS, kk1 * KS, tag0 + 4u);
        PD_SYNC_ALIVE();
        if (!alive) break;
        PD_STAMP(12 + (kk0 > w2_lo ? 1 : 0));
        gemm(acc, w2_lo, ksw_w2, kk0, kk1, std::integral_constant<int, 1>{});
      }
      if (!alive) break;
      kill_set();
      store_red(acc, 1);
    }
    if (has_d) {
      pd_barrier();
      const bool last = l + 1 == L;
      const bool okf = finish_d_tile(xb.W2P(par), tag0 + 7u, [&](int mt, int t256, int row, int col, float v) {
        if (!last)
          publish(xb.X(1 - par), D, row, col, v, row < M, tag0 + 8u);   // = edge 0 of layer l + 1
        else if (row < M)
          DT<T>::st(reinterpret_cast<T*>(a.x) + (size_t)row * D + col, v);
      });
      if (!okf) {
        alive = false;
        break;
      }
      pd_barrier();   // red / resid / wpart are rewritten by the next layer
    }
    // the set is free: qkv of the next layer
    if (l + 1 < L && has_qkv) load_set(layers_c[l + 1].wqkv, D, wg * 16, 0, 0, ksw_d, std::integral_constant<int, 1>{});
    PD_STAMP(14);
  }
#undef PD_SYNC_ALIVE
#ifdef VLG_PD_PROF
  pd_barrier();
  if (a.prof && threadIdx.x < 32) a.prof[(size_t)blockIdx.x * 32 + threadIdx.x] = prof_s[threadIdx.x];
#endif
  // a wait ran out: tell the host (vlg_gpt_status); the word lives in pinned host memory
  if (!alive && tid == 0 && a.fault) __hip_atomic_store(a.fault, kFaultDecode | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int pd_cu_count() {
  static int cache[64] = {};
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev < 64 && cache[dev] > 0) return cache[dev];
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (dev < 64) cache[dev] = cus;
  return cus;
}

// K elements of the w2 GEMM staged per chunk: the largest multiple of 8 K steps that keeps 16 MT rows of it (and of D) within the budget
int pd_kchunk(int MT, int D, int F, int esz) {
  const int ks8 = 8 * (64 / esz);
  const int budget = (MT == 1 ? 64 : 88) * 1024;
  int kc = ((budget / (16 * MT) - 32) / esz) / ks8 * ks8;
  if (kc > F) kc = (F + ks8 - 1) / ks8 * ks8;
  if (kc < ks8) kc = ks8;
  return kc;
}
int pd_ksplit(int D, int G) {
  int ks = G / (D / 16);
  if (ks > PD_MAXKS) ks = PD_MAXKS;
  if (ks < 1) ks = 1;
  return ks;
}
int pd_ns_max(int M, int H, int G) {
  // upper bound of the KV splits (the kernel picks per step, by context length): up to 4 items per workgroup
  int ns = 4 * G / (M * H);
  if (ns > PD_MAXNS) ns = PD_MAXNS;
  if (ns < 1) ns = 1;
  return ns;
}

}  // namespace

size_t pd_xbuf_bytes(int M, int D, int H, int hd, int F, int esz) { return PdXbuf(M, D, F, H, hd, esz, PD_MAXNS, PD_MAXKS).bytes(); }

template <typename T>
bool pd_ok(int M, int D, int H, int hd, int F, int S, int cus) {
  constexpr int ESZ = (int)sizeof(T), KS = 64 / ESZ;
  if (cus < 64 || M < 1 || M > 32) return false;
  if (!(hd == 32 || hd == 64 || hd == 96 || hd == 100 || hd == 128) || H * hd != D) return false;
  if (D % 16 != 0 || F % 16 != 0 || D % KS != 0 || F % KS != 0 || (size_t)D * ESZ > 4096) return false;
  if ((hd * ESZ) % 8 != 0) return false;
  const int G = cus;
  if (3 * D / 16 > G || F / 16 > G) return false;                               // one tile per workgroup and phase
  const int ksp = pd_ksplit(D, G);
  const int ks_d = cdiv(D / KS, PD_NW), ks_w2 = cdiv(cdiv(F / KS, ksp), PD_NW);
  if (2 * ks_d > PD_NF || ks_w2 > PD_NF) return false;                           // the register set holds a phase's fragments
  const int MT = M > 16 ? 2 : 1;
  if (PdLds(MT, D, pd_kchunk(MT, D, F, ESZ), ESZ, hd).total > 159 * 1024) return false;
  (void)S;
  return true;
}
template bool pd_ok<float>(int, int, int, int, int, int, int);
template bool pd_ok<bf16>(int, int, int, int, int, int, int);

namespace {
template <typename T, int MT, int HD, int VEC, int LPR>
int pd_launch(PdArgs a, int G, hipStream_t st) {
  auto kern = pd_layers_kernel<T, MT, HD, VEC, LPR>;
  static bool attr[64] = {};
  int dev = 0;
  VLG_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr[dev]) {
    VLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    if (dev >= 0 && dev < 64) attr[dev] = true;
  }
  size_t ldsb = PdLds(MT, a.D, a.kchunk, (int)sizeof(T), HD).total;
  if (ldsb < 84 * 1024) ldsb = 84 * 1024;   // one workgroup per compute unit (the hand-off forms are measured for that; correctness does not depend on it)
  kern<<<G, PD_NTHR, ldsb, st>>>(a);
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}
}  // namespace

template <typename T>
int pd_layers(PdArgs a, hipStream_t st) {
  const int G = pd_cu_count();
  if (!pd_ok<T>(a.M, a.D, a.H, a.hd, a.F, a.S, G)) {
    set_error("pd_layers: shape M=%d D=%d H=%d hd=%d F=%d not covered", a.M, a.D, a.H, a.hd, a.F);
    return VLG_ERR_UNSUPPORTED;
  }
  const int MT = a.M > 16 ? 2 : 1;
  a.kchunk = pd_kchunk(MT, a.D, a.F, (int)sizeof(T));
  a.ns_max = pd_ns_max(a.M, a.H, G);
  a.ksplit = pd_ksplit(a.D, G);
  a.xbuf_bytes = (unsigned)pd_xbuf_bytes(a.M, a.D, a.H, a.hd, a.F, (int)sizeof(T));
  if (a.spin_max <= 0) a.spin_max = 1 << 20;
#define PD_GO(HD_, VEC_, LPR_)                                              \
  do {                                                                      \
    if (MT == 2) return pd_launch<T, 2, HD_, VEC_, LPR_>(a, G, st);         \
    return pd_launch<T, 1, HD_, VEC_, LPR_>(a, G, st);                      \
  } while (0)
  if constexpr (sizeof(T) == 2) {
    if (a.hd == 64) PD_GO(64, 8, 8);
    if (a.hd == 128) PD_GO(128, 8, 16);
    if (a.hd == 100) PD_GO(100, 4, 32);
    if (a.hd == 96) PD_GO(96, 8, 16);
    if (a.hd == 32) PD_GO(32, 8, 4);
  } else {
    if (a.hd == 64) PD_GO(64, 4, 16);
    if (a.hd == 128) PD_GO(128, 4, 32);
    if (a.hd == 100) PD_GO(100, 4, 32);
    if (a.hd == 96) PD_GO(96, 4, 32);
    if (a.hd == 32) PD_GO(32, 4, 8);
  }
#undef PD_GO
  set_error("pd_layers: head_dim %d", a.hd);
  return VLG_ERR_UNSUPPORTED;
}
template int pd_layers<float>(PdArgs, hipStream_t);
template int pd_layers<bf16>(PdArgs, hipStream_t);

}  // namespace vlg
