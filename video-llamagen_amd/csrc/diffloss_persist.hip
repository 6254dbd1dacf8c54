// Persistent DiffLoss sampler: the whole reverse process of one token (S DDPM steps x SimpleMLPAdaLN) in ONE launch.
//
// Replaces the per-step launch chain of gpt.hip::diffloss_head_fused (8 launches x S steps per token, launch-latency bound) for
// DiffLoss.sample (autoregressive/models/diffloss.py:35-52,217-238) + p_sample_loop (diffusion/gaussian_diffusion.py:376-468).
//
// Every reverse step is a chain of 2 * depth + 1 Linear layers, each an all-to-all over the W hidden units of a row.  Rows are
// independent, so the batch is cut into GROUPS of 4 rows and every group is served by P = W / 32 workgroups, one per 32-column tile of
// the hidden layers (256 workgroups for 32 rows, W 1024: one per CU).  A workgroup keeps its group's full-width residual stream
// h [4][W] in LDS, computes its 32 columns of every Linear layer on the matrix cores (16x16x32 bf16 / 16x16x4 fp32, K dealt to the 4
// waves, weights read as fragments from L2 and requested at the start of the wait for the activations), and exchanges activations with
// the other P - 1 workgroups of its group through global memory - 2 * depth all-gathers per reverse step instead of 8 kernel boundaries.
// Exchange = flag-in-data (the low-latency protocol of collective libraries): every 4 bytes of a tile travel with the phase's epoch in
// one 8-byte write-through (sc1) store, which is single-copy atomic; a consumer reads the P tiles of its group with 16-byte sc1 loads
// (two units each, past the L1) and repeats a load until both tags carry the epoch (bounded spin).  Seeing the tag IS seeing the data:
// no store drain, no separate flag, no ordering between different stores needed.  Buffers are double-buffered by epoch parity: a
// workgroup reaches phase e + 2 only after it has collected phase e + 1, which every producer published only after collecting phase e.
// The buffer is zeroed by a memset node in front of the launch; epochs count within the launch.
// LayerNorm + modulate, the final layer (2C <= 16 outputs), the DDPM update and the next step's input projection (K = C <= 8) are cheap
// and are computed redundantly by every workgroup of a group for its 4 rows - no exchange needed for them.  Constants of the launch
// (input_proj, LayerNorm affine, biases of the workgroup's columns) live in LDS.  Rounding points are those of the launch chain
// (gemm_ln_kernel / EPI_GATED / dl_step_proj_kernel), so both paths agree to fp32 summation order.
// A poll that does not complete within its bound poisons the group's outputs with NaN and leaves (no hang).
// Weights: a workgroup needs the same 32 columns of every matrix in all S steps.  With the depth known at compile time (3, the reference's)
// the block loop is unrolled and the fragments of the first GEMM phases of a step stay in registers for the whole launch, one more phase's
// in LDS; the rest is streamed from L2 per step, requested at the start of the exchange that precedes its GEMM.
// Measured (tools/microbench/dl_persist_lab.hip, MI355X, 32 rows, W 1024, depth 3, bf16): 21.1 us per reverse step against ~65 us of
// the launch chain; of those, 6 x 1.3-2.2 us are the exchanges (store -> fabric -> load), 6 x 0.5 us the GEMMs, 4 x 0.85 us LayerNorms.
#include <type_traits>

#include "gpt_kernels.h"

namespace vlg {

typedef __bf16 dp_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float dp_f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int dp_u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int dp_u32x2_t __attribute__((ext_vector_type(2)));

namespace {

constexpr int DP_R = 4;     // rows per group: 4, or 8 (round 4) when groups of 4 would need more workgroups than the chip has compute units
                            // (W 1024: 33..64 rows, e.g. 32 samples under DiffLoss.sample's own guidance) - template parameter R of the kernel
constexpr int DP_TC = 32;   // hidden columns per workgroup

__device__ __forceinline__ float dp_silu(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ float dp_philox_normal(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
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
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);   // Box-Muller, identical to diffloss.hip
}

template <typename T>
__device__ __forceinline__ void dp_mfma(const dp_u32x4_t& a, const dp_u32x4_t& b, dp_f32x4_t& acc) {
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dp_bf16x8_t, a), __builtin_bit_cast(dp_bf16x8_t, b), acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
  }
}

// wave-wide sum on the DPP network (6 dependent v_add + one readlane; ds_bpermute butterflies cost ~2x the latency of the whole LayerNorm)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dp_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float dp_wave_sum(float v) {
  v += dp_dpp<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dp_dpp<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dp_dpp<0x141, 0xf>(v);   // row_half_mirror
  v += dp_dpp<0x140, 0xf>(v);   // row_mirror: every lane holds the sum of its row of 16
  v += dp_dpp<0x142, 0xa>(v);   // row_bcast15 into rows 1, 3
  v += dp_dpp<0x143, 0xc>(v);   // row_bcast31 into rows 2, 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

#ifndef VLG_DP_R8_LESS
#define VLG_DP_R8_LESS 1   // resident phases given up at 8 rows per group
#endif
#ifndef VLG_DP_NRES2
#define VLG_DP_NRES2 3   // register-resident GEMM phases at NKBW 2 (W 1024 bf16): 4 would spill
#endif
constexpr int DP_MAXKB = 4;   // K blocks (256 bytes of a row) per wave: W * sizeof(T) / 256 / 4 <= 4  (W <= 2048 bf16, 1024 fp32)

template <typename T>
__device__ __forceinline__ void dp_unpack(const dp_u32x4_t& raw, float (&o)[16 / sizeof(T)]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = __uint_as_float((j & 1) ? (raw[j >> 1] & 0xffff0000u) : (raw[j >> 1] << 16));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __uint_as_float(raw[j]);
  }
}
// fp32 -> bf16 on the hardware converter (v_cvt_pk_bf16_f32: round to nearest even, the rounding of DT<bf16>::st for every finite value);
// the single wave that runs a LayerNorm row or an epilogue is bound by its instruction count, and the software rounding is ~8 of them
typedef __bf16 dp_bf2_t __attribute__((ext_vector_type(2)));
typedef float dp_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dp_pack2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(dp_f2_t{a, b}, dp_bf2_t));
}
template <typename T>
__device__ __forceinline__ float dp_rt(float x) {
  if constexpr (sizeof(T) == 2) return __uint_as_float(dp_pack2(x, 0.f) << 16);
  else return x;
}
template <typename T>
__device__ __forceinline__ dp_u32x4_t dp_pack(const float (&v)[16 / sizeof(T)]) {   // rounds like DT<T>::st
  dp_u32x4_t o = dp_u32x4_t{0u, 0u, 0u, 0u};
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = dp_pack2(v[2 * j], v[2 * j + 1]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __float_as_uint(v[j]);
  }
  return o;
}

template <typename T>
struct DpRaw;
template <>
struct DpRaw<bf16> {
  typedef unsigned short type;
  static __device__ __forceinline__ float f(type v) { return __uint_as_float((unsigned)v << 16); }
};
template <>
struct DpRaw<float> {
  typedef float type;
  static __device__ __forceinline__ float f(type v) { return v; }
};

// LDS carve-up (bytes), shared by the kernel and the launcher
template <typename T>
struct DpLds {
  size_t hfull, afull, red, outs, xs, cfs, wip, bip, ln, bias, bfin, wl, total;
  __host__ __device__ DpLds(int W, int depth, int lds_phases = 0, int R = DP_R) {
    size_t o = 0;
    hfull = o; o += (size_t)R * W * sizeof(T);
    afull = o; o += (size_t)R * W * sizeof(T);
    red = o; o += 4 * 2 * 256 * sizeof(float);
    outs = o; o += R * 16 * sizeof(float);
    xs = o; o += R * 16 * sizeof(float);
    cfs = o; o += R * 16 * 8 * sizeof(float);
    wip = o; o += (size_t)W * 8 * sizeof(float);
    bip = o; o += (size_t)W * sizeof(float);
    ln = o; o += (size_t)depth * 2 * W * sizeof(T);
    bias = o; o += (size_t)depth * 2 * DP_TC * sizeof(float);
    bfin = o; o += 16 * sizeof(float);
    o = (o + 15) & ~(size_t)15;
    wl = o; o += (size_t)lds_phases * 32768 * ((W * sizeof(T) + 1023) / 1024);   // fragments of GEMM phases kept in LDS: [2][NKBW][4][256 threads] x 16 B each
    total = o;
  }
};

#ifdef VLG_DP_PROF   // time stamps of reverse step 1 in workgroup 0, kept in LDS until the end (a global store per stamp would sit in front
                     // of the next vmcnt wait and distort what it measures)
#define DP_STAMP(id)                                             \
  do {                                                           \
    if (tid == 0 && k == 1) prof_s[id] = wall_clock64();         \
  } while (0)
#else
#define DP_STAMP(id) \
  do {               \
  } while (0)
#endif

// NKBW: K blocks (256 bytes of a row) per wave = 16-byte chunks of a row per lane = ceil(W * sizeof(T) / 1024)
// FULL: W * sizeof(T) is a multiple of 1024, every wave / lane has exactly NKBW blocks / chunks (no guards: the compiler can count the
// loads in flight and wait for the oldest only)
// DEPTH > 0: the number of res blocks at compile time - the block loop is unrolled and the weight fragments of the first NRES GEMM
// phases of a reverse step (w0[0], w2[0], w0[1], ...) stay in registers for the whole launch (a workgroup uses the same 32 columns of
// every matrix in all S steps): no loads in front of those phases' polls, no wait behind them.  DEPTH = 0: runtime depth, all streamed.
// R: rows per group, 4 or 8.  R = 8: a wave runs the LayerNorm of rows wave and wave + 4 one after the other, rows 4..7 of a GEMM tile sit in
// lanes 16..31 of the accumulators, waves 0 and 1 publish four rows each; three resident phases as at 4 rows (LATE_MOD below frees their registers)
// but no LDS-resident phase (the rows' streams take its LDS).
template <typename T, int NKBW, bool FULL, int DEPTH, int R>
__global__ __launch_bounds__(256) void dl_persist_kernel(DlPersist p) {
  static_assert(R == 4 || R == 8, "rows per group");
  constexpr int RW = R / 4;    // LayerNorm rows per wave = publishing waves
  constexpr int HP = R / 2;    // guidance pairs per group
#ifndef VLG_DP_LATE_MOD
#define VLG_DP_LATE_MOD 1
#endif
  // LATE_MOD (8 rows per group): no second register set for the next step's first modulation rows - they are requested after the final layer's
  // LayerNorm has used the first set, in flight under the final GEMM and the DDPM update - and the 32 registers go to a third resident GEMM
  // phase: 29.65 -> 29.20 us per reverse step at 64 rows (dl_persist_lab; -DVLG_DP_LATE_MOD=0 is the A/B build)
  constexpr bool LATE_MOD = R == 8 && VLG_DP_LATE_MOD != 0;
  // 32 * NKBW VGPRs per resident phase: all 2 * DEPTH phases at NKBW 1, VLG_DP_NRES2 of them at NKBW 2 (W 1024 bf16; 4 would spill)
  constexpr int NRES = (DEPTH == 0 || NKBW > 2) ? 0 : (NKBW == 1 ? (2 * DEPTH < 8 ? 2 * DEPTH : 8) : VLG_DP_NRES2 - (VLG_DP_LATE_MOD ? 0 : VLG_DP_R8_LESS) * (RW - 1));
  // ... and the next NLDS phases keep their fragments in LDS (64 KB per phase at NKBW 2), read back per lane right before the GEMM
  constexpr int NLDS = (R == 4 && NRES > 0 && NRES < 2 * DEPTH && NKBW == 2) ? 1 : 0;
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int KBLK = 256 / (int)sizeof(T);
  const int W = p.W, C = p.C, S = p.S, MR = p.MR, depth = p.depth;
  const int P = W / DP_TC;
  const int nkb = W / KBLK;
  const int nch = W / EPV;                           // 16-byte chunks per activation row
  extern __shared__ __attribute__((aligned(16))) char dp_smem[];
  const DpLds<T> L(W, depth, NLDS, R);
  T* hfull = reinterpret_cast<T*>(dp_smem + L.hfull);       // [R][W] residual stream of the group's rows
  T* afull = reinterpret_cast<T*>(dp_smem + L.afull);       // [R][W] current GEMM input (modulated LN output / mlp.0 output)
  float* red = reinterpret_cast<float*>(dp_smem + L.red);   // [4 waves][2 n-tiles][256]
  float* outs = reinterpret_cast<float*>(dp_smem + L.outs); // [R][16] final layer outputs (eps | v)
  float* xs = reinterpret_cast<float*>(dp_smem + L.xs);     // [R][16] current x_t rows (rounded to T)
  float* cfs = reinterpret_cast<float*>(dp_smem + L.cfs);   // [64][8] this step's DDPM coefficients + noise draw per (row, channel) thread
  float* wip_s = reinterpret_cast<float*>(dp_smem + L.wip); // [8][W] input_proj.weight transposed, zero rows beyond C
  float* bip_s = reinterpret_cast<float*>(dp_smem + L.bip); // [W]
  T* ln_s = reinterpret_cast<T*>(dp_smem + L.ln);           // [depth][weight | bias][W]
  float* bias_s = reinterpret_cast<float*>(dp_smem + L.bias); // [depth][mlp.0 | mlp.2][32] biases of this workgroup's columns
  float* bfin_s = reinterpret_cast<float*>(dp_smem + L.bfin); // [16] final layer bias
  dp_u32x4_t* wl_s = reinterpret_cast<dp_u32x4_t*>(dp_smem + L.wl);   // [phase][2][NKBW][4][256 threads] fragments, each thread its own
  __shared__ int ok_sm;
#ifdef VLG_DP_PROF
  __shared__ unsigned long long prof_s[16];
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  // workgroup -> (group, column tile); tiles with equal (tile * 8 / P) share an XCD under round-robin placement, so an XCD's L2 keeps its
  // eighth of every weight matrix across the S steps (speed only)
  int grp, tile;
  {
    const int b = blockIdx.x;
    grp = b / P;
    const int j = b - grp * P;
    tile = (P % 8 == 0) ? (j % 8) * (P / 8) + j / 8 : j;
  }
  const int col0 = tile * DP_TC;
  // local row j of the group <-> batch row.  Plain: rows R grp .. R grp + R - 1.  Guidance (n_half = B / 2 > 0, DiffLoss.sample's cfg,
  // diffloss.py:37-41): the group serves the HP = R / 2 PAIRS HP grp .. HP grp + HP - 1 - local rows 0 .. HP - 1 are their conditional rows
  // (batch rows pair), local rows HP .. R - 1 their unconditional partners (batch rows pair + n_half), so that a pair's two network
  // outputs meet in one workgroup.
  const int n_half = p.n_half;
  auto row_valid = [&](int j) __attribute__((always_inline)) { return n_half ? (HP * grp + (j % HP)) < n_half : (grp * R + j) < p.B; };
  auto row_batch = [&](int j) __attribute__((always_inline)) {   // batch row of local row j (an invalid local row maps to the group's first row: a valid address)
    if (n_half) {
      const int pair = row_valid(j) ? HP * grp + (j % HP) : HP * grp;
      return pair + (row_valid(j) ? (j / HP) : 0) * n_half;
    }
    return row_valid(j) ? grp * R + j : grp * R;
  };
  // token index of a batch row: the launch's (StepState) or, in a session, the row's own (every slot at its own position)
  const int step_uniform = p.state->step;
  auto step_of = [&](int b) __attribute__((always_inline)) { return p.row_step ? p.row_step[b] : step_uniform; };
  unsigned epoch = 0;
  // exchange buffer: [parity][workgroup][NWD] 8-byte units {4 bytes of the tile, epoch tag}
  constexpr int NWD = R * DP_TC * (int)sizeof(T) / 4;        // data words per published tile: 64 (bf16) / 128 (fp32) at 4 rows
  constexpr int NWDW = NWD / RW;                             // ... per publishing wave
  const __amdgpu_buffer_rsrc_t rs_pay = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, (int)(2 * (size_t)gridDim.x * NWD * 8), 0x00020000);

  // ---- constants of the whole launch -> LDS -------------------------------------------------------------------------------
  {
    const T* wip = reinterpret_cast<const T*>(p.wip);
    const T* bip = reinterpret_cast<const T*>(p.bip);
    for (int e = tid; e < W * 8; e += 256) {   // transposed: the 8 columns a lane projects sit side by side for every input channel
      const int c = e / W, w = e - c * W;
      wip_s[e] = c < C ? DT<T>::ld(wip + (size_t)w * C + c) : 0.f;
    }
    for (int e = tid; e < W; e += 256) bip_s[e] = DT<T>::ld(bip + e);
    for (int blk = 0; blk < depth; ++blk) {
      const T* lw = reinterpret_cast<const T*>(p.ln_w[blk]);
      const T* lb = reinterpret_cast<const T*>(p.ln_b[blk]);
      for (int e = tid; e < W; e += 256) {
        ln_s[(size_t)(blk * 2 + 0) * W + e] = lw[e];
        ln_s[(size_t)(blk * 2 + 1) * W + e] = lb[e];
      }
      if (tid < DP_TC) {
        bias_s[(blk * 2 + 0) * DP_TC + tid] = DT<T>::ld(reinterpret_cast<const T*>(p.b0[blk]) + col0 + tid);
        bias_s[(blk * 2 + 1) * DP_TC + tid] = DT<T>::ld(reinterpret_cast<const T*>(p.b2[blk]) + col0 + tid);
      }
    }
    if (tid < 16) bfin_s[tid] = tid < 2 * C ? DT<T>::ld(reinterpret_cast<const T*>(p.bf) + tid) : 0.f;
    // a fault that is already pending (an earlier token's exchange timed out; the remaining tokens were enqueued long ago): this launch
    // gives up at once instead of spinning each of its exchanges to the bound
    if (tid == 0) ok_sm = (p.fault == nullptr || __hip_atomic_load(p.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) ? 1 : 0;
  }

  // ---- helpers ------------------------------------------------------------------------------------------------------------
  // Loads that leave the chip's caches (this table, the gate rows, the step's noise) are issued right before a collect() and consumed
  // after it: collect() waits for its own loads, which return after every older one, so nothing else in the step waits on memory.
  dp_u32x4_t vsc[RW][NKBW], vsh[RW][NKBW];       // rows of the next LayerNorm (this wave's rows: wave, wave + 4)
  dp_u32x4_t vsc2[RW][NKBW], vsh2[RW][NKBW];     // rows of the next reverse step's first LayerNorm (requested while the final layer's rows are still in use)
  auto prefetch_mod = [&](dp_u32x4_t (&rsc)[RW][NKBW], dp_u32x4_t (&rsh)[RW][NKBW], const T* shift, const T* scale) __attribute__((always_inline)) {
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
      const int grow = row_batch(wave + 4 * rw);
      const dp_u32x4_t* scg = reinterpret_cast<const dp_u32x4_t*>(scale + (size_t)grow * MR);
      const dp_u32x4_t* shg = reinterpret_cast<const dp_u32x4_t*>(shift + (size_t)grow * MR);
#pragma unroll
      for (int it = 0; it < NKBW; ++it) {
        const int c = lane + 64 * it;
        if (FULL || c < nch) {
          rsc[rw][it] = scg[c];
          rsh[rw][it] = shg[c];
        }
      }
    }
  };
  // afull = rt(rt(LN(hfull) [* lnw + lnb]) * (1 + scale) + shift); one wave per row, the row in registers, two-pass statistics like
  // dl_ln_modulate_kernel / gemm_ln_kernel
  // FROMX (the first LayerNorm of a reverse step): the row is not read from hfull but computed here from the step's x_t -
  // h[w] = rt(bias[w] + sum_c x[c] * wip[w][c])  (input_proj, diffloss.py:226; xs and wip_s are zero beyond C) - and stored to hfull
  // for the residual adds: no separate projection pass, no barrier between it and the statistics
  auto ln_modulate = [&](const T* lnw, const T* lnb, auto FROMX) __attribute__((always_inline)) {
    // the wave's rows one after the other (row = wave + 4 rw).  A row costs ~1.5 us of instruction issue (unpack, statistics, modulate, pack:
    // ~700 instructions of one wave), not latency: running the two rows of an 8-row group side by side through the stages measured the same 3.0 us
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
      const int row = wave + 4 * rw;
      float hv[NKBW][EPV];
      float s = 0.f;
      float xr[8];
      if constexpr (decltype(FROMX)::value) {
        const int xrow = n_half ? (row % HP) : row;   // guidance: both rows of a pair are evaluated on the CONDITIONAL row's x_t (diffloss.py:38-39)
        const dp_f32x4_t a = *reinterpret_cast<const dp_f32x4_t*>(xs + xrow * 16), b = *reinterpret_cast<const dp_f32x4_t*>(xs + xrow * 16 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          xr[c] = a[c];
          xr[4 + c] = b[c];
        }
      }
#pragma unroll
      for (int it = 0; it < NKBW; ++it) {
        const int c = lane + 64 * it;
        if (FULL || c < nch) {
          if constexpr (decltype(FROMX)::value) {
            float acc[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {          // channel-major: fmaf chain per element in channel order, as dl_step_proj_kernel
              float wv[EPV];
#pragma unroll
              for (int v4 = 0; v4 < EPV / 4; ++v4) {
                const dp_f32x4_t t4 = *reinterpret_cast<const dp_f32x4_t*>(wip_s + (size_t)cc * W + (size_t)c * EPV + 4 * v4);
#pragma unroll
                for (int u = 0; u < 4; ++u) wv[4 * v4 + u] = t4[u];
              }
#pragma unroll
              for (int j = 0; j < EPV; ++j) acc[j] = fmaf(xr[cc], wv[j], acc[j]);
            }
            float bb[EPV];
#pragma unroll
            for (int v4 = 0; v4 < EPV / 4; ++v4) {
              const dp_f32x4_t t4 = *reinterpret_cast<const dp_f32x4_t*>(bip_s + (size_t)c * EPV + 4 * v4);
#pragma unroll
              for (int u = 0; u < 4; ++u) bb[4 * v4 + u] = t4[u];
            }
#pragma unroll
            for (int j = 0; j < EPV; ++j) hv[it][j] = dp_rt<T>(acc[j] + bb[j]);
            reinterpret_cast<dp_u32x4_t*>(hfull + (size_t)row * W)[c] = dp_pack<T>(hv[it]);
          } else {
            dp_unpack<T>(reinterpret_cast<const dp_u32x4_t*>(hfull + (size_t)row * W)[c], hv[it]);
          }
#pragma unroll
          for (int j = 0; j < EPV; ++j) s += hv[it][j];
        }
      }
      s = dp_wave_sum(s);
      const float mu = s / (float)W;
      float d2 = 0.f;
#pragma unroll
      for (int it = 0; it < NKBW; ++it) {
        const int c = lane + 64 * it;
        if (FULL || c < nch) {
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            const float d = hv[it][j] - mu;
            d2 += d * d;
          }
        }
      }
      d2 = dp_wave_sum(d2);
      const float rstd = 1.0f / sqrtf(d2 / (float)W + 1e-6f);
#pragma unroll
      for (int it = 0; it < NKBW; ++it) {
        const int c = lane + 64 * it;
        if (FULL || c < nch) {
          float sc[EPV], sh[EPV], lw[EPV], lb[EPV], o[EPV];
          dp_unpack<T>(vsc[rw][it], sc);
          dp_unpack<T>(vsh[rw][it], sh);
          if (lnw) {
            dp_unpack<T>(reinterpret_cast<const dp_u32x4_t*>(lnw)[c], lw);
            dp_unpack<T>(reinterpret_cast<const dp_u32x4_t*>(lnb)[c], lb);
          }
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            float n = (hv[it][j] - mu) * rstd;
            if (lnw) n = n * lw[j] + lb[j];
            n = dp_rt<T>(n);
            o[j] = n * (1.0f + sc[j]) + sh[j];
          }
          reinterpret_cast<dp_u32x4_t*>(afull + (size_t)row * W)[c] = dp_pack<T>(o);
        }
      }
    }
  };
  // weight fragments of this workgroup's columns for this wave's K blocks (requested early: they do not depend on the activations)
  auto load_w = [&](const T* wmat, int nrows, int c0, auto& bfx) __attribute__((always_inline)) {
    constexpr int NT = (int)(sizeof(bfx) / sizeof(bfx[0]));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < NKBW; ++i) {
        const int kb = wave + 4 * i;
        if (FULL || kb < nkb) {
          int wrow = c0 + nt * 16 + r;
          wrow = wrow < nrows ? wrow : nrows - 1;      // final layer: 2C <= 16 rows
          const T* wr = wmat + (size_t)wrow * W + (size_t)kb * KBLK;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) bfx[nt][i][s2] = reinterpret_cast<const dp_u32x4_t*>(wr)[s2 * 4 + q];
        }
      }
  };
  // red[wave][nt][256] = partial sums of afull[rows][K] . w[cols][K]^T over this wave's K blocks
  auto gemm = [&](const auto& bfx) __attribute__((always_inline)) {
    constexpr int NT = (int)(sizeof(bfx) / sizeof(bfx[0]));
    dp_f32x4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = dp_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NKBW; ++i) {
      const int kb = wave + 4 * i;
      if (FULL || kb < nkb) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          dp_u32x4_t a = dp_u32x4_t{0u, 0u, 0u, 0u};
          if (r < R) a = *reinterpret_cast<const dp_u32x4_t*>(afull + (size_t)r * W + (size_t)kb * KBLK + (s2 * 4 + q) * EPV);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) dp_mfma<T>(a, bfx[nt][i][s2], acc[nt]);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(wave * 2 + nt) * 256 + e * 64 + lane] = acc[nt][e];
  };
  // element (row, col) of the reduced tile: accumulator layout col = lane & 15, row = 4 * (lane >> 4) + e  ->  rows 0..3 sit in lanes 0..15,
  // rows 4..7 in lanes 16..31
  auto reduced = [&](int nt, int row, int col) __attribute__((always_inline)) {
    const int idx = (row & 3) * 64 + 16 * (row >> 2) + col;   // e = row & 3, lane = 16 (row >> 2) + col
    return red[(0 * 2 + nt) * 256 + idx] + red[(1 * 2 + nt) * 256 + idx] + red[(2 * 2 + nt) * 256 + idx] + red[(3 * 2 + nt) * 256 + idx];
  };
  // Exchange, flag-in-data: every 4 bytes of a tile travel with the epoch in one 8-byte unit (single-copy atomic), so a consumer that
  // sees the tag sees the data - no store drain, no separate flag, no ordering between different stores needed.
  // waves < RW publish, wave w rows 4 w .. 4 w + 3 of the tile: lane l holds elements elem(0), elem(1) of those four rows = one data word
  // (bf16) / words l and l + 64 (fp32)
  auto elem = [&](int j) __attribute__((always_inline)) { return sizeof(T) == 2 ? 2 * lane + j : lane + 64 * j; };
  auto publish = [&](float v0, float v1) __attribute__((always_inline)) {
    const int par = (int)(epoch & 1u);
    const size_t base = ((size_t)par * gridDim.x + blockIdx.x) * NWD + (size_t)wave * NWDW;
    if constexpr (sizeof(T) == 2) {
      const dp_u32x2_t u = dp_u32x2_t{dp_pack2(v0, v1), epoch};
      __builtin_amdgcn_raw_buffer_store_b64(u, rs_pay, (int)((base + lane) * 8), 0, 16);
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(dp_u32x2_t{__float_as_uint(v0), epoch}, rs_pay, (int)((base + lane) * 8), 0, 16);
      __builtin_amdgcn_raw_buffer_store_b64(dp_u32x2_t{__float_as_uint(v1), epoch}, rs_pay, (int)((base + lane + 64) * 8), 0, 16);
    }
  };
  // the workgroup gathers the P tiles of its group into dst [R][W]: 16-byte loads (two units each) past the L1, repeated until both tags
  // carry this epoch (bounded); false = a producer never arrived
  auto collect = [&](T* dst, auto&& pre) __attribute__((always_inline)) -> bool {
    const int par = (int)(epoch & 1u);
    const size_t base = ((size_t)par * gridDim.x + (size_t)grp * P) * NWD;
    const int npair = P * NWD / 2;
    // one batch = four 16-byte loads per thread.  In the first batch the caller's loads for the next phase (weights, table rows) go out
    // in front of the first poll: the poll cannot succeed before the slowest producer's store has crossed the fabric anyway, and
    // the next GEMM needs the weights right after the barrier (measured both orders: this one is 0.7 us per reverse step faster).
    auto batch = [&](int c0, auto with_pre) __attribute__((always_inline)) {
      dp_u32x4_t v[4];
      bool need[4];
      if constexpr (decltype(with_pre)::value) pre();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 256 < npair ? c0 + u * 256 : npair - 1;
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_pay, (int)(base * 8 + (size_t)c * 16), 0, 16);
      }
      for (int spin = 0;; ++spin) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          need[u] = c0 + u * 256 < npair && (v[u][1] != epoch || v[u][3] != epoch);
          any = any || need[u];
        }
        if (!any) break;
        if (spin >= p.spin_max) {
          ok_sm = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (need[u]) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_pay, (int)(base * 8 + (size_t)(c0 + u * 256) * 16), 0, 16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 256;
        if (c < npair) {
          const int slot = c / (NWD / 2), wd = 2 * (c - slot * (NWD / 2));            // slot = workgroup index inside the group
          const int tl = (P % 8 == 0) ? (slot % 8) * (P / 8) + slot / 8 : slot;         // the column tile that workgroup owns
          constexpr int RB = DP_TC * (int)sizeof(T);                                    // bytes per tile row
          const int row = wd * 4 / RB, cb = wd * 4 - row * RB;
          *reinterpret_cast<dp_u32x2_t*>(reinterpret_cast<char*>(dst) + ((size_t)row * W + (size_t)tl * DP_TC) * sizeof(T) + cb) = dp_u32x2_t{v[u][0], v[u][2]};
        }
      }
    };
    batch(tid, std::true_type{});
    for (int c0 = tid + 4 * 256; c0 < npair; c0 += 4 * 256) batch(c0, std::false_type{});
    __syncthreads();
    return ok_sm != 0;
  };

  // ---- x_T and the first input projection ------------------------------------------------------------------------------------
  if (tid < R * 16) {
    const int row = tid / 16, c = tid % 16;
    float v = 0.f;
    if (c < C && row_valid(row)) {
      const int b = n_half ? row_batch(row) % n_half : row_batch(row);   // one x_T draw per pair (diffloss.py:38-39)
      const int step_tok = step_of(b);
      v = p.noise ? p.noise[(((size_t)step_tok * (S + 1)) * p.B_total + p.b_off + b) * C + c]
                  : dp_philox_normal(p.seed, (uint32_t)c, (uint32_t)(p.b_off + b), (uint32_t)step_tok, 0u);
      v = DT<T>::rt(v);
    }
    xs[tid] = v;
  }
  __syncthreads();

  bool alive = ok_sm != 0;   // uniform: read behind the barrier above
  if (!alive) return;
  dp_u32x4_t bf[2][NKBW][4];     // hidden layers: two 16-column tiles (the streamed phases)
  dp_u32x4_t bff[1][NKBW][4];    // final layer: one tile of 2C <= 16 outputs, resident
  dp_u32x4_t wres[NRES > 0 ? NRES : 1][2][NKBW][4];
  load_w(reinterpret_cast<const T*>(p.wf), 2 * C, 0, bff);
  // fragments of an LDS-resident phase <-> bf
  auto lds_put = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < NKBW; ++i)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) wl_s[((((size_t)slot * 2 + nt) * NKBW + i) * 4 + s2) * 256 + tid] = bf[nt][i][s2];
  };
  auto lds_get = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < NKBW; ++i)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) bf[nt][i][s2] = wl_s[((((size_t)slot * 2 + nt) * NKBW + i) * 4 + s2) * 256 + tid];
  };
  if constexpr (NRES > 0) {
#pragma unroll
    for (int ph = 0; ph < NRES; ++ph) load_w(reinterpret_cast<const T*>((ph & 1) ? p.w2[ph >> 1] : p.w0[ph >> 1]), W, col0, wres[ph]);
#pragma unroll
    for (int j = 0; j < NLDS; ++j) {
      const int ph = NRES + j;
      load_w(reinterpret_cast<const T*>((ph & 1) ? p.w2[ph >> 1] : p.w0[ph >> 1]), W, col0, bf);
      lds_put(j);
    }
  } else {
    load_w(reinterpret_cast<const T*>(p.w0[0]), W, col0, bf);
  }
  // publishing waves: this lane's two elements of the published [R][32] tile
  const int e0 = elem(0), e1 = elem(1);
  const int pwv = wave < RW ? wave : 0;
  const int prow0 = 4 * pwv + e0 / DP_TC, pc0 = e0 % DP_TC, prow1 = 4 * pwv + e1 / DP_TC, pc1 = e1 % DP_TC;
  const int pg0 = row_batch(prow0), pg1 = row_batch(prow1);
  {
    const T* mod = reinterpret_cast<const T*>(p.mod_all) + (size_t)(S - 1) * p.B * MR;
    prefetch_mod(vsc, vsh, mod, mod + W);
  }
  // one res block (diffloss.py:99-129).  WA / WB: register-resident fragments of mlp.0 / mlp.2; SA / SB / SN: where mlp.0 / mlp.2 of this
  // block / mlp.0 of the next block live - 0 registers, 1 streamed into `bf` (requested at the start of the preceding exchange), 2 LDS
  auto res_block = [&](int blk, int k, int i, const T* mod, const auto& WA, const auto& WB, auto SA, auto SB, auto SN, DdpmCoef& cf, float& nz) __attribute__((always_inline)) {
    const T* m0 = mod + (size_t)blk * 3 * W;          // [shift | scale | gate]  (diffloss.py:125)
    if (blk == 1) DP_STAMP(12);
    if (blk == 0) ln_modulate(ln_s, ln_s + W, std::true_type{});
    else ln_modulate(ln_s + (size_t)(blk * 2) * W, ln_s + (size_t)(blk * 2 + 1) * W, std::false_type{});
    if (blk == 1) DP_STAMP(13);
    __syncthreads();
    if (blk == 0) DP_STAMP(1);
    if (blk == 1) DP_STAMP(14);
    if constexpr (decltype(SA)::value == 2) lds_get(2 * blk - NRES);
    if constexpr (decltype(SA)::value != 0) gemm(bf);
    else gemm(WA);
    __syncthreads();
    if (blk == 0) DP_STAMP(2);
    epoch += 1;
    if (wave < RW) {          // mlp.0: rt(silu(rt(acc + bias)))
      const float v0 = dp_rt<T>(reduced(pc0 >> 4, prow0, pc0 & 15) + bias_s[(blk * 2 + 0) * DP_TC + pc0]);
      const float v1 = dp_rt<T>(reduced(pc1 >> 4, prow1, pc1 & 15) + bias_s[(blk * 2 + 0) * DP_TC + pc1]);
      publish(dp_silu(v0), dp_silu(v1));
    }
    if (blk == 0) DP_STAMP(3);
    typename DpRaw<T>::type g0r = 0, g1r = 0;      // gate values as stored (converted where they are used, not where they are requested)
    alive = collect(afull, [&]() __attribute__((always_inline)) {
      if constexpr (decltype(SB)::value == 1) load_w(reinterpret_cast<const T*>(p.w2[blk]), W, col0, bf);   // in flight during the exchange
      if (wave < RW) {
        g0r = *reinterpret_cast<const typename DpRaw<T>::type*>(m0 + (size_t)pg0 * MR + 2 * W + col0 + pc0);
        g1r = *reinterpret_cast<const typename DpRaw<T>::type*>(m0 + (size_t)pg1 * MR + 2 * W + col0 + pc1);
        if (blk == 0) {
          const int row = 4 * wave + lane / 16, c = lane % 16;
          if (c < C && row_valid(row)) {
            cf = p.coef[i];
            const int b = row_batch(row);
            const int step_tok = step_of(b);
            nz = p.noise ? p.noise[(((size_t)step_tok * (S + 1) + 1 + k) * p.B_total + p.b_off + b) * C + c]
                         : dp_philox_normal(p.seed, (uint32_t)c, (uint32_t)(p.b_off + b), (uint32_t)step_tok, (uint32_t)(1 + k));
          }
        }
      }
    });
    if (blk == 0) DP_STAMP(4);
    if (!alive) return;
    if (blk == 0 && wave < RW) {   // the step's coefficients and noise are in: park them in LDS, so that the DDPM update at the end of
                                   // the step does not wait on whatever loads are in flight by then
      float* o = cfs + (wave * 64 + lane) * 8;
      o[0] = cf.sqrt_recip; o[1] = cf.sqrt_recipm1; o[2] = cf.coef1; o[3] = cf.coef2; o[4] = cf.min_log; o[5] = cf.max_log;
      o[6] = __int_as_float(cf.nonzero); o[7] = nz;
    }
    if constexpr (decltype(SB)::value == 2) lds_get(2 * blk + 1 - NRES);
    if constexpr (decltype(SB)::value != 0) gemm(bf);
    else gemm(WB);
    __syncthreads();
    if (blk == 0) DP_STAMP(5);
    epoch += 1;
    if (wave < RW) {          // mlp.2 + gate + residual: h = rt(h + rt(gate * rt(acc + bias)))   (diffloss.py:128)
      const float v0 = reduced(pc0 >> 4, prow0, pc0 & 15) + bias_s[(blk * 2 + 1) * DP_TC + pc0];
      const float v1 = reduced(pc1 >> 4, prow1, pc1 & 15) + bias_s[(blk * 2 + 1) * DP_TC + pc1];
      const float h0 = DT<T>::ld(hfull + (size_t)prow0 * W + col0 + pc0), h1 = DT<T>::ld(hfull + (size_t)prow1 * W + col0 + pc1);
      const float g0 = DpRaw<T>::f(g0r), g1 = DpRaw<T>::f(g1r);
      publish(h0 + dp_rt<T>(g0 * dp_rt<T>(v0)), h1 + dp_rt<T>(g1 * dp_rt<T>(v1)));
    }
    if (blk == 0) DP_STAMP(6);
    alive = collect(hfull, [&]() __attribute__((always_inline)) {
      prefetch_mod(vsc, vsh, m0 + 3 * W, m0 + 4 * W);        // the next block's, or the final layer's, [shift | scale]
      if (blk + 1 < depth) {
        if constexpr (decltype(SN)::value == 1) load_w(reinterpret_cast<const T*>(p.w0[blk + 1]), W, col0, bf);
      } else if (k + 1 < S) {
        if constexpr (!LATE_MOD) {
          const T* mn = reinterpret_cast<const T*>(p.mod_all) + (size_t)(i - 1) * p.B * MR;
          prefetch_mod(vsc2, vsh2, mn, mn + W);
        }
      }
    });
    if (blk == 0) DP_STAMP(7);
    if (blk == 1) DP_STAMP(15);
  };
  for (int k = 0; k < S && alive; ++k) {
    const int i = S - 1 - k;
    const T* mod = reinterpret_cast<const T*>(p.mod_all) + (size_t)i * p.B * MR;
    DP_STAMP(0);
    DdpmCoef cf{};     // this step's posterior coefficients and noise draw (threads of the DDPM update)
    float nz = 0.f;
    if constexpr (DEPTH == 0) {
      for (int blk = 0; blk < depth && alive; ++blk) res_block(blk, k, i, mod, bf, bf, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, cf, nz);
    } else {
      // unrolled: phase 2 * blk (mlp.0) and 2 * blk + 1 (mlp.2) are resident while < NRES
#define DP_KIND(PH_) ((PH_) < NRES ? 0 : ((PH_) < NRES + NLDS ? 2 : 1))
#define DP_BLOCK(B_)                                                                                                                    \
  if constexpr ((B_) < DEPTH) {                                                                                                         \
    if (alive)                                                                                                                          \
      res_block((B_), k, i, mod, wres[(2 * (B_) < NRES) ? 2 * (B_) : 0], wres[(2 * (B_) + 1 < NRES) ? 2 * (B_) + 1 : 0],                \
                std::integral_constant<int, DP_KIND(2 * (B_))>{}, std::integral_constant<int, DP_KIND(2 * (B_) + 1)>{},                 \
                std::integral_constant<int, DP_KIND(2 * (B_) + 2)>{}, cf, nz);                                                          \
  }
      static_assert(DEPTH <= 4, "unrolled for up to 4 res blocks");
      DP_BLOCK(0)
      DP_BLOCK(1)
      DP_BLOCK(2)
      DP_BLOCK(3)
#undef DP_BLOCK
#undef DP_KIND
    }
    if (!alive) break;
    DP_STAMP(8);
    // final layer (diffloss.py:141-148): modulate(LN(h)) -> Linear W -> 2C, computed by every workgroup for its rows
    {
      ln_modulate(nullptr, nullptr, std::false_type{});
      if constexpr (LATE_MOD) {   // requested here, behind the final layer's use of the same registers: in flight under the final GEMM and the DDPM update
        if (k + 1 < S) {
          const T* mn = reinterpret_cast<const T*>(p.mod_all) + (size_t)(i - 1) * p.B * MR;
          prefetch_mod(vsc, vsh, mn, mn + W);
        }
      } else {
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int it = 0; it < NKBW; ++it) {
            vsc[rw][it] = vsc2[rw][it];
            vsh[rw][it] = vsh2[rw][it];
          }
      }
      __syncthreads();
      gemm(bff);
      __syncthreads();
      DP_STAMP(9);
      // streamed form: the next step's first weights, consumed after that step's LayerNorm (measured: requested before the last collect()
      // instead, they sit in front of its polls and cost more than they hide)
      if constexpr (NRES == 0)
        if (k + 1 < S) load_w(reinterpret_cast<const T*>(p.w0[0]), W, col0, bf);
      // p_sample (gaussian_diffusion.py:254-332,376-420): learned-range variance, eps prediction, clip_denoised = False
      if (tid < R * 16) {
        const int row = tid / 16, c = tid % 16;
        if (c < C && row_valid(row)) {
          const float* cs = cfs + tid * 8;       // {sqrt_recip, sqrt_recipm1, coef1, coef2, min_log, max_log, nonzero, noise} of this step
          float eps = dp_rt<T>(reduced(0, row, c) + bfin_s[c]);
          const float v = dp_rt<T>(reduced(0, row, C + c) + bfin_s[C + c]);   // final layer outputs (eps | v)
          if (n_half) {   // forward_with_cfg (diffloss.py:240-248): eps = u + cfg (c - u) from the pair's two rows; variance, draw and x_t stay per row
            const float ce = dp_rt<T>(reduced(0, row % HP, c) + bfin_s[c]), ue = dp_rt<T>(reduced(0, HP + row % HP, c) + bfin_s[c]);
            eps = dp_rt<T>(ue + dp_rt<T>(p.cfg * dp_rt<T>(ce - ue)));
          }
          const float xv = xs[row * 16 + c];
          const float frac = (v + 1.0f) / 2.0f;
          const float logvar = frac * cs[5] + (1.0f - frac) * cs[4];
          const float x0 = cs[0] * xv - cs[1] * eps;
          const float mean = cs[2] * x0 + cs[3] * xv;
          float rr = mean;
          if (__float_as_int(cs[6]) != 0) rr = mean + expf(0.5f * logvar) * cs[7] * p.temperature;
          xs[row * 16 + c] = dp_rt<T>(rr);
        }
      }
      __syncthreads();
      DP_STAMP(10);
      DP_STAMP(11);
    }
  }
#ifdef VLG_DP_PROF
  if (p.prof && blockIdx.x == 0 && tid < 16) p.prof[tid] = prof_s[tid];
#endif
  // an exchange wait ran out: tell the host (vlg_gpt_status); system scope - the word lives in pinned host memory
  if (!alive && tid == 0 && p.fault) __hip_atomic_store(p.fault, kFaultDlPersist | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // ---- results: the workgroup of column tile 0 writes its group's rows (NaN when an exchange timed out) -------------------------
  if (tile == 0 && tid < R * 16) {
    const int row = tid / 16, c = tid % 16;
    if (c < C && row_valid(row)) {
      const int b = row_batch(row);
      const int step_tok = step_of(b);
      const float v = alive ? xs[row * 16 + c] : __int_as_float(0x7fc00000);
      p.cur[(size_t)b * C + c] = v;
      p.out_lat[((size_t)b * p.N + step_tok) * C + c] = v;
      if (p.trace) p.trace[((size_t)step_tok * p.B_total + p.b_off + b) * C + c] = v;
    }
  }
}

}  // namespace

namespace {
// compute units of the current device: every workgroup of the launch must be resident at once (they wait for each other), and each takes a
// whole CU (LDS >= 96 KB)
int dp_cu_count() {   // per device: a process may drive several
  static int cache[64] = {};
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev < 64 && cache[dev] > 0) return cache[dev];
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (dev < 64) cache[dev] = cus;
  return cus;
}
// rows per group of a launch: 4 while that fits one workgroup per CU, else 8; `want` (option dl_persist = 4 / 8) forces one of them; 0 = no fit
int dp_group_rows(int B, int W, int want) {
  if (B < 1 || W < DP_TC) return 0;
  const int P = W / DP_TC, cus = dp_cu_count();
  for (int R = 4; R <= 8; R += 4) {
    if (want == 4 || want == 8) {
      if (R != want) continue;
    }
    if (cdiv(B, R) * P <= cus) return R;
  }
  return 0;
}
// the kernel's NLDS for (R, depth, K blocks per wave)
template <typename T>
int dp_lds_phases(int R, int W, int depth) {
  const int nkbw = cdiv(W * (int)sizeof(T) / 256, 4);
  return (R == 4 && depth == 3 && nkbw == 2 && VLG_DP_NRES2 > 0 && VLG_DP_NRES2 < 6) ? 1 : 0;
}
}  // namespace

// exchange buffer: [parity][workgroup][R * 32 elements as 8-byte units]; sized for either group height
size_t dl_persist_xbuf_bytes(int B, int W, int esz) {
  size_t best = 0;
  for (int R = 4; R <= 8; R += 4) {
    const size_t n = (size_t)2 * cdiv(B, R) * (W / DP_TC) * (R * DP_TC * esz / 4) * 8;
    best = n > best ? n : best;
  }
  return best;
}

template <typename T>
bool dl_persist_ok(int B, int W, int C, int depth, int rows) {
  if (B < 1 || W < 256 || W % 256 != 0 || C < 1 || 2 * C > 16 || depth < 1 || depth > 8) return false;
  const int nkb = W * (int)sizeof(T) / 256;
  const int R = dp_group_rows(B, W, rows);   // one workgroup per CU: every participant of an exchange is resident
  if (R == 8 && cdiv(nkb, 4) > 2 && depth != 3) return false;   // not instantiated (register budget)
  return R > 0 && nkb <= 4 * DP_MAXKB && W / DP_TC <= 64 && DpLds<T>(W, depth, dp_lds_phases<T>(R, W, depth), R).total <= 150 * 1024;
}
template bool dl_persist_ok<float>(int, int, int, int, int);
template bool dl_persist_ok<bf16>(int, int, int, int, int);

namespace {
template <typename T, int NKBW, bool FULL, int DEPTH, int R>
int dp_launch2(const DlPersist& p, int grid, size_t lds, hipStream_t st) {
  static bool attr[64] = {};   // the attribute is per device (a code object is loaded per device)
  int dev = 0;
  VLG_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr[dev]) {
    VLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dl_persist_kernel<T, NKBW, FULL, DEPTH, R>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    if (dev >= 0 && dev < 64) attr[dev] = true;
  }
  dl_persist_kernel<T, NKBW, FULL, DEPTH, R><<<grid, 256, lds, st>>>(p);
  return VLG_OK;
}
template <typename T, int NKBW, bool FULL, int DEPTH>
int dp_launch1(const DlPersist& p, int R, int grid, size_t lds, hipStream_t st) {
  if constexpr (NKBW <= 2 || DEPTH == 3) {   // (8 rows x 4 K blocks per wave x runtime depth would spill 340 VGPRs: not built, dl_persist_ok says no)
    if (R == 8) return dp_launch2<T, NKBW, FULL, DEPTH, 8>(p, grid, lds, st);
  }
  return dp_launch2<T, NKBW, FULL, DEPTH, 4>(p, grid, lds, st);
}
template <typename T, int NKBW, bool FULL>
int dp_launch(const DlPersist& p, int R, int grid, size_t lds, hipStream_t st) {
  if (p.depth == 3) return dp_launch1<T, NKBW, FULL, 3>(p, R, grid, lds, st);   // the reference's depth (gpt_video_diff.py:77): resident weights
  return dp_launch1<T, NKBW, FULL, 0>(p, R, grid, lds, st);
}
}  // namespace

template <typename T>
int dl_persist(const DlPersist& p, hipStream_t st) {
  if (!dl_persist_ok<T>(p.B, p.W, p.C, p.depth, p.rows)) {
    set_error("dl_persist: shape B=%d W=%d C=%d depth=%d (rows per group %d) not covered", p.B, p.W, p.C, p.depth, p.rows);
    return VLG_ERR_UNSUPPORTED;
  }
  const int R = dp_group_rows(p.B, p.W, p.rows);
  if (p.n_half && p.B != 2 * p.n_half) {
    set_error("dl_persist: guidance pairs rows (b, b + B/2): n_half %d does not match %d rows", p.n_half, p.B);
    return VLG_ERR_BAD_SHAPE;
  }
  const int groups = cdiv(p.B, R), P = p.W / DP_TC;
  VLG_HIP(hipMemsetAsync(p.xbuf, 0, dl_persist_xbuf_bytes(p.B, p.W, (int)sizeof(T)), st));   // epoch tags count within the launch
  // LDS: the activations plus padding up to > 80 KB so that no two workgroups share a CU (table row 1 of the hand-off forms is measured
  // for one workgroup per CU; correctness does not depend on it, the exchange latency does)
  size_t lds = DpLds<T>(p.W, p.depth, dp_lds_phases<T>(R, p.W, p.depth), R).total;   // = the carve-up of the kernel that will run
  if (lds < 96 * 1024) lds = 96 * 1024;
  const int nkbw = cdiv(p.W * (int)sizeof(T) / 256, 4);
  const bool full = (p.W * (int)sizeof(T)) % 1024 == 0 && nkbw != 3;
  int rc;
  if (nkbw <= 1) rc = full ? dp_launch<T, 1, true>(p, R, groups * P, lds, st) : dp_launch<T, 1, false>(p, R, groups * P, lds, st);
  else if (nkbw == 2) rc = full ? dp_launch<T, 2, true>(p, R, groups * P, lds, st) : dp_launch<T, 2, false>(p, R, groups * P, lds, st);
  else rc = full ? dp_launch<T, 4, true>(p, R, groups * P, lds, st) : dp_launch<T, 4, false>(p, R, groups * P, lds, st);
  if (rc != VLG_OK) return rc;
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}
template int dl_persist<float>(const DlPersist&, hipStream_t);
template int dl_persist<bf16>(const DlPersist&, hipStream_t);

}  // namespace vlg
