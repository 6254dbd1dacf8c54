// Conv / GroupNorm / spatial-attention kernels for the VQ-16 and CausalVideoVAE decoders (gfx950).
//
// Reference behaviour: tokenizer/tokenizer_image/vq_model.py:279-378 (ResnetBlock, AttnBlock, Normalize, Upsample),
// CausalVideoVAE/causalvideovae/model/modules/conv.py:76-130 (CausalConv3d), resnet_block.py:140-172,
// attention.py:40-76 (AttnBlock3D, Q12 reshape), updownsample.py:124-153,182-194, normalize.py:14-17.
//
// Convolution = implicit GEMM on MFMA (v_mfma_f32_32x32x16_bf16): M = output positions (channels-last, so one
// position's Cin slice is contiguous), N = Cout, K = taps x Cin.  The causal time pad (replicate frame 0), the zero
// H/W pad and the nearest-2x upsample are index arithmetic in the A-tile gather - no padded or upsampled tensor is ever
// materialised.  128 x BN x 32 tiles, LDS double buffer with register-staged prefetch (global loads of step k+1 are in
// flight while step k computes), 16-byte-chunk XOR swizzle so ds_read_b128 fragment reads are conflict-free.
#include <type_traits>
#include "conv_kernels.h"

namespace vlg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// x * sigmoid(x); the reciprocal is the hardware's (1 ulp) instead of an IEEE division (10 instructions per element of every GroupNorm + swish
// pass): far inside the rounding of the T-typed store that follows
__device__ __forceinline__ float swish_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------------------------
// implicit-GEMM convolution, bf16
// ---------------------------------------------------------------------------------------------------------------
struct PosDec {
  int b, t, y, x;
  bool ok;
};

__device__ __forceinline__ PosDec decode_pos(long long p, const ConvDesc& d, long long ptot) {
  PosDec r;
  r.ok = p < ptot;
  if (!r.ok) p = 0;
  r.x = (int)(p % d.Wo);
  p /= d.Wo;
  r.y = (int)(p % d.Ho);
  p /= d.Ho;
  r.t = (int)(p % d.To);
  r.b = (int)(p / d.To);
  return r;
}

// element offset of the input row feeding output position `pd` through tap (a,i,j), or -1 (zero padding)
__device__ __forceinline__ long long tap_src(const PosDec& pd, const ConvDesc& d, int a, int i, int j) {
  int ti;
  if (d.tmode == 0) {
    ti = pd.t + a - (d.kt - 1);
    ti = ti < 0 ? 0 : ti;                                 // causal: frame 0 replicated in front (conv.py:126-129)
  } else {
    ti = pd.t + a - ((d.kt - 1) / 2 + (d.kt - 1) % 2);    // SamePadConv3d: zero pad (p//2 + p%2, p//2)
    if (ti < 0 || ti >= d.Ti) return -1;
  }
  const int ph0 = d.ph0 < 0 ? d.kh / 2 : d.ph0, pw0 = d.pw0 < 0 ? d.kw / 2 : d.pw0;
  const int uy = pd.y * d.sh + i - ph0, ux = pd.x * d.sh + j - pw0;
  const int He = d.Hi << d.up, We = d.Wi << d.up;                           // extent of the (virtually upsampled) input
  if (!pd.ok || uy < 0 || ux < 0 || uy >= He || ux >= We) return -1;        // zero padding
  const int iy = uy >> d.up, ix = ux >> d.up;             // nearest 2x upsample folded in
  return ((((long long)pd.b * d.Ti + ti) * d.Hi + iy) * d.Wi + ix) * d.Cin;
}

// T = bf16: 32-channel K steps on v_mfma_f32_32x32x16_bf16.  T = float (round 2; the reference runs the VQ-16 decoder in fp32): 16-channel
// K steps - the same 64-byte LDS rows - on the exact-fp32 v_mfma_f32_32x32x2_f32, so the 1x1 convolutions of the attention blocks, the
// shortcut convolutions and conv_out (Cout 3) of an fp32 handle no longer run the direct-convolution kernel (69 % of an fp32 decode).
// SMALLC: Cin below one K step (post_quant_conv: 8 -> 4, CausalVAE conv_in: 4 -> 512): one zero-padded K step per tap, operands gathered
// element by element - tiny layers, but they no longer fall to the direct-convolution kernel.
template <typename T, int BN, bool SMALLC = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvDesc d, const T* __restrict__ in, const T* __restrict__ w,
                                                        const float* __restrict__ bias, const T* __restrict__ residual,
                                                        T* __restrict__ out_cl, float* __restrict__ out_planar) {
  constexpr int EPV = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int KC = 4 * EPV;                // channels per K step
  constexpr int BM = 128;
  constexpr int WM = (BN == 128) ? 64 : 32, WN = (BN == 128) ? 64 : 32;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int BCH = BN * 4 / 256;  // B chunks per thread (2 for BN=128, 0.5 -> handled by predicate for BN=32)
  __shared__ uint4 As[2][BM * 4];
  __shared__ uint4 Bs[2][BN * 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = (BN == 128) ? (wave >> 1) : wave, wave_n = (BN == 128) ? (wave & 1) : 0;
  const long long ptot = (long long)d.B * d.To * d.Ho * d.Wo;
  const long long pos0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int taps = d.kt * d.kh * d.kw;
  const int ncc = SMALLC ? 1 : d.Cin / KC;
  const int nk = taps * ncc;

  // loader roles
  PosDec pd[2];
  int arow[2], ach[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    arow[i] = c >> 2;
    ach[i] = c & 3;
    pd[i] = decode_pos(pos0 + arow[i], d, ptot);
  }
  constexpr int NBL = (BCH >= 1) ? BCH : 1;
  int brow[NBL], bch[NBL];
  bool bok[NBL];
#pragma unroll
  for (int i = 0; i < NBL; ++i) {
    const int c = tid + 256 * i;
    brow[i] = c >> 2;
    bch[i] = c & 3;
    bok[i] = (c < BN * 4);
  }

  uint4 ra[2], rb[NBL];
  auto gload = [&](int ks) {
    const int tap = ks / ncc, cc = ks - tap * ncc;
    const int a = tap / (d.kh * d.kw), rem = tap - a * (d.kh * d.kw);
    const int ii = rem / d.kw, jj = rem - ii * d.kw;
    if constexpr (SMALLC) {
      auto gather = [&](const T* base, bool ok, int chunk) {
        alignas(16) T tmp[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const int c = chunk * EPV + e;
          DT<T>::st(&tmp[e], (ok && c < d.Cin) ? DT<T>::ld(base + c) : 0.f);
        }
        return *reinterpret_cast<const uint4*>(tmp);
      };
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const long long src = tap_src(pd[i], d, a, ii, jj);
        ra[i] = gather(in + (src >= 0 ? src : 0), src >= 0, ach[i]);
      }
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
        const int co = n0 + brow[i];
        const bool ok = bok[i] && co < d.Cout;
        rb[i] = gather(w + ((size_t)(ok ? co : 0) * taps + tap) * d.Cin, ok, bch[i]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long src = tap_src(pd[i], d, a, ii, jj);
      ra[i] = make_uint4(0, 0, 0, 0);
      if (src >= 0) ra[i] = *reinterpret_cast<const uint4*>(in + src + cc * KC + ach[i] * EPV);
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      rb[i] = make_uint4(0, 0, 0, 0);
      const int co = n0 + brow[i];
      if (bok[i] && co < d.Cout) rb[i] = *reinterpret_cast<const uint4*>(w + ((size_t)co * taps + tap) * d.Cin + cc * KC + bch[i] * EPV);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) As[buf][arow[i] * 4 + (ach[i] ^ ((arow[i] >> 2) & 3))] = ra[i];
#pragma unroll
    for (int i = 0; i < NBL; ++i)
      if (bok[i]) Bs[buf][brow[i] * 4 + (bch[i] ^ ((brow[i] >> 2) & 3))] = rb[i];
  };

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  gload(0);
  lstore(0);
  __syncthreads();
  const int r32 = lane & 31, hh = lane >> 5;
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) gload(ks + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[MI], bfr[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wave_m * WM + mi * 32 + r32;
        af[mi] = As[buf][row * 4 + ((2 * kk + hh) ^ ((row >> 2) & 3))];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = wave_n * WN + ni * 32 + r32;
        bfr[ni] = Bs[buf][row * 4 + ((2 * kk + hh) ^ ((row >> 2) & 3))];
      }
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[mi]), __builtin_bit_cast(bf16x8_t, bfr[ni]),
                                                                  acc[mi][ni], 0, 0, 0);
      } else {
        // K = 2 per MFMA: element e of the hh = 0 lanes' chunk pairs with element e of the hh = 1 lanes' chunk (as in conv_halo_kernel<float>)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float((&af[mi].x)[e]), __uint_as_float((&bfr[ni].x)[e]), acc[mi][ni],
                                                                 0, 0, 0);
      }
    }
    if (ks + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: + bias (+ residual) -> channels-last bf16 or planar fp32
  const long long pper = (long long)d.To * d.Ho * d.Wo;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int co = n0 + wave_n * WN + ni * 32 + r32;
      const float bv = (co < d.Cout && bias) ? bias[co] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int prow = wave_m * WM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        const long long p = pos0 + prow;
        if (p < ptot && co < d.Cout) {
          float v = acc[mi][ni][e] + bv;
          if (residual) v += DT<T>::ld(residual + p * d.Cout + co);
          if (out_cl)
            DT<T>::st(out_cl + p * d.Cout + co, v);
          else
            out_planar[((p / pper) * d.Cout + co) * pper + (p % pper)] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// halo-tile convolution for the 3x3 (x kt) stride-1 causal convs with Cout % 128 == 0 - every heavy layer of both decoders.
//
// conv_mfma_kernel re-fetches each input element once per tap (27x for 3x3x3) through L2 -> LDS: 64 FLOP per staged byte, and
// the ablation shows the staging path, not the MFMAs, bounds it.  Here a workgroup owns a 2 x 4 x 32 block of
// output positions (256 rows of the GEMM) x 128 output channels, stages the input patch it needs - (2 + kt - 1) x 6 x 34
// positions x 32 channels, <= 51 KB - into LDS ONCE per 32-channel chunk and walks the taps over it: the A fragment of tap
// (a,i,j) is the same LDS image read at a row offset.  Per chunk: 45 KB of input + taps x 8 KB of weights for taps x 2.1 MFLOP
// (212 FLOP per staged byte at 27 taps).  Weights: one kernel row (3 taps) per step, register prefetch two steps ahead, LDS
// double buffer: 24 MFMAs per wave between barriers.  Padding (zero in H/W, replicate-first-frame in T) and the nearest-2x upsample are
// resolved when the patch is gathered (LDS-DMA, a share per step).  8 MFMA waves (4 along positions x 2 along channels), each a 64 x 64
// sub-tile of 32x32x16 MFMAs; bf16: + 4 loader waves (template parameter WS below).  The epilogue leaves through LDS as whole rows.
// ---------------------------------------------------------------------------------------------------------------
// Tile width 32: the 32 rows of one MFMA block are 32 consecutive patch rows (one output row of the tile), which is what makes the
// ds_read_b128 A-fragment reads conflict-free under the (row >> 2) & 3 chunk swizzle for every tap offset - the instruction's four
// lane groups are {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md §LDS); with a 16-wide tile lanes 16-31 sit one
// patch pitch (18 rows) away and half of the tap offsets are 2-way conflicts.
constexpr int HT_TW = 32, HT_HW = HT_TW + 2;
constexpr int HT_MAXROWS = (2 + 2) * (4 + 2) * HT_HW;          // 816 patch positions at kt = 3, tile 2 x 4 x 32 (1 x 10 x 34 = 340 for images)
template <int NTHR>
constexpr int ht_slots() { return (HT_MAXROWS * 4 + NTHR - 1) / NTHR; }   // 16-byte chunks per thread per patch: 7 at 512 threads, 13 at 256
constexpr size_t HT_LDS_BYTES = (size_t)(2 * HT_MAXROWS * 4 + 2 * 3 * 128 * 4) * sizeof(uint4);   // 2 patches + 2 x 3 weight taps

// T = bf16: 32-channel chunks, v_mfma_f32_32x32x16_bf16.  T = float (the reference runs the VQ-16 decoder in fp32): 16-channel
// chunks - the same 64-byte patch rows and LDS image - on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak instead of a direct conv).
// NWM: 32-row MFMA blocks per wave along the positions: 8 waves (4 x 2), each 64 x 64.  (Round 2 measured 16 waves of 32 x 64, 4 waves of
// 128 x 64 and taps 1 / 2 as DPP lane shifts of tap 0's fragment: all slower; removed in round 3.)
__device__ __attribute__((aligned(64))) const unsigned conv_zero_chunk[16] = {};   // the source of padding rows in conv_halo_kernel's LDS-DMA gather (64 B)

#ifdef VLG_CONV_LAB
// lab build only (tools/conv_lab.py): in-kernel phase stamps of every 61st workgroup - records of 24 x u64:
// {Cin, Wo, kt, Q (steps), t_start, t_roles, t_prologue, t_loop, t_end, shader cycles at loop start, at loop end, -, stamps after
// steps 0..11}; wall_clock64 = 100 MHz
__device__ unsigned long long* conv_lab_buf = nullptr;
__device__ unsigned conv_lab_count = 0;
__device__ int conv_lab_mode = 0;   // 1: every workgroup of the Cin 128 / 256-wide / kt 3 layers (gaps between workgroups on a CU)
extern "C" int vlg_conv_lab_mode(int mode) { return hipMemcpyToSymbol(HIP_SYMBOL(conv_lab_mode), &mode, sizeof(mode)) != hipSuccess; }
extern "C" int vlg_conv_lab_set(void* d_buf) {
  unsigned zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(conv_lab_buf), &d_buf, sizeof(d_buf)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(conv_lab_count), &zero, sizeof(zero)) != hipSuccess;
}
extern "C" int vlg_conv_lab_count(unsigned* n) { return hipMemcpyFromSymbol(n, HIP_SYMBOL(conv_lab_count), sizeof(*n)) != hipSuccess; }
#define LAB_STAMP(i)                                    \
  do {                                                  \
    if (lab_rec) lab_rec[i] = wall_clock64();           \
  } while (0)
#define LAB_CYCLES(i)                                        \
  do {                                                       \
    if (lab_rec) lab_rec[i] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#else
#define LAB_CYCLES(i) \
  do {                \
  } while (0)
#define LAB_STAMP(i) \
  do {               \
  } while (0)
#endif
// timing ablations of the lab build (wrong results): -DVLG_CONV_LAB_NOW no weight loads in the loop, -DVLG_CONV_LAB_NOMFMA no fragment
// reads / MFMAs, -DVLG_CONV_LAB_NOGATHER no patch gather in the loop, -DVLG_CONV_LAB_NOWST no weight LDS stores in the loop
#ifdef VLG_CONV_LAB_NOW
#define LAB_W_GLOAD(wr) do { } while (0)
#else
#define LAB_W_GLOAD(wr) w_gload(wr)
#endif
#ifdef VLG_CONV_LAB_NOWST
#define LAB_W_LSTORE(wr, b) do { } while (0)
#else
#define LAB_W_LSTORE(wr, b) w_lstore(wr, b)
#endif
#ifdef VLG_CONV_LAB_NOMFMA
#define LAB_COMPUTE(a, b, c) do { } while (0)
#else
#define LAB_COMPUTE(a, b, c) compute(a, b, c)
#endif
#ifdef VLG_CONV_LAB_NOGATHER
#define LAB_HALO_DMA(a, b, c) do { } while (0)
#else
#define LAB_HALO_DMA(a, b, c) halo_dma(a, b, c)
#endif
// WS (wave-specialised): 12 waves - waves 0..7 only read fragments and issue MFMAs, waves 8..11 (one per SIMD) do all the global
// loads, LDS-DMA and weight LDS stores.  The stamps and ablations of tools/conv_lab.py: a step's memory instructions cost 0.45 us on
// top of 0.96 us of fragment reads + MFMAs when they sit in the same instruction streams (each vector-memory instruction holds its
// wave at the issue stage for 100+ cycles, MFMAs of that wave behind it); in waves of their own they run beside the MFMAs.
// Chunk swizzle of a 64-byte LDS row (patch rows and weight rows): the 16-byte chunk c of row r sits at position c ^ ht_swz<T>(r).
//   fp32 (32x32x2 MFMAs, lane = (row of 32, K half)): (r >> 2) & 3, conflict-free for the 32-row fragment reads at every tap offset.
//   bf16 (round 4: 16x16x32 MFMAs, lane = (row of 16, chunk 0..3)): ((r >> 2) & 1) << 1 - searched over all 4-entry tables of (r >> 2) & 3 and
//   all 64 alignments of the first row against ds_read_b128's lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: this one is
//   conflict-free everywhere, (r >> 2) & 3 is 2-way on most alignments for that lane shape.
template <typename T>
__device__ __forceinline__ int ht_swz(int r) {
  if constexpr (sizeof(T) == 2)
    return ((r >> 2) & 1) << 1;
  else
    return (r >> 2) & 3;
}

template <typename T, int HT_TT, int HT_TH, bool WS>   // output tile: HT_TT frames x HT_TH rows x 32 columns = 256 positions (2 x 4 video, 1 x 8 images)
__global__ __launch_bounds__(WS ? 768 : 512) void conv_halo_kernel(ConvDesc d, const T* __restrict__ in, const T* __restrict__ w,
                                                               const float* __restrict__ bias, const T* __restrict__ residual,
                                                               T* __restrict__ out_cl, float* __restrict__ out_planar, double* __restrict__ gn_part) {
  constexpr int NWM = 2;
  constexpr int NTHR = 1024 / NWM;           // 512 MFMA threads
  constexpr int NL = WS ? 256 : NTHR;        // threads that load
  constexpr int HT_SLOTS = ht_slots<NL>();
  constexpr int WPT = 512 / NL;              // weight chunks per loader thread per tap: 128 rows x 4 chunks over the loaders
  constexpr int EPV = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int KC = 4 * EPV;                // channels per chunk step: one 64-byte patch row
  static_assert(HT_TT * HT_TH * HT_TW == 256, "256 positions per tile");
  constexpr int HT_HH = HT_TH + 2;
  constexpr int TSH = (HT_TH == 4) ? 7 : 8;   // log2(HT_TH * 32)
  extern __shared__ __attribute__((aligned(16))) uint4 ht_smem[];
  auto halo = [&](int buf) { return ht_smem + buf * (HT_MAXROWS * 4); };
  auto wts = [&](int buf) { return ht_smem + 2 * HT_MAXROWS * 4 + buf * (3 * 128 * 4); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_loader = !WS || tid >= NTHR;   // wave-uniform
  const int ltid = WS ? tid - NTHR : tid, lwave = ltid >> 6;               // index among the loaders
  const int wave_m = (wave >> 1) & 3, wave_n = wave & 1;
  const int r32 = lane & 31, hh = lane >> 5;
  const int n0 = blockIdx.y * 128;
  const int taps = d.kt * 9;
  const int ncc = d.Cin / KC;
#ifdef VLG_CONV_LAB
  unsigned long long* lab_rec = nullptr;
  if (conv_lab_buf != nullptr && tid == 0 &&
      (conv_lab_mode == 0 ? blockIdx.x % 61 == 7 : (d.Cin == 128 && d.Wo == 256 && d.kt == 3))) {
    const unsigned slot = atomicAdd(&conv_lab_count, 1u);
    if (slot < 32768) {
      lab_rec = conv_lab_buf + (size_t)slot * 24;
      lab_rec[0] = d.Cin;
      lab_rec[1] = d.Wo;
      lab_rec[2] = d.kt;
      lab_rec[3] = (unsigned long long)ncc * d.kt * 3;
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
      lab_rec[11] = ((unsigned long long)(xcc & 0xf) << 32) | hw;   // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    }
  }
  LAB_STAMP(4);
#endif

  // tile origin
  const int nTw = (d.Wo + HT_TW - 1) / HT_TW, nTh = (d.Ho + HT_TH - 1) / HT_TH, nTt = (d.To + HT_TT - 1) / HT_TT;
  int bid = blockIdx.x;
  const int x0 = (bid % nTw) * HT_TW;
  bid /= nTw;
  const int y0 = (bid % nTh) * HT_TH;
  bid /= nTh;
  const int t0 = (bid % nTt) * HT_TT;
  const int b = bid / nTt;

  // weight roles: row tid >> 2 of the 128-channel tile, chunk tid & 3, the three taps (a, i, 0..2) of one step; w is
  // [Cout][taps][Cin].  One step = one kernel row of taps = 24 MFMAs per wave per barrier.
  // (with 256 loader threads - WS - a thread also takes the row 64 further: WPT = 2)
  const int wrow = (ltid & 511) >> 2, wch = ltid & 3;
  const uint4* wbase = reinterpret_cast<const uint4*>(w) + ((size_t)(n0 + wrow) * taps * d.Cin) / EPV + wch;
  const size_t wrow2 = ((size_t)64 * taps * d.Cin) / EPV;   // 64 weight rows further
  const int wslot = wrow * 4 + (wch ^ ht_swz<T>(wrow));   // rows r and r + 64 share the swizzle: the second slot is wslot + 256
  const int cin8 = d.Cin / EPV;   // 16-byte chunks per (cout, tap) weight row
  const int rows_per_chunk = d.kt * 3;          // steps per channel chunk
  const int Q = ncc * rows_per_chunk;
  int l_row = 0, l_cc = 0;   // load iterator: two steps ahead of the MFMAs
  // two register sets as scalars (hipcc leaves uint4 arrays swapped between roles in scratch); 3..5 only with 256 threads
  uint4 wrA0, wrA1, wrA2, wrA3, wrA4, wrA5, wrB0, wrB1, wrB2, wrB3, wrB4, wrB5;
  wrA0 = wrA1 = wrA2 = wrA3 = wrA4 = wrA5 = wrB0 = wrB1 = wrB2 = wrB3 = wrB4 = wrB5 = make_uint4(0, 0, 0, 0);
#define w_gload(wr)                                                                   \
  do {                                                                                \
    wr##0 = wbase[(size_t)(l_row * 3 + 0) * cin8 + l_cc * 4];                         \
    wr##1 = wbase[(size_t)(l_row * 3 + 1) * cin8 + l_cc * 4];                         \
    wr##2 = wbase[(size_t)(l_row * 3 + 2) * cin8 + l_cc * 4];                         \
    if constexpr (WPT == 2) {                                                         \
      wr##3 = wbase[wrow2 + (size_t)(l_row * 3 + 0) * cin8 + l_cc * 4];               \
      wr##4 = wbase[wrow2 + (size_t)(l_row * 3 + 1) * cin8 + l_cc * 4];               \
      wr##5 = wbase[wrow2 + (size_t)(l_row * 3 + 2) * cin8 + l_cc * 4];               \
    }                                                                                 \
    if (++l_row == rows_per_chunk) {                                                  \
      l_row = 0;                                                                      \
      if (++l_cc == ncc) l_cc = 0; /* past the end: reload a valid tile, unused */    \
    }                                                                                 \
  } while (0)
#define w_lstore(wr, buf)                      \
  do {                                         \
    uint4* wb_ = wts(buf);                     \
    wb_[0 * 512 + wslot] = wr##0;              \
    wb_[1 * 512 + wslot] = wr##1;              \
    wb_[2 * 512 + wslot] = wr##2;              \
    if constexpr (WPT == 2) {                  \
      wb_[0 * 512 + 256 + wslot] = wr##3;      \
      wb_[1 * 512 + 256 + wslot] = wr##4;      \
      wb_[2 * 512 + 256 + wslot] = wr##5;      \
    }                                          \
  } while (0)

  // the weights of steps 0 and 1 are requested before the gather roles are worked out (a microsecond of integer arithmetic)
  if (is_loader) {
    w_gload(wrA);
    w_gload(wrB);
  }

  // patch gather roles (LDS-DMA, global_load_lds_dwordx4: lane l of a wave writes LDS chunk base + l): LDS chunk slot e = tid + NTHR k
  // holds patch row e >> 2; under the read-side swizzle that slot is the row's channel chunk (e & 3) ^ ((row >> 2) & 3)
  const int HF = HT_TT + d.kt - 1;
  const int nrows = HF * HT_HH * HT_HW;
  const int He = d.Hi << d.up, We = d.Wi << d.up;
  const uint4* in16 = reinterpret_cast<const uint4*>(in);
  long long hoff[HT_SLOTS];   // source chunk index (16-byte units) at channel chunk 0, -1 = zero padding, -2 = no such slot
#pragma unroll
  for (int k = 0; k < HT_SLOTS; ++k) {
    const int e = ltid + NL * k;
    const int hr = e >> 2, ch = (e & 3) ^ ht_swz<T>(hr);
    hoff[k] = -2;
    if (is_loader && hr < nrows) {
      const int f = hr / (HT_HH * HT_HW), rem = hr - f * (HT_HH * HT_HW);
      const int y = rem / HT_HW, x = rem - y * HT_HW;
      int ti = t0 + f - (d.kt - 1);
      ti = ti < 0 ? 0 : ti;                       // causal: frame 0 replicated in front (conv.py:126-129)
      ti = ti > d.Ti - 1 ? d.Ti - 1 : ti;         // frames past the end feed only rows that are never written
      const int uy = y0 + y - 1, ux = x0 + x - 1;
      hoff[k] = -1;
      if (uy >= 0 && ux >= 0 && uy < He && ux < We)
        hoff[k] = (((((long long)b * d.Ti + ti) * d.Hi + (uy >> d.up)) * d.Wi + (ux >> d.up)) * d.Cin) / EPV + ch;
    }
  }
  // The gather goes global -> LDS without registers.  (Through registers, the loads sit behind a uniform branch - once per chunk -
  // and hipcc joins the two paths with copies of the loaded registers, i.e. a wait for the loads right where they were issued: the
  // in-kernel stamps of tools/conv_lab.py showed 3.0 instead of 1.3 us for every step that starts a gather.)  Padding slots copy a
  // zero chunk.  The compiler does not see these loads: HALO_DMA_WAIT() before the barrier that publishes the patch; its own vmcnt
  // waits for the weight registers only get stricter by loads it does not count.
  const uint4* zero16 = reinterpret_cast<const uint4*>(conv_zero_chunk);
  const unsigned lds0 = (unsigned)(uintptr_t)ht_smem;   // LDS byte address of the dynamic segment
  // part / nparts: slot k goes out in call k % nparts of a chunk - one or two wave instructions per step instead of all of them in
  // one (each touches 16 separate 64-byte pieces; a batch of seven holds the waves at the issue stage: stamps 3.0 vs 1.3 us per step)
  int hpart[HT_SLOTS];   // uniform: the step of a chunk in which slot k goes out
#pragma unroll
  for (int k = 0; k < HT_SLOTS; ++k) hpart[k] = k % (d.kt * 3 - 1);
  auto halo_dma = [&](int cc, int buf, int part) __attribute__((always_inline)) {   // part < 0: every slot
#pragma unroll
    for (int k = 0; k < HT_SLOTS; ++k)
      if (hoff[k] != -2 && (part < 0 || hpart[k] == part)) {
        const uint4* src = hoff[k] >= 0 ? in16 + hoff[k] + cc * 4 : zero16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * (HT_MAXROWS * 4) + NL * k + lwave * 64) * 16));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(dst)
                     : "memory");
      }
  };
#define HALO_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

#define HALO_PATCH_SHARE(row, cc)                                                                                                     \
  do {                                                                                                                                \
    /* the next chunk's patch: a share per step (its buffer was last read before the chunk's first step), complete before the */      \
    /* barrier of the chunk's last step */                                                                                            \
    if ((row) < rows_per_chunk - 1 && (cc) + 1 < ncc) LAB_HALO_DMA((cc) + 1, ((cc) + 1) & 1, (row));                                  \
    if ((row) == rows_per_chunk - 1) HALO_DMA_WAIT();                                                                                 \
  } while (0)
  if constexpr (WS) {
    if (is_loader) {
      // the loader waves' whole life: patch of chunk 0 and the weights of steps 0 and 1, then per step the weights two steps ahead
      // into registers, the next step's weights registers -> LDS, a share of the next chunk's patch, the step's barrier
      halo_dma(0, 0, -1);
      w_lstore(wrA, 0);
      HALO_DMA_WAIT();
      __syncthreads();
      int row = 0, cc = 0;
#define VLG_LOADER_STEP(q, MINE, NEXT)                        \
  do {                                                        \
    LAB_W_GLOAD(MINE);                                        \
    if ((q) + 1 < Q) LAB_W_LSTORE(NEXT, ((q) + 1) & 1);       \
    HALO_PATCH_SHARE(row, cc);                                \
    __syncthreads();                                          \
    if (++row == rows_per_chunk) {                            \
      row = 0;                                                \
      ++cc;                                                   \
    }                                                         \
  } while (0)
      for (int q = 0; q < Q; q += 2) {
        VLG_LOADER_STEP(q, wrA, wrB);
        if (q + 1 < Q) VLG_LOADER_STEP(q + 1, wrB, wrA);
      }
#undef VLG_LOADER_STEP
      if (out_cl != nullptr) {
        __syncthreads();   // the epilogue's barrier
        if (gn_part != nullptr) {   // ... and the three of its statistics reduction
          __syncthreads();
          __syncthreads();
          __syncthreads();
        }
      }
      return;
    }
  }

  // fragment roles.  bf16: 16x16x32 MFMAs - four 16-position blocks x four 16-channel blocks per wave, lane = (row r16 of the block, 16-byte
  // chunk q of the 64-byte row): one ds_read_b128 per block and tap feeds four MFMAs of K = 32, the same LDS bytes and the same MFMA cycles
  // as the 32x32x16 form (2 K halves x 2 x 2 blocks) it replaces - but the chip holds a higher clock on this shape under load
  // (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15 x the FLOP/s at equal cycles).  fp32: 32x32x2 as before.
  constexpr bool B16 = sizeof(T) == 2;
  constexpr int NMB = B16 ? 2 * NWM : NWM, NNB = B16 ? 4 : 2;   // position blocks / channel blocks per wave
  constexpr int ACCN = B16 ? 4 : 16;
  const int r16 = lane & 15, q4 = lane >> 4;
  int hb[NMB];
#pragma unroll
  for (int mi = 0; mi < NMB; ++mi) {
    const int m = B16 ? wave_m * (32 * NWM) + mi * 16 + r16 : wave_m * (32 * NWM) + mi * 32 + r32;
    hb[mi] = ((m >> TSH) * HT_HH + ((m >> 5) & (HT_TH - 1))) * HT_HW + (m & 31);
  }
  typedef float accv_t __attribute__((ext_vector_type(ACCN)));
  accv_t acc[NMB][NNB];
#pragma unroll
  for (int mi = 0; mi < NMB; ++mi)
#pragma unroll
    for (int ni = 0; ni < NNB; ++ni)
#pragma unroll
      for (int e = 0; e < ACCN; ++e) acc[mi][ni][e] = 0.f;

  auto compute = [&](const uint4* hbuf, const uint4* wbuf, int toff) __attribute__((always_inline)) {   // toff: patch row offset of tap (a, i, 0)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if constexpr (B16) {
        uint4 af[NMB], bfr[NNB];
#pragma unroll
        for (int mi = 0; mi < NMB; ++mi) {
          const int row = hb[mi] + toff + j;
          af[mi] = hbuf[row * 4 + (q4 ^ ht_swz<T>(row))];
        }
#pragma unroll
        for (int ni = 0; ni < NNB; ++ni) {
          const int row = wave_n * 64 + ni * 16 + r16;
          bfr[ni] = wbuf[j * 512 + row * 4 + (q4 ^ ht_swz<T>(row))];
        }
#pragma unroll
        for (int mi = 0; mi < NMB; ++mi)
#pragma unroll
          for (int ni = 0; ni < NNB; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[mi]), __builtin_bit_cast(bf16x8_t, bfr[ni]),
                                                                  acc[mi][ni], 0, 0, 0);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          uint4 af[NMB], bfr[NNB];
#pragma unroll
          for (int mi = 0; mi < NMB; ++mi) {
            const int row = hb[mi] + toff + j;
            af[mi] = hbuf[row * 4 + ((2 * kk + hh) ^ ht_swz<T>(row))];
          }
#pragma unroll
          for (int ni = 0; ni < NNB; ++ni) {
            const int row = wave_n * 64 + ni * 32 + r32;
            bfr[ni] = wbuf[j * 512 + row * 4 + ((2 * kk + hh) ^ ht_swz<T>(row))];
          }
          // K = 2 per MFMA: element e of the hh = 0 lanes' chunk pairs with element e of the hh = 1 lanes' chunk - any pairing of
          // the 16 channels works as long as both operands use the same one
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mi = 0; mi < NMB; ++mi)
#pragma unroll
              for (int ni = 0; ni < NNB; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float((&af[mi].x)[e]), __uint_as_float((&bfr[ni].x)[e]),
                                                                   acc[mi][ni], 0, 0, 0);
        }
      }
    }
  };
  // accumulator element e of block (mi, ni) of this lane -> (position m of the tile, channel c of the 128)
  auto acc_pos = [&](int mi, int e) __attribute__((always_inline)) {
    return B16 ? wave_m * (32 * NWM) + mi * 16 + 4 * q4 + e : wave_m * (32 * NWM) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
  };
  auto acc_ch = [&](int ni) __attribute__((always_inline)) { return B16 ? wave_n * 64 + ni * 16 + r16 : wave_n * 64 + ni * 32 + r32; };

  // prologue: patch of chunk 0, weights of steps 0 and 1
  LAB_STAMP(5);
  if constexpr (!WS) {
    halo_dma(0, 0, -1);
    w_lstore(wrA, 0);
    HALO_DMA_WAIT();
  }
  __syncthreads();
  LAB_STAMP(6);
  LAB_CYCLES(9);

  int row = 0, cc = 0;   // compute iterator: row = a * 3 + i
  // MINE held step q's weights; they went to LDS during step q - 1, so the set is free for step q + 2.  (A macro, not a lambda
  // taking the sets by reference: hipcc keeps reference-passed register arrays in scratch.)
#define VLG_HALO_STEP(q, MINE, NEXT)                                                        \
  do {                                                                                      \
    if constexpr (!WS) LAB_W_GLOAD(MINE); /* unconditional: a branch here makes every later wait a vmcnt(0) */                        \
    LAB_COMPUTE(halo(cc & 1), wts((q) & 1), ((row / 3) * HT_HH + (row % 3)) * HT_HW);                                                 \
    if constexpr (!WS) {                                                                                                              \
      if ((q) + 1 < Q) LAB_W_LSTORE(NEXT, ((q) + 1) & 1);                                                                             \
      HALO_PATCH_SHARE(row, cc);                                                                                                      \
    }                                                                                                                                 \
    __syncthreads();                                                                        \
    if ((q) < 12) LAB_STAMP(12 + (q));                                                      \
    if (++row == rows_per_chunk) {                                                          \
      row = 0;                                                                              \
      ++cc;                                                                                 \
    }                                                                                       \
  } while (0)
  // (a third register set = weights three steps ahead: measured, no gain - the wait for them is not what stretches a step)
  for (int q = 0; q < Q; q += 2) {
    VLG_HALO_STEP(q, wrA, wrB);
    if (q + 1 < Q) VLG_HALO_STEP(q + 1, wrB, wrA);
  }
#undef VLG_HALO_STEP
#undef HALO_PATCH_SHARE
#undef HALO_DMA_WAIT
#undef w_gload
#undef w_lstore
  LAB_STAMP(7);
  LAB_CYCLES(10);

  // epilogue: + bias (+ residual) -> channels-last bf16 or planar fp32
  if (out_cl != nullptr) {
    // through LDS (free after the last step's barrier): the accumulator layout has one channel x 16 positions per lane, i.e. 2-byte
    // stores 64 B apart (measured: 10-15 us per workgroup, a fifth of the Cin = 128 layers' time).  fp32 tile [256 positions][128 + 4],
    // then every lane moves 16 bytes of consecutive channels: whole rows per 16 / 32 lanes for the residual read and the store.  The
    // arithmetic is unchanged: (acc + bias) + residual in fp32, one rounding.
    constexpr int LP = 128 + 4;
    float* Ls = reinterpret_cast<float*>(ht_smem);
    static_assert((size_t)256 * LP * sizeof(float) <= HT_LDS_BYTES, "epilogue tile fits the main loop's LDS");
    // 16 bytes of T per lane and access (8 bf16 / 4 fp32 channels): the tail is bound by the ISSUE of its global instructions
    constexpr int CH = 16 / (int)sizeof(T), IPR = 128 / CH;   // channels per item, items per 128-channel row
    constexpr int NIT = 256 * IPR / NTHR, NB = NIT < 8 ? NIT : 8;   // batches of <= 8 items: registers (fp32 has 16 items)
    long long off[NB];
    uint4 rv[NB];
    auto load_batch = [&](int k0) __attribute__((always_inline)) {   // addresses of a batch and its residual values
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int i = tid + NTHR * (k0 + k);
        const int m = i / IPR, c0 = (i % IPR) * CH;
        const int t = t0 + (m >> TSH), y = y0 + ((m >> 5) & (HT_TH - 1)), x = x0 + (m & 31);
        off[k] = (t < d.To && y < d.Ho && x < d.Wo) ? ((((long long)b * d.To + t) * d.Ho + y) * d.Wo + x) * d.Cout + n0 + c0 : -1;
      }
      if (residual) {
#pragma unroll
        for (int k = 0; k < NB; ++k) rv[k] = *reinterpret_cast<const uint4*>(residual + (off[k] >= 0 ? off[k] : 0));
      }
    };
    load_batch(0);   // in flight while the accumulators go to LDS
#pragma unroll
    for (int mi = 0; mi < NMB; ++mi)
#pragma unroll
      for (int ni = 0; ni < NNB; ++ni) {
        const int c = acc_ch(ni);
        const float bv = bias ? bias[n0 + c] : 0.f;
#pragma unroll
        for (int e = 0; e < ACCN; ++e) Ls[acc_pos(mi, e) * LP + c] = acc[mi][ni][e] + bv;
      }
    __syncthreads();
    // GroupNorm statistics of the tensor being written (round 4): a thread's items all lie in the same CH channels (i % IPR does
    // not depend on k), so it keeps CH running sums / sums of squares of the ROUNDED values it stores; reduced per group below in a fixed order
    float gs[CH], gq[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) gs[j] = gq[j] = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < NIT; k0 += NB) {
      if (k0 > 0) load_batch(k0);
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int i = tid + NTHR * (k0 + k);
        const float* lrow = Ls + (i / IPR) * LP + (i % IPR) * CH;
        float vv[CH];
#pragma unroll
        for (int j = 0; j < CH; j += 4) *reinterpret_cast<float4*>(vv + j) = *reinterpret_cast<const float4*>(lrow + j);
        uint4 ov;
        T* oe = reinterpret_cast<T*>(&ov);
        const T* re = reinterpret_cast<const T*>(&rv[k]);
#pragma unroll
        for (int j = 0; j < CH; ++j) DT<T>::st(oe + j, residual ? vv[j] + DT<T>::ld(re + j) : vv[j]);
        if (off[k] >= 0) {
          *reinterpret_cast<uint4*>(out_cl + off[k]) = ov;
          if (gn_part != nullptr) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              const float o = DT<T>::ld(oe + j);
              gs[j] += o;
              gq[j] += o * o;
            }
          }
        }
      }
    }
    if (gn_part != nullptr) {
      // per-thread partials -> LDS (the tile image is dead) -> stage 1: thread (comp, channel) adds the NTHR / IPR threads that share its
      // channel chunk, in thread order (independent LDS loads: a serial chain of dependent ones here cost 3 us per workgroup) -> stage 2:
      // thread (group, comp) adds its group's channels.  Doubles from stage 1 on; part[b][tile][group][{sum, sumsq}] as gn_finalize_kernel reads it
      __syncthreads();
      float* Ps = reinterpret_cast<float*>(ht_smem);
      constexpr int PST = CH + 1;   // row stride: the 8 chunks a wave reads in stage 1 fall on different banks
      static_assert((size_t)(2 * NTHR * PST) * sizeof(float) + 256 * sizeof(double) <= HT_LDS_BYTES, "statistics partials fit the main loop's LDS");
      double* Qs = reinterpret_cast<double*>(Ps + 2 * NTHR * PST);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        Ps[(0 * NTHR + tid) * PST + j] = gs[j];
        Ps[(1 * NTHR + tid) * PST + j] = gq[j];
      }
      __syncthreads();
      if (tid < 256) {
        const int comp = tid >> 7, ch = tid & 127, chunk = ch / CH, j = ch - chunk * CH;
        double tot = 0.0;
#pragma unroll 8
        for (int t2 = chunk; t2 < NTHR; t2 += IPR) tot += (double)Ps[(comp * NTHR + t2) * PST + j];
        Qs[tid] = tot;
      }
      __syncthreads();
      const int cg = d.Cout / 32, ngr = 128 / cg;   // channels per group, groups inside this 128-channel tile
      if (tid < 2 * ngr) {
        const int g = tid >> 1, comp = tid & 1;
        double tot = 0.0;
        for (int c = g * cg; c < (g + 1) * cg; ++c) tot += Qs[comp * 128 + c];
        const int tiles_b = nTt * nTh * nTw;
        const int tile = (int)blockIdx.x - b * tiles_b;
        gn_part[(((size_t)b * tiles_b + tile) * 32 + n0 / cg + g) * 2 + comp] = tot;
      }
    }
    LAB_STAMP(8);
    return;
  }
  const long long pper = (long long)d.To * d.Ho * d.Wo;
#pragma unroll
  for (int mi = 0; mi < NMB; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NNB; ++ni) {
      const int co = n0 + acc_ch(ni);
      const float bv = bias ? bias[co] : 0.f;
      long long pe[ACCN];
      float rv[ACCN];
#pragma unroll
      for (int e = 0; e < ACCN; ++e) {
        const int m = acc_pos(mi, e);
        const int t = t0 + (m >> TSH), y = y0 + ((m >> 5) & (HT_TH - 1)), x = x0 + (m & 31);
        pe[e] = (t < d.To && y < d.Ho && x < d.Wo) ? (((long long)b * d.To + t) * d.Ho + y) * d.Wo + x : -1;
      }
      if (residual) {   // all residual values requested before the first use
#pragma unroll
        for (int e = 0; e < ACCN; ++e) rv[e] = DT<T>::ld(residual + (pe[e] >= 0 ? pe[e] : 0) * d.Cout + co);
      }
#pragma unroll
      for (int e = 0; e < ACCN; ++e) {
        if (pe[e] < 0) continue;
        float v = acc[mi][ni][e] + bv;
        if (residual) v += rv[e];
        if (out_cl)
          DT<T>::st(out_cl + pe[e] * d.Cout + co, v);
        else
          out_planar[((pe[e] / pper) * d.Cout + co) * pper + (pe[e] % pper)] = v;
      }
    }
  }
  LAB_STAMP(8);
}

// ---------------------------------------------------------------------------------------------------------------
// Narrow-output halo convolution (round 4): the 3x3 (x kt) stride-1 causal convs with at most 4 output channels - conv_out of both
// decoders (128 -> 3, on the LARGEST activation of a decode: [4, 17, 256, 256, 128] bf16 = 1.14 GB).  On conv_mfma_kernel<32> that layer
// re-fetched every input element once per tap through L2 -> LDS (27 x 1.14 GB) for 3 useful columns of a 32-wide MFMA tile: 3.5 ms of a
// 94 ms decode call.  Here: the halo tile of conv_halo_kernel (256 output positions, the (2 + kt - 1) x 6 x 34 input patch of a
// 32-channel chunk staged in LDS once and walked by the taps), 16x16x32 MFMAs whose B operand holds the <= 4 weight rows in lanes
// r16 < NCO and zeros elsewhere, 4 waves x 4 position blocks, single-buffered (66 KB of LDS: two workgroups per compute unit overlap
// each other's gather and compute phases).  bf16 only; planar fp32 or channels-last output; no residual.
// ---------------------------------------------------------------------------------------------------------------
template <int HT_TT, int HT_TH, int NCO>
__global__ __launch_bounds__(256) void conv_narrow_kernel(ConvDesc d, const bf16* __restrict__ in, const bf16* __restrict__ w,
                                                         const float* __restrict__ bias, bf16* __restrict__ out_cl, float* __restrict__ out_planar) {
  typedef bf16 T;
  constexpr int EPV = 8, KC = 32;
  static_assert(HT_TT * HT_TH * HT_TW == 256, "256 positions per tile");
  constexpr int HT_HH = HT_TH + 2;
  constexpr int TSH = (HT_TH == 4) ? 7 : 8;
  extern __shared__ __attribute__((aligned(16))) uint4 nt_smem[];
  uint4* hbuf = nt_smem;                        // patch [816 rows][4 chunks], swizzled
  uint4* wbuf = nt_smem + HT_MAXROWS * 4;       // weights of the chunk [taps][NCO][4 chunks]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q4 = lane >> 4;
  const int taps = d.kt * 9, ncc = d.Cin / KC;
  const int nTw = (d.Wo + HT_TW - 1) / HT_TW, nTh = (d.Ho + HT_TH - 1) / HT_TH, nTt = (d.To + HT_TT - 1) / HT_TT;
  int bid = blockIdx.x;
  const int x0 = (bid % nTw) * HT_TW;
  bid /= nTw;
  const int y0 = (bid % nTh) * HT_TH;
  bid /= nTh;
  const int t0 = (bid % nTt) * HT_TT;
  const int b = bid / nTt;
  const int HF = HT_TT + d.kt - 1;
  const int nrows = HF * HT_HH * HT_HW;
  const int He = d.Hi << d.up, We = d.Wi << d.up;
  const uint4* in16 = reinterpret_cast<const uint4*>(in);
  // gather roles: LDS slot e = tid + 256 k holds patch row e >> 2, source chunk (e & 3) ^ swizzle(row)
  constexpr int NS = (HT_MAXROWS * 4 + 255) / 256;   // 13
  long long hoff[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int e = tid + 256 * k;
    const int hr = e >> 2, ch = (e & 3) ^ ht_swz<T>(hr);
    hoff[k] = -2;
    if (hr < nrows) {
      const int f = hr / (HT_HH * HT_HW), rem = hr - f * (HT_HH * HT_HW);
      const int y = rem / HT_HW, x = rem - y * HT_HW;
      int ti = t0 + f - (d.kt - 1);
      ti = ti < 0 ? 0 : ti;                       // causal: frame 0 replicated in front (conv.py:126-129)
      ti = ti > d.Ti - 1 ? d.Ti - 1 : ti;
      const int uy = y0 + y - 1, ux = x0 + x - 1;
      hoff[k] = -1;
      if (uy >= 0 && ux >= 0 && uy < He && ux < We)
        hoff[k] = (((((long long)b * d.Ti + ti) * d.Hi + (uy >> d.up)) * d.Wi + (ux >> d.up)) * d.Cin) / EPV + ch;
    }
  }
  int hb[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = wave * 64 + mi * 16 + r16;
    hb[mi] = ((m >> TSH) * HT_HH + ((m >> 5) & (HT_TH - 1))) * HT_HW + (m & 31);
  }
  typedef float f32x4n_t __attribute__((ext_vector_type(4)));
  f32x4n_t acc[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) acc[mi] = f32x4n_t{0.f, 0.f, 0.f, 0.f};
  const int nwch = taps * NCO * 4;   // weight chunks of one channel chunk
  for (int cc = 0; cc < ncc; ++cc) {
    uint4 pv[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) pv[k] = hoff[k] >= 0 ? in16[hoff[k] + cc * 4] : make_uint4(0, 0, 0, 0);
    uint4 wv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;          // (tap, co, chunk)
      const int ch = e & 3, co = (e >> 2) % NCO, tap = e / (4 * NCO);
      wv[k] = (e < nwch && co < d.Cout) ? reinterpret_cast<const uint4*>(w)[((size_t)(co * taps + tap) * d.Cin + cc * KC) / EPV + ch] : make_uint4(0, 0, 0, 0);
    }
    if (cc > 0) __syncthreads();            // the previous chunk's fragments have been read
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (hoff[k] != -2) hbuf[tid + 256 * k] = pv[k];
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 256 * k < nwch) wbuf[tid + 256 * k] = wv[k];
    __syncthreads();
    for (int tap = 0; tap < taps; ++tap) {
      const int a = tap / 9, ij = tap - a * 9;
      const int toff = (a * HT_HH + ij / 3) * HT_HW + ij % 3;
      const uint4 bfr = r16 < NCO ? wbuf[(tap * NCO + r16) * 4 + q4] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row = hb[mi] + toff;
        const uint4 af = hbuf[row * 4 + (q4 ^ ht_swz<T>(row))];
        acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bfr), acc[mi], 0, 0, 0);
      }
    }
  }
  // epilogue: lane (channel r16, q4) holds positions 4 q4 .. 4 q4 + 3 of each 16-position block = four consecutive x
  if (r16 >= d.Cout) return;
  const float bv = bias ? bias[r16] : 0.f;
  const long long pper = (long long)d.To * d.Ho * d.Wo;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = wave * 64 + mi * 16 + 4 * q4;
    const int t = t0 + (m >> TSH), y = y0 + ((m >> 5) & (HT_TH - 1)), x = x0 + (m & 31);
    if (t >= d.To || y >= d.Ho) continue;
    const long long p = (((long long)b * d.To + t) * d.Ho + y) * d.Wo + x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (x + e >= d.Wo) break;
      const float v = acc[mi][e] + bv;
      if (out_cl)
        DT<T>::st(out_cl + (p + e) * d.Cout + r16, v);
      else
        out_planar[((long long)b * d.Cout + r16) * pper + (p + e - (long long)b * pper)] = v;
    }
  }
}
constexpr int NARROW_NCO = 4;
constexpr size_t NARROW_LDS_BYTES = (size_t)(HT_MAXROWS * 4 + 27 * NARROW_NCO * 4) * sizeof(uint4);

static bool conv_halo_ok(const ConvDesc& d, int kc) {
  return d.kh == 3 && d.kw == 3 && (d.kt == 1 || d.kt == 3) && d.sh == 1 && d.tmode == 0 && d.ph0 < 0 && d.pw0 < 0 && d.Cin % kc == 0 &&
         d.Cout % 128 == 0 && d.To == d.Ti && d.Ho == (d.Hi << d.up) && d.Wo == (d.Wi << d.up);
}
size_t conv_gn_part_doubles(const ConvDesc& d) {   // the larger of the two tilings (images: 8 x 32, video: 2 x 4 x 32)
  const size_t tv = (size_t)cdiv(d.To, 2) * cdiv(d.Ho, 4) * cdiv(d.Wo, HT_TW), ti = (size_t)cdiv(d.Ho, 8) * cdiv(d.Wo, HT_TW) * (size_t)d.To;
  return (size_t)d.B * std::max(tv, ti) * 64;
}

// direct convolution for layers the MFMA kernel does not tile (Cin % 32 != 0: z_channels / codebook_embed_dim inputs)
// and for the fp32 handle dtype (parity tests at toy sizes).  One thread per (position, cout).
template <typename T>
__global__ __launch_bounds__(256) void conv_naive_kernel(ConvDesc d, const T* __restrict__ in, const T* __restrict__ w,
                                                         const float* __restrict__ bias, const T* __restrict__ residual,
                                                         T* __restrict__ out_cl, float* __restrict__ out_planar) {
  const long long ptot = (long long)d.B * d.To * d.Ho * d.Wo;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= ptot * d.Cout) return;
  const long long p = gid / d.Cout;
  const int co = (int)(gid % d.Cout);
  const PosDec pd = decode_pos(p, d, ptot);
  const int taps = d.kt * d.kh * d.kw;
  float acc = 0.f;
  int tap = 0;
  for (int a = 0; a < d.kt; ++a)
    for (int i = 0; i < d.kh; ++i)
      for (int j = 0; j < d.kw; ++j, ++tap) {
        const long long src = tap_src(pd, d, a, i, j);
        if (src < 0) continue;
        const T* xi = in + src;
        const T* wi = w + ((size_t)co * taps + tap) * d.Cin;
        for (int c = 0; c < d.Cin; ++c) acc = fmaf(DT<T>::ld(xi + c), DT<T>::ld(wi + c), acc);
      }
  float v = acc + (bias ? bias[co] : 0.f);
  if (residual) v += DT<T>::ld(residual + p * d.Cout + co);
  const long long pper = (long long)d.To * d.Ho * d.Wo;
  if (out_cl)
    DT<T>::st(out_cl + p * d.Cout + co, v);
  else
    out_planar[((p / pper) * d.Cout + co) * pper + (p % pper)] = v;
}

// ---- optional event bracket around every halo-tile conv launch (bench.py's MFMA roofline entry; off by default) ----------------
namespace {
struct ConvTimer {
  bool on = false;
  std::vector<hipEvent_t> ev;   // pairs
  size_t used = 0;
  double flops = 0;
};
ConvTimer& conv_timer() {
  static ConvTimer t;
  return t;
}
hipEvent_t conv_timer_event() {
  ConvTimer& t = conv_timer();
  if (t.used == t.ev.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    t.ev.push_back(e);
  }
  return t.ev[t.used++];
}
}  // namespace

int conv_timing_enable(bool on) {
  ConvTimer& t = conv_timer();
  if (on) {          // a new measurement starts; switching off keeps what was recorded for conv_timing_read
    t.used = 0;
    t.flops = 0;
  }
  t.on = on;
  return VLG_OK;
}
int conv_timing_read(double* ms_sum, double* flop_sum, long long* launches) {
  ConvTimer& t = conv_timer();
  double ms = 0;
  for (size_t i = 0; i + 1 < t.used; i += 2) {
    VLG_HIP(hipEventSynchronize(t.ev[i + 1]));
    float m = 0.f;
    VLG_HIP(hipEventElapsedTime(&m, t.ev[i], t.ev[i + 1]));
    ms += m;
  }
  if (ms_sum) *ms_sum = ms;
  if (flop_sum) *flop_sum = t.flops;
  if (launches) *launches = (long long)(t.used / 2);
  return VLG_OK;
}

template <typename T>
int conv_forward(const ConvDesc& d, const T* in, const T* w, const float* bias, const T* residual, T* out_cl, float* out_planar,
                 hipStream_t st, double* gn_part, int* gn_nblk) {
  if (gn_nblk) *gn_nblk = 0;
  if ((out_cl == nullptr) == (out_planar == nullptr)) {
    set_error("conv_forward: exactly one output must be given");
    return VLG_ERR_BAD_ARG;
  }
  const long long ptot = (long long)d.B * d.To * d.Ho * d.Wo;
  {
    static const bool halo_off = getenv("VLG_CONV_HALO") != nullptr && atoi(getenv("VLG_CONV_HALO")) == 0;   // A/B knob
    if (!halo_off && conv_halo_ok(d, sizeof(T) == 2 ? 32 : 16)) {
      static LdsAttrOnce attr_once;   // per instantiation of conv_forward<T>
      VLG_TRY(set_max_dynamic_lds(attr_once, {reinterpret_cast<const void*>(conv_halo_kernel<T, 2, 4, false>), reinterpret_cast<const void*>(conv_halo_kernel<T, 1, 8, false>),
                                              reinterpret_cast<const void*>(conv_halo_kernel<T, 2, 4, true>), reinterpret_cast<const void*>(conv_halo_kernel<T, 1, 8, true>)},
                                  (int)HT_LDS_BYTES));
      const bool timed = conv_timer().on;
      hipEvent_t e0 = timed ? conv_timer_event() : nullptr, e1 = timed ? conv_timer_event() : nullptr;
      if (e0 && e1) {
        conv_timer().flops += 2.0 * (double)ptot * d.kt * d.kh * d.kw * d.Cin * d.Cout;
        VLG_HIP(hipEventRecord(e0, st));
      }
      struct Stop {   // records the closing event on every exit path below
        hipEvent_t e;
        hipStream_t s;
        ~Stop() {
          if (e) (void)hipEventRecord(e, s);
        }
      } stop{(e0 && e1) ? e1 : nullptr, st};
      // loader waves of their own: bf16 only (measured on the fp32 VQ-16 decode: 53.5 ms with them, 51.6 ms without - its 32x32x2 MFMAs
      // leave the issue slots the loads need).  VLG_CONV_WS=0 / 1: A/B knob for both dtypes
      static const int ws_knob = getenv("VLG_CONV_WS") ? atoi(getenv("VLG_CONV_WS")) : -1;
      const bool ws = ws_knob < 0 ? sizeof(T) == 2 : ws_knob != 0;
      if (out_cl == nullptr || d.Cout % 32 != 0 || 128 % (d.Cout / 32) != 0) gn_part = nullptr;   // statistics ride on the channels-last epilogue only
      if (d.To == 1 && d.kt == 1) {   // images: the whole 256-position tile in one frame (patch 10 x 34 <= HT_MAXROWS)
        if (gn_part && gn_nblk) *gn_nblk = cdiv(d.Ho, 8) * cdiv(d.Wo, HT_TW);
        const dim3 grid((unsigned)((long long)d.B * cdiv(d.Ho, 8) * cdiv(d.Wo, HT_TW)), (unsigned)(d.Cout / 128));
        if (ws)
          conv_halo_kernel<T, 1, 8, true><<<grid, 768, HT_LDS_BYTES, st>>>(d, in, w, bias, residual, out_cl, out_planar, gn_part);
        else
          conv_halo_kernel<T, 1, 8, false><<<grid, 512, HT_LDS_BYTES, st>>>(d, in, w, bias, residual, out_cl, out_planar, gn_part);
      } else {
        if (gn_part && gn_nblk) *gn_nblk = cdiv(d.To, 2) * cdiv(d.Ho, 4) * cdiv(d.Wo, HT_TW);
        const dim3 grid((unsigned)((long long)d.B * cdiv(d.To, 2) * cdiv(d.Ho, 4) * cdiv(d.Wo, HT_TW)), (unsigned)(d.Cout / 128));
        if (ws)
          conv_halo_kernel<T, 2, 4, true><<<grid, 768, HT_LDS_BYTES, st>>>(d, in, w, bias, residual, out_cl, out_planar, gn_part);
        else
          conv_halo_kernel<T, 2, 4, false><<<grid, 512, HT_LDS_BYTES, st>>>(d, in, w, bias, residual, out_cl, out_planar, gn_part);
      }
      return VLG_OK;
    }
  }
  if constexpr (sizeof(T) == 2) {
    // narrow outputs (conv_out: Cout 3) on the halo tile instead of the im2col kernel: conv_narrow_kernel
    static const bool narrow_off = getenv("VLG_CONV_NARROW") != nullptr && atoi(getenv("VLG_CONV_NARROW")) == 0;   // A/B knob
    if (!narrow_off && residual == nullptr && d.Cout <= NARROW_NCO && d.kh == 3 && d.kw == 3 && (d.kt == 1 || d.kt == 3) && d.sh == 1 && d.tmode == 0 &&
        d.ph0 < 0 && d.pw0 < 0 && d.Cin % 32 == 0 && d.To == d.Ti && d.Ho == (d.Hi << d.up) && d.Wo == (d.Wi << d.up)) {
      static LdsAttrOnce narrow_once;
      VLG_TRY(set_max_dynamic_lds(narrow_once, {reinterpret_cast<const void*>(conv_narrow_kernel<2, 4, NARROW_NCO>), reinterpret_cast<const void*>(conv_narrow_kernel<1, 8, NARROW_NCO>)},
                                  (int)NARROW_LDS_BYTES));
      const bf16* in_b = reinterpret_cast<const bf16*>(in);
      const bf16* w_b = reinterpret_cast<const bf16*>(w);
      bf16* out_b = reinterpret_cast<bf16*>(out_cl);
      if (d.To == 1 && d.kt == 1)
        conv_narrow_kernel<1, 8, NARROW_NCO><<<(unsigned)((long long)d.B * cdiv(d.Ho, 8) * cdiv(d.Wo, HT_TW)), 256, NARROW_LDS_BYTES, st>>>(d, in_b, w_b, bias, out_b, out_planar);
      else
        conv_narrow_kernel<2, 4, NARROW_NCO><<<(unsigned)((long long)d.B * cdiv(d.To, 2) * cdiv(d.Ho, 4) * cdiv(d.Wo, HT_TW)), 256, NARROW_LDS_BYTES, st>>>(d, in_b, w_b, bias, out_b, out_planar);
      VLG_HIP(hipGetLastError());
      return VLG_OK;
    }
  }
  if (d.Cin % (64 / (int)sizeof(T)) == 0) {   // whole 64-byte K steps: 32 bf16 / 16 fp32 channels
    const bool wide = (d.Cout % 128 == 0);
    const int bn = wide ? 128 : 32;
    dim3 grid((unsigned)cdiv64(ptot, 128), (unsigned)cdiv(d.Cout, bn));
    if (wide)
      conv_mfma_kernel<T, 128><<<grid, 256, 0, st>>>(d, in, w, bias, residual, out_cl, out_planar);
    else
      conv_mfma_kernel<T, 32><<<grid, 256, 0, st>>>(d, in, w, bias, residual, out_cl, out_planar);
    return VLG_OK;
  }
  if (d.Cin < 64 / (int)sizeof(T)) {          // fewer channels than one K step: zero-padded K step per tap
    const bool wide = (d.Cout % 128 == 0);
    dim3 grid((unsigned)cdiv64(ptot, 128), (unsigned)cdiv(d.Cout, wide ? 128 : 32));
    if (wide)
      conv_mfma_kernel<T, 128, true><<<grid, 256, 0, st>>>(d, in, w, bias, residual, out_cl, out_planar);
    else
      conv_mfma_kernel<T, 32, true><<<grid, 256, 0, st>>>(d, in, w, bias, residual, out_cl, out_planar);
    return VLG_OK;
  }
  const long long total = ptot * d.Cout;
  conv_naive_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(d, in, w, bias, residual, out_cl, out_planar);
  return VLG_OK;
}
template int conv_forward<float>(const ConvDesc&, const float*, const float*, const float*, const float*, float*, float*, hipStream_t, double*, int*);
template int conv_forward<bf16>(const ConvDesc&, const bf16*, const bf16*, const float*, const bf16*, bf16*, float*, hipStream_t, double*, int*);

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm (32 groups) + optional swish, channels-last
//   thread -> fixed VW-channel vector (cv = tid % (C/VW)), sweeps positions; the workgroup's per-group partials are summed in a
//   fixed order (LDS, no atomics) and written to part[b][block][g][{sum, sumsq}]; gn_finalize_kernel adds the blocks in block
//   order.  Results do not depend on scheduling: two runs are bit-identical.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int N>
struct alignas(sizeof(T) * N) GnVec {
  T v[N];
};

template <typename T, int VW>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* __restrict__ part, long long P, int C,
                                                       int pos_per_block) {
  __shared__ float sm[2][256][VW + 1];
  const int b = blockIdx.y;
  const int nvec = C / VW;                // vectors per position
  const int lanes_p = 256 / nvec;         // positions handled in parallel
  const int cv = threadIdx.x % nvec, pl = threadIdx.x / nvec;
  const int cg = C / 32;
  float s[VW], q[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) s[e] = q[e] = 0.f;
  const long long p0 = (long long)blockIdx.x * pos_per_block;
  const long long p1 = (p0 + pos_per_block < P) ? p0 + pos_per_block : P;
  if (pl < lanes_p) {
    for (long long p = p0 + pl; p < p1; p += lanes_p) {
      const GnVec<T, VW> in = *reinterpret_cast<const GnVec<T, VW>*>(x + ((size_t)b * P + p) * C + cv * VW);
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        const float v = DT<T>::ld(&in.v[e]);
        s[e] += v;
        q[e] += v * v;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VW; ++e) {
    sm[0][threadIdx.x][e] = s[e];
    sm[1][threadIdx.x][e] = q[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int g = threadIdx.x >> 1, comp = threadIdx.x & 1;
    double tot = 0.0;
    for (int l = 0; l < lanes_p; ++l)
      for (int c = g * cg; c < (g + 1) * cg; ++c) tot += (double)sm[comp][l * nvec + c / VW][c % VW];
    part[(((size_t)b * gridDim.x + blockIdx.x) * 32 + g) * 2 + comp] = tot;
  }
}

// sums part[b][block][g][{sum, sumsq}] over the blocks in a fixed order (16 interleaved chains, then in chain order) and turns them
// into the group's (mean, rstd) once, in double, rounded to fp32: mr[b][g] = {mean, 1/sqrt(var + eps)}
// One workgroup per (batch element, group) - 128 of them instead of B (with the convolutions' per-tile partials nblk is 4608 on the largest
// tensors, and four workgroups summing 300 K doubles took 40 us per norm): thread t adds blocks t, t + 256, ... for both components, then the
// 256 thread sums are added in thread order.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, float2* __restrict__ mr, int nblk, double cnt,
                                                         double eps) {
  __shared__ double sm[2][256];
  const int b = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
  double su = 0.0, sq = 0.0;
  for (int k = t; k < nblk; k += 256) {
    const double2 v = *reinterpret_cast<const double2*>(part + (((size_t)b * nblk + k) * 32 + g) * 2);
    su += v.x;
    sq += v.y;
  }
  sm[0][t] = su;
  sm[1][t] = sq;
  __syncthreads();
  if (t < 2) {
    double tot = 0.0;
    for (int k = 0; k < 256; ++k) tot += sm[t][k];
    sm[t][0] = tot;
  }
  __syncthreads();
  if (t == 0) {
    const double mean = sm[0][0] / cnt;
    double var = sm[1][0] / cnt - mean * mean;
    var = var < 0 ? 0 : var;
    mr[(size_t)b * 32 + g] = make_float2((float)mean, (float)(1.0 / sqrt(var + eps)));
  }
}

// thread -> fixed VW-channel vector (scale/shift folded once: y = x * a + b with a = rstd * gamma, b = beta - mean * a... kept as
// the reference's (x - mean) * rstd * gamma + beta so results match the oracle bit for bit), sweeps the block's positions with
// four vector loads in flight.
template <typename T, int VW>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float2* __restrict__ mr, long long P,
                                                       int C, int do_swish, int pos_per_block) {
  const int b = blockIdx.y;
  const int nvec = C / VW;
  const int lanes_p = 256 / nvec;
  const int cv = threadIdx.x % nvec, pl = threadIdx.x / nvec;
  const int cg = C / 32;
  float mean[VW], rstd[VW], ga[VW], be[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) {
    const int c = cv * VW + e;
    const float2 st = mr[(size_t)b * 32 + c / cg];
    mean[e] = st.x;
    rstd[e] = st.y;
    ga[e] = gamma[c];
    be[e] = beta[c];
  }
  const long long p0 = (long long)blockIdx.x * pos_per_block;
  const long long p1 = (p0 + pos_per_block < P) ? p0 + pos_per_block : P;
  const size_t base = (size_t)b * P * C + (size_t)cv * VW;
  constexpr int UN = 4;
  for (long long p = p0 + pl; p < p1; p += (long long)UN * lanes_p) {
    GnVec<T, VW> in[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long pp = p + (long long)u * lanes_p;
      in[u] = *reinterpret_cast<const GnVec<T, VW>*>(x + base + (size_t)(pp < p1 ? pp : p) * C);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long pp = p + (long long)u * lanes_p;
      if (pp >= p1) break;
      GnVec<T, VW> o;
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        float v = (DT<T>::ld(&in[u].v[e]) - mean[e]) * rstd[e] * ga[e] + be[e];
        if (do_swish) v = swish_f(v);
        DT<T>::st(&o.v[e], v);
      }
      *reinterpret_cast<GnVec<T, VW>*>(y + base + (size_t)pp * C) = o;
    }
  }
}

size_t group_norm_scratch_bytes(int B, long long P) { return (size_t)B * 64 * sizeof(double) * (1 + (size_t)cdiv64(P, kGnPosPerBlock)); }

template <typename T>
int group_norm(const T* x, T* y, const float* gamma, const float* beta, double* stats, int B, long long P, int C, float eps,
               bool swish, hipStream_t st, const double* given_part, int given_nblk) {
  if (C % 32 != 0) {
    set_error("group_norm: C=%d not divisible by 32 groups", C);
    return VLG_ERR_BAD_SHAPE;
  }
  const int ppb = kGnPosPerBlock;
  const int nblk = (int)cdiv64(P, ppb);
  dim3 g1((unsigned)nblk, B);
  double* part = stats + (size_t)B * 64;   // scratch layout: [B][32] float2 (mean, rstd) in the first B*64 doubles' space, then
                                           // [B][nblk][64] per-block partials
  float2* mr = reinterpret_cast<float2*>(stats);
  const double cnt = (double)P * (C / 32);
  // given_part: the producing convolution already left per-tile partials [B][given_nblk][32][2] (conv_halo_kernel's epilogue): no statistics pass
  const bool have = given_part != nullptr && given_nblk > 0;
  // (apply pass: 512 positions per workgroup as the statistics pass - 1024 / 2048 / 8192 measured +1 / +4 / +18 ms per decode call, 256 / 128 within noise)
  if (C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0) {
    if (!have) gn_stats_kernel<T, 8><<<g1, 256, 0, st>>>(x, part, P, C, ppb);
    gn_finalize_kernel<<<dim3(B, 32), 256, 0, st>>>(have ? given_part : part, mr, have ? given_nblk : nblk, cnt, (double)eps);
    gn_apply_kernel<T, 8><<<g1, 256, 0, st>>>(x, y, gamma, beta, mr, P, C, swish ? 1 : 0, ppb);
  } else if (C <= 256 && 256 % C == 0) {
    if (!have) gn_stats_kernel<T, 1><<<g1, 256, 0, st>>>(x, part, P, C, ppb);
    gn_finalize_kernel<<<dim3(B, 32), 256, 0, st>>>(have ? given_part : part, mr, have ? given_nblk : nblk, cnt, (double)eps);
    gn_apply_kernel<T, 1><<<g1, 256, 0, st>>>(x, y, gamma, beta, mr, P, C, swish ? 1 : 0, ppb);
  } else {
    set_error("group_norm: unsupported channel count %d", C);
    return VLG_ERR_UNSUPPORTED;
  }
  return VLG_OK;
}
template int group_norm<float>(const float*, float*, const float*, const float*, double*, int, long long, int, float, bool, hipStream_t, const double*, int);
template int group_norm<bf16>(const bf16*, bf16*, const float*, const float*, double*, int, long long, int, float, bool, hipStream_t, const double*, int);

// ---------------------------------------------------------------------------------------------------------------
// spatial single-head attention: one wave per query row, online softmax over the frame's HW keys
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int EPL>
__global__ __launch_bounds__(256) void spatial_attn_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                           T* __restrict__ out, long long nq, int HW, int C, float scale) {
  const int lane = threadIdx.x & 63;
  const long long qi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= nq) return;
  const long long f = qi / HW;
  float qf[EPL], acc[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    qf[e] = DT<T>::ld(q + qi * C + lane * EPL + e);
    acc[e] = 0.f;
  }
  float mx = -INFINITY, l = 0.f;
  const T* kb = k + f * HW * (size_t)C + lane * EPL;
  const T* vb = v + f * HW * (size_t)C + lane * EPL;
  for (int j = 0; j < HW; ++j) {
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) d = fmaf(qf[e], DT<T>::ld(kb + (size_t)j * C + e), d);
    for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
    d *= scale;
    const float mn = fmaxf(mx, d);
    const float a = __expf(mx - mn), pj = __expf(d - mn);
    l = l * a + pj;
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * a + pj * DT<T>::ld(vb + (size_t)j * C + e);
    mx = mn;
  }
#pragma unroll
  for (int e = 0; e < EPL; ++e) DT<T>::st(out + qi * C + lane * EPL + e, acc[e] / l);
}

// ---------------------------------------------------------------------------------------------------------------
// The same attention on the matrix cores (bf16): the one place on this path where every key / value is reused by many queries
// (HW queries per frame share the frame's K and V), so K and V are staged through LDS and both products run as 32x32x16 MFMAs.
//
// Workgroup = 4 waves = 32 query rows of one frame; wave w owns the channel slice [w C/4, (w+1) C/4).  Per block of 32 keys:
//   1. K block -> LDS [32 keys][C] (16-byte chunks XOR-swizzled by key so the fragment reads are conflict-free), V block -> LDS
//      TRANSPOSED [C][40] (keys contiguous per channel: the B operand of P.V wants 8 consecutive keys of one channel per lane);
//   2. every wave: partial scores Q[:, slice] . K[:, slice]^T over its channel slice (C/64 MFMAs) -> LDS, summed over the 4 waves;
//   3. online softmax of the 32 x 32 score block by all 256 threads (8 lanes per query row, running max / sum in registers),
//      probabilities rounded to bf16 into LDS (the reference's bf16 path rounds its softmax output too, vq_model.py:343-346);
//   4. every wave: O[:, slice] = O * alpha + P . V[:, slice]  (C/64 MFMAs), the P fragments read back from LDS as the A operand.
// HW need not be a multiple of 32 (keys beyond HW are masked, rows beyond HW never stored).  fp32 handles keep the VALU kernel above.
// ---------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void spatial_attn_mfma_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                                bf16* __restrict__ out, int HW, float scale) {
  static_assert(C % 128 == 0, "four channel slices of whole 32-wide MFMA tiles");
  constexpr int CW = C / 4;          // channels per wave
  constexpr int KS = CW / 16;        // MFMA k-steps of the score product
  constexpr int NTI = CW / 32;       // 32-channel output tiles per wave
  constexpr int CH = C / 8;          // 16-byte chunks per key row
  constexpr int VP = 40;             // padded key pitch of the transposed V block (bf16 elements; 80 B: conflict-free b128 reads)
  extern __shared__ __attribute__((aligned(16))) char sa_smem[];
  uint4* Ks = reinterpret_cast<uint4*>(sa_smem);                                   // [32][CH]
  unsigned short* Vt = reinterpret_cast<unsigned short*>(sa_smem + 32 * C * 2);    // [C][VP]
  float* Sp = reinterpret_cast<float*>(sa_smem + 32 * C * 2 + C * VP * 2);         // [4][32][33]
  unsigned short* Pb = reinterpret_cast<unsigned short*>(Sp + 4 * 32 * 33);        // [32][VP]
  float* al = reinterpret_cast<float*>(Pb + 32 * VP);                              // [32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int c0 = wave * CW;
  const long long f = blockIdx.y;
  const int q0 = blockIdx.x * 32;
  const bf16* qf = q + f * HW * (long long)C;
  const bf16* kf = k + f * HW * (long long)C;
  const bf16* vf = v + f * HW * (long long)C;

  // Q fragments (A operand): lane (row r32, half hh) holds channels c0 + 16 s + 8 hh .. + 7 of its query row
  bf16x8_t qa[KS];
  {
    const int qr = (q0 + r32) < HW ? (q0 + r32) : HW - 1;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2)
      qa[s2] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qf + (long long)qr * C + c0 + 16 * s2 + 8 * hh));
  }
  f32x16_t oacc[NTI];
#pragma unroll
  for (int n = 0; n < NTI; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[n][e] = 0.f;
  // softmax roles: thread -> query row tid >> 3, key columns 4 (tid & 7) .. + 3; the row's running max / sum live in its 8 lanes
  const int srow = tid >> 3, scg = tid & 7;
  float m_run = -INFINITY, l_run = 0.f;

  const int nblk = (HW + 31) / 32;
  for (int kb = 0; kb < nblk; ++kb) {
    // ---- 1. stage K (swizzled rows) and V (transposed) ----
#pragma unroll
    for (int i = 0; i < (32 * CH) / 256; ++i) {
      const int e = tid + 256 * i;
      const int row = e / CH, ch = e - row * CH;
      int key = kb * 32 + row;
      key = key < HW ? key : HW - 1;
      const uint4 kv = *reinterpret_cast<const uint4*>(kf + (long long)key * C + ch * 8);
      const uint4 vv = *reinterpret_cast<const uint4*>(vf + (long long)key * C + ch * 8);
      Ks[row * CH + (ch ^ (row & 15))] = kv;
      const unsigned short* ve = reinterpret_cast<const unsigned short*>(&vv);
#pragma unroll
      for (int j = 0; j < 8; ++j) Vt[(ch * 8 + j) * VP + row] = ve[j];
    }
    __syncthreads();
    // ---- 2. partial scores over this wave's channel slice ----
    f32x16_t sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      const int chunk = c0 / 8 + 2 * s2 + hh;
      const bf16x8_t kfrag = __builtin_bit_cast(bf16x8_t, Ks[r32 * CH + (chunk ^ (r32 & 15))]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[s2], kfrag, sacc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) Sp[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * 33 + r32] = sacc[e];
    __syncthreads();
    // ---- 3. online softmax of the 32 x 32 block ----
    {
      float sv[4], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 4 * scg + j;
        const float t = Sp[(0 * 32 + srow) * 33 + col] + Sp[(1 * 32 + srow) * 33 + col] + Sp[(2 * 32 + srow) * 33 + col] + Sp[(3 * 32 + srow) * 33 + col];
        sv[j] = (kb * 32 + col) < HW ? t * scale : -INFINITY;
        mx = fmaxf(mx, sv[j]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 1));
      mx = fmaxf(mx, __shfl_xor(mx, 2));
      mx = fmaxf(mx, __shfl_xor(mx, 4));
      const float mnew = fmaxf(m_run, mx);          // finite: the block holds at least one key < HW
      const float alpha = __expf(m_run - mnew);      // 0 on the first block
      float ps = 0.f;
      unsigned short pb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pj = __expf(sv[j] - mnew);
        ps += pj;
        pb[j] = f32_to_bf16(pj);
      }
      ps += __shfl_xor(ps, 1);
      ps += __shfl_xor(ps, 2);
      ps += __shfl_xor(ps, 4);
      l_run = l_run * alpha + ps;
      m_run = mnew;
      *reinterpret_cast<uint2*>(Pb + srow * VP + 4 * scg) = make_uint2((unsigned)pb[0] | ((unsigned)pb[1] << 16), (unsigned)pb[2] | ((unsigned)pb[3] << 16));
      if (scg == 0) al[srow] = alpha;
    }
    __syncthreads();
    // ---- 4. O = O * alpha + P . V over this wave's channel slice ----
    {
      float a16[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) a16[e] = al[(e & 3) + 8 * (e >> 2) + 4 * hh];
#pragma unroll
      for (int n = 0; n < NTI; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[n][e] *= a16[e];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t pfrag = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Pb + r32 * VP + 16 * s2 + 8 * hh));
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
          const bf16x8_t vfrag = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Vt + (c0 + 32 * n + r32) * VP + 16 * s2 + 8 * hh));
          oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pfrag, vfrag, oacc[n], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  if (scg == 0) al[srow] = 1.0f / l_run;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
    if (q0 + row >= HW) continue;
    const float inv = al[row];
    bf16* op = out + (f * HW + q0 + row) * (long long)C + c0 + r32;
#pragma unroll
    for (int n = 0; n < NTI; ++n) op[32 * n].v = f32_to_bf16(oacc[n][e] * inv);
  }
}

template <int C>
static int spatial_attention_mfma(const bf16* q, const bf16* k, const bf16* v, bf16* out, int NF, int HW, float scale, hipStream_t st) {
  const size_t lds = (size_t)32 * C * 2 + (size_t)C * 40 * 2 + (size_t)4 * 32 * 33 * 4 + (size_t)32 * 40 * 2 + 32 * 4;
  static LdsAttrOnce attr_once;   // per instantiation
  VLG_TRY(set_max_dynamic_lds(attr_once, {reinterpret_cast<const void*>(spatial_attn_mfma_kernel<C>)}, 160 * 1024));
  spatial_attn_mfma_kernel<C><<<dim3((unsigned)cdiv(HW, 32), (unsigned)NF), 256, lds, st>>>(q, k, v, out, HW, scale);
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}

template <typename T>
int spatial_attention(const T* q, const T* k, const T* v, T* out, int NF, int HW, int C, hipStream_t st) {
  const long long nq = (long long)NF * HW;
  const float scale = 1.0f / sqrtf((float)C);   // int(c) ** (-0.5)
  if constexpr (sizeof(T) == 2) {
    static const bool mfma_off = getenv("VLG_ATTN_MFMA") != nullptr && atoi(getenv("VLG_ATTN_MFMA")) == 0;   // A/B knob
    if (!mfma_off) {
      if (C == 512) return spatial_attention_mfma<512>(q, k, v, out, NF, HW, scale, st);
      if (C == 256) return spatial_attention_mfma<256>(q, k, v, out, NF, HW, scale, st);
      if (C == 128) return spatial_attention_mfma<128>(q, k, v, out, NF, HW, scale, st);
    }
  }
  dim3 grid((unsigned)cdiv64(nq, 4));
  if (C == 512)
    spatial_attn_kernel<T, 8><<<grid, 256, 0, st>>>(q, k, v, out, nq, HW, C, scale);
  else if (C == 256)
    spatial_attn_kernel<T, 4><<<grid, 256, 0, st>>>(q, k, v, out, nq, HW, C, scale);
  else if (C == 128)
    spatial_attn_kernel<T, 2><<<grid, 256, 0, st>>>(q, k, v, out, nq, HW, C, scale);
  else if (C == 64)
    spatial_attn_kernel<T, 1><<<grid, 256, 0, st>>>(q, k, v, out, nq, HW, C, scale);
  else {
    set_error("spatial_attention: unsupported channel count %d", C);
    return VLG_ERR_UNSUPPORTED;
  }
  return VLG_OK;
}
template int spatial_attention<float>(const float*, const float*, const float*, float*, int, int, int, hipStream_t);
template int spatial_attention<bf16>(const bf16*, const bf16*, const bf16*, bf16*, int, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// tokenizer_video VQ-VAE decoder pieces
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ rm, const float* __restrict__ rv,
                                                      long long total, int C, int relu) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  float v = (DT<T>::ld(x + i) - rm[c]) / sqrtf(rv[c] + 1e-5f) * gamma[c] + beta[c];
  if (relu) v = fmaxf(v, 0.f);
  DT<T>::st(y + i, v);
}
template <typename T>
int bn_relu(const T* x, T* y, const float* gamma, const float* beta, const float* rm, const float* rv, long long n_pos, int C, bool relu,
            hipStream_t st) {
  const long long total = n_pos * C;
  bn_relu_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(x, y, gamma, beta, rm, rv, total, C, relu ? 1 : 0);
  return VLG_OK;
}
template int bn_relu<float>(const float*, float*, const float*, const float*, const float*, const float*, long long, int, bool, hipStream_t);
template int bn_relu<bf16>(const bf16*, bf16*, const float*, const float*, const float*, const float*, long long, int, bool, hipStream_t);

// out[o] = bias + sum over taps k, ci with padded-input index i = (o + 3 - k) / 2 (when even), original index i - 1
template <typename T>
__global__ __launch_bounds__(256) void convt_k4s2_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ out_cl, float* __restrict__ out_planar, int B, int Ti, int Hi, int Wi,
                                                         int Cin, int Cout, int relu) {
  const int To = 2 * Ti, Ho = 2 * Hi, Wo = 2 * Wi;
  const long long ptot = (long long)B * To * Ho * Wo;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= ptot * Cout) return;
  const int co = (int)(gid % Cout);
  long long p = gid / Cout;
  const long long pp = p;
  const int ox = (int)(p % Wo);
  p /= Wo;
  const int oy = (int)(p % Ho);
  p /= Ho;
  const int ot = (int)(p % To);
  const int b = (int)(p / To);
  float acc = bias ? bias[co] : 0.f;
  for (int a = 0; a < 4; ++a) {
    const int nt = ot + 3 - a;
    if (nt & 1) continue;
    const int it = nt / 2 - 1;
    if (it < 0 || it >= Ti) continue;
    for (int i = 0; i < 4; ++i) {
      const int ny = oy + 3 - i;
      if (ny & 1) continue;
      const int iy = ny / 2 - 1;
      if (iy < 0 || iy >= Hi) continue;
      for (int j = 0; j < 4; ++j) {
        const int nx = ox + 3 - j;
        if (nx & 1) continue;
        const int ix = nx / 2 - 1;
        if (ix < 0 || ix >= Wi) continue;
        const T* xi = in + ((((long long)b * Ti + it) * Hi + iy) * Wi + ix) * Cin;
        const T* wi = w + ((size_t)co * 64 + (a * 16 + i * 4 + j)) * Cin;
        for (int c = 0; c < Cin; ++c) acc = fmaf(DT<T>::ld(xi + c), DT<T>::ld(wi + c), acc);
      }
    }
  }
  if (relu) acc = fmaxf(acc, 0.f);
  const long long pper = (long long)To * Ho * Wo;
  if (out_cl)
    DT<T>::st(out_cl + pp * Cout + co, acc);
  else
    out_planar[((pp / pper) * Cout + co) * pper + (pp % pper)] = acc;
}
template <typename T>
int conv_transpose_k4s2(const T* in, const T* w, const float* bias, T* out_cl, float* out_planar, int B, int Ti, int Hi, int Wi, int Cin,
                        int Cout, bool relu, hipStream_t st) {
  const long long total = (long long)B * 8 * Ti * Hi * Wi * Cout;
  convt_k4s2_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(in, w, bias, out_cl, out_planar, B, Ti, Hi, Wi, Cin, Cout, relu ? 1 : 0);
  return VLG_OK;
}
template int conv_transpose_k4s2<float>(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, bool, hipStream_t);
template int conv_transpose_k4s2<bf16>(const bf16*, const bf16*, const float*, bf16*, float*, int, int, int, int, int, int, bool, hipStream_t);

// one wave per (position, head): online softmax over the L positions along `axis`
template <typename T>
__global__ __launch_bounds__(256) void axial_attn_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                         T* __restrict__ out, long long nq, int Tn, int H, int W, int nh, int dk, int axis) {
  const int lane = threadIdx.x & 63;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= nq) return;
  const int head = (int)(wid % nh);
  const long long pos = wid / nh;   // flat (b,t,h,w)
  const int C = nh * dk;
  const int x = (int)(pos % W), y = (int)((pos / W) % H), tt = (int)((pos / ((long long)W * H)) % Tn);
  const long long b = pos / ((long long)W * H * Tn);
  const int L = axis == 1 ? Tn : (axis == 2 ? H : W);
  const long long stride = axis == 1 ? (long long)H * W : (axis == 2 ? W : 1);
  const long long base = (b * Tn * H * W) + (axis == 1 ? (long long)y * W + x : (axis == 2 ? (long long)tt * H * W + x : ((long long)tt * H + y) * W));
  const int e0 = lane * 2;          // dk <= 128: two elements per lane
  float q0 = 0.f, q1 = 0.f;
  if (e0 < dk) q0 = DT<T>::ld(q + pos * C + head * dk + e0);
  if (e0 + 1 < dk) q1 = DT<T>::ld(q + pos * C + head * dk + e0 + 1);
  const float scale = 1.0f / sqrtf((float)dk);
  float mx = -INFINITY, l = 0.f, a0 = 0.f, a1 = 0.f;
  for (int j = 0; j < L; ++j) {
    const long long pj = base + j * stride;
    float d = 0.f;
    if (e0 < dk) d += q0 * DT<T>::ld(k + pj * C + head * dk + e0);
    if (e0 + 1 < dk) d += q1 * DT<T>::ld(k + pj * C + head * dk + e0 + 1);
    for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
    d *= scale;
    const float mn = fmaxf(mx, d);
    const float al = __expf(mx - mn), pj2 = __expf(d - mn);
    l = l * al + pj2;
    a0 = a0 * al + (e0 < dk ? pj2 * DT<T>::ld(v + pj * C + head * dk + e0) : 0.f);
    a1 = a1 * al + (e0 + 1 < dk ? pj2 * DT<T>::ld(v + pj * C + head * dk + e0 + 1) : 0.f);
    mx = mn;
  }
  if (e0 < dk) DT<T>::st(out + pos * C + head * dk + e0, a0 / l);
  if (e0 + 1 < dk) DT<T>::st(out + pos * C + head * dk + e0 + 1, a1 / l);
}
template <typename T>
int axial_attention(const T* q, const T* k, const T* v, T* out, int B, int T_, int H, int W, int nh, int dk, int axis, hipStream_t st) {
  if (dk > 128 || axis < 1 || axis > 3) {
    set_error("axial_attention: unsupported head dim %d / axis %d", dk, axis);
    return VLG_ERR_UNSUPPORTED;
  }
  const long long nq = (long long)B * T_ * H * W * nh;
  axial_attn_kernel<T><<<dim3((unsigned)cdiv64(nq, 4)), 256, 0, st>>>(q, k, v, out, nq, T_, H, W, nh, dk, axis);
  return VLG_OK;
}
template int axial_attention<float>(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, hipStream_t);
template int axial_attention<bf16>(const bf16*, const bf16*, const bf16*, bf16*, int, int, int, int, int, int, int, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void add4_kernel(const T* __restrict__ r, const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                                   T* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  DT<T>::st(out + i, DT<T>::ld(r + i) + ((DT<T>::ld(a + i) + DT<T>::ld(b + i)) + DT<T>::ld(c + i)));
}
template <typename T>
int add4(const T* r, const T* a, const T* b, const T* c, T* out, long long n, hipStream_t st) {
  add4_kernel<T><<<dim3((unsigned)cdiv64(n, 256)), 256, 0, st>>>(r, a, b, c, out, n);
  return VLG_OK;
}
template int add4<float>(const float*, const float*, const float*, const float*, float*, long long, hipStream_t);
template int add4<bf16>(const bf16*, const bf16*, const bf16*, const bf16*, bf16*, long long, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void relayout_wt_kernel(const float* __restrict__ src, T* __restrict__ dst, int Cin, int Cout, int taps, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  const long long r = i / Cin;
  const int tap = (int)(r % taps);
  const long long co = r / taps;
  DT<T>::st(dst + i, src[((long long)ci * Cout + co) * taps + tap]);
}
template <typename T>
int relayout_convt_weight(const float* src, T* dst, int Cin, int Cout, int taps, hipStream_t st) {
  const long long total = (long long)Cin * Cout * taps;
  relayout_wt_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(src, dst, Cin, Cout, taps, total);
  return VLG_OK;
}
template int relayout_convt_weight<float>(const float*, float*, int, int, int, hipStream_t);
template int relayout_convt_weight<bf16>(const float*, bf16*, int, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// layout glue
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void q12_permute_kernel(const T* __restrict__ x, T* __restrict__ y, int Tn, long long HW, int C,
                                                          int inverse, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long long r = i / C;
  const long long hw = r % HW;
  r /= HW;
  const int t = (int)(r % Tn);
  const long long b = r / Tn;
  int ts, cs;
  if (!inverse) {  // y[b][t'][hw][c'] = x[b][(t'C+c') % T][hw][(t'C+c') / T]
    const long long j = (long long)t * C + c;
    ts = (int)(j % Tn);
    cs = (int)(j / Tn);
  } else {         // y[b][t][hw][c] = x[b][(cT+t) / C][hw][(cT+t) % C]
    const long long j = (long long)c * Tn + t;
    ts = (int)(j / C);
    cs = (int)(j % C);
  }
  y[i] = x[((b * Tn + ts) * HW + hw) * C + cs];
}
template <typename T>
int q12_permute(const T* x, T* y, int B, int T_, int HW, int C, bool inverse, hipStream_t st) {
  const long long total = (long long)B * T_ * HW * C;
  q12_permute_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(x, y, T_, HW, C, inverse ? 1 : 0, total);
  return VLG_OK;
}
template int q12_permute<float>(const float*, float*, int, int, int, int, bool, hipStream_t);
template int q12_permute<bf16>(const bf16*, bf16*, int, int, int, int, bool, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void time_up_kernel(const T* __restrict__ x, T* __restrict__ y, int Tn, long long HWC, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int To = 2 * Tn - 1;
  const long long e = i % HWC;
  const long long r = i / HWC;
  const int to = (int)(r % To);
  const long long b = r / To;
  const T* xb = x + b * Tn * HWC + e;
  if (to == 0) {
    y[i] = xb[0];
    return;
  }
  const int j = to - 1, nrest = Tn - 1;
  float src = ((float)j + 0.5f) * 0.5f - 0.5f;           // align_corners=False source coordinate
  src = src < 0.f ? 0.f : src;
  const int i0 = (int)src;
  const int i1 = i0 + 1 < nrest ? i0 + 1 : nrest - 1;
  const float lam = src - (float)i0;
  const float a = DT<T>::ld(xb + (size_t)(1 + i0) * HWC), c = DT<T>::ld(xb + (size_t)(1 + i1) * HWC);
  DT<T>::st(y + i, (1.0f - lam) * a + lam * c);
}
template <typename T>
int time_upsample2x(const T* x, T* y, int B, int T_, long long HWC, hipStream_t st) {
  if (T_ <= 1) {
    hipError_t e = hipMemcpyAsync(y, x, (size_t)B * HWC * sizeof(T), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
      set_error("hipMemcpyAsync: %s", hipGetErrorString(e));
      return VLG_ERR_HIP;
    }
    return VLG_OK;
  }
  const long long total = (long long)B * (2 * T_ - 1) * HWC;
  time_up_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(x, y, T_, HWC, total);
  return VLG_OK;
}
template int time_upsample2x<float>(const float*, float*, int, int, long long, hipStream_t);
template int time_upsample2x<bf16>(const bf16*, bf16*, int, int, long long, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void time_down_kernel(const T* __restrict__ x, T* __restrict__ y, int Tn, int To, long long HWC, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long e = i % HWC;
  const long long r = i / HWC;
  const int to = (int)(r % To);
  const long long b = r / To;
  const T* xb = x + b * Tn * HWC + e;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int ti = 2 * to + k - 2;          // two replicated copies of frame 0 in front
    ti = ti < 0 ? 0 : ti;
    s += DT<T>::ld(xb + (size_t)ti * HWC);
  }
  DT<T>::st(y + i, s / 3.0f);
}
template <typename T>
int time_downsample2x(const T* x, T* y, int B, int T_, long long HWC, hipStream_t st) {
  const int To = (T_ - 1) / 2 + 1;
  const long long total = (long long)B * To * HWC;
  time_down_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(x, y, T_, To, HWC, total);
  return VLG_OK;
}
template int time_downsample2x<float>(const float*, float*, int, int, long long, hipStream_t);
template int time_downsample2x<bf16>(const bf16*, bf16*, int, int, long long, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void cl_to_planar_kernel(const T* __restrict__ x, float* __restrict__ y, int C, long long P, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long r = i / C;
  const long long p = r % P;
  const long long b = r / P;
  y[(b * C + c) * P + p] = DT<T>::ld(x + i);
}
template <typename T>
int cl_to_planar_f32(const T* x, float* y, int B, int C, long long P, hipStream_t st) {
  const long long total = (long long)B * C * P;
  cl_to_planar_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(x, y, C, P, total);
  return VLG_OK;
}
template int cl_to_planar_f32<float>(const float*, float*, int, int, long long, hipStream_t);
template int cl_to_planar_f32<bf16>(const bf16*, float*, int, int, long long, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void planar_to_cl_kernel(const float* __restrict__ x, T* __restrict__ y, int C, long long P, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long r = i / C;
  const long long p = r % P;
  const long long b = r / P;
  DT<T>::st(y + i, x[(b * C + c) * P + p]);
}
template <typename T>
int planar_f32_to_cl(const float* x, T* y, int B, int C, long long P, hipStream_t st) {
  const long long total = (long long)B * C * P;
  planar_to_cl_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(x, y, C, P, total);
  return VLG_OK;
}
template int planar_f32_to_cl<float>(const float*, float*, int, int, long long, hipStream_t);
template int planar_f32_to_cl<bf16>(const float*, bf16*, int, int, long long, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void relayout_w_kernel(const float* __restrict__ src, T* __restrict__ dst, int Cin, int taps, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i % Cin);
  const long long r = i / Cin;
  const int tap = (int)(r % taps);
  const long long co = r / taps;
  DT<T>::st(dst + i, src[(co * Cin + ci) * taps + tap]);
}
template <typename T>
int relayout_conv_weight(const float* src, T* dst, int Cout, int Cin, int taps, hipStream_t st) {
  const long long total = (long long)Cout * Cin * taps;
  relayout_w_kernel<T><<<dim3((unsigned)cdiv64(total, 256)), 256, 0, st>>>(src, dst, Cin, taps, total);
  return VLG_OK;
}
template int relayout_conv_weight<float>(const float*, float*, int, int, int, hipStream_t);
template int relayout_conv_weight<bf16>(const float*, bf16*, int, int, int, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void codebook_lookup_kernel(const float* __restrict__ E, const int32_t* __restrict__ codes,
                                                              T* __restrict__ out, long long n, int n_e, int e_dim, int l2norm) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int id = codes[i];
  id = id < 0 ? 0 : (id >= n_e ? n_e - 1 : id);
  const float* row = E + (size_t)id * e_dim;
  float inv = 1.f;
  if (l2norm) {
    float ss = 0.f;
    for (int c = 0; c < e_dim; ++c) ss += row[c] * row[c];
    inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize eps
  }
  for (int c = 0; c < e_dim; ++c) DT<T>::st(out + i * e_dim + c, row[c] * inv);
}
template <typename T>
int codebook_lookup(const float* E, const int32_t* codes, T* out, long long n, int n_e, int e_dim, bool l2norm, hipStream_t st) {
  codebook_lookup_kernel<T><<<dim3((unsigned)cdiv64(n, 256)), 256, 0, st>>>(E, codes, out, n, n_e, e_dim, l2norm ? 1 : 0);
  return VLG_OK;
}
template int codebook_lookup<float>(const float*, const int32_t*, float*, long long, int, int, bool, hipStream_t);
template int codebook_lookup<bf16>(const float*, const int32_t*, bf16*, long long, int, int, bool, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// codebook nearest neighbour: one wave per z row, lanes stride over the codes, wavefront arg-min (first minimum)
//   l2norm = 1: d = (|z|^2 + |e|^2) - 2 z.e on row-normalised z, e       (vq_model.py:221-232)
//   l2norm = 0: d = (|z|^2 - 2 z.e) + |e|^2                              (tokenizer_video/vqvae.py:166-168)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmin_kernel(const float* __restrict__ z, long long zs_row, long long zs_c, long long rows_per_batch,
                                                     long long zs_batch, const float* __restrict__ E, long long n, int n_e, int dim,
                                                     int l2norm, int32_t* __restrict__ idx) {
  extern __shared__ float zsm[];  // [4][dim]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wv;
  float* zr = zsm + (size_t)wv * dim;
  float zz = 0.f;
  if (row < n) {
    const float* zp = z + (row / rows_per_batch) * zs_batch + (row % rows_per_batch) * zs_row;
    float ss = 0.f;
    for (int c = lane; c < dim; c += 64) {
      const float v = zp[(size_t)c * zs_c];
      zr[c] = v;
      ss += v * v;
    }
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    zz = ss;
    if (l2norm) {
      const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
      __builtin_amdgcn_wave_barrier();
      float s2 = 0.f;
      for (int c = lane; c < dim; c += 64) {
        zr[c] *= inv;
        s2 += zr[c] * zr[c];
      }
      for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
      zz = s2;
    }
  }
  __syncthreads();
  if (row >= n) return;
  float best = INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < n_e; j += 64) {
    const float* e = E + (size_t)j * dim;
    float ee = 0.f, dot = 0.f;
    float inv = 1.f;
    if (l2norm) {
      float s = 0.f;
      for (int c = 0; c < dim; ++c) s += e[c] * e[c];
      inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    }
    for (int c = 0; c < dim; ++c) {
      const float ev = e[c] * inv;
      ee += ev * ev;
      dot = fmaf(zr[c], ev, dot);
    }
    const float d = l2norm ? __fsub_rn(__fadd_rn(zz, ee), __fmul_rn(2.0f, dot)) : __fadd_rn(__fsub_rn(zz, __fmul_rn(2.0f, dot)), ee);
    if (d < best) {
      best = d;
      bi = j;
    }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ob < best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if (lane == 0) idx[row] = bi;
}

int codebook_argmin(const float* z, long long zs_row, long long zs_c, long long rows_per_batch, long long zs_batch, const float* E,
                    long long n, int n_e, int dim, bool l2norm, int32_t* idx, hipStream_t st) {
  if (dim > 4096) {
    set_error("codebook_argmin: dim %d too large", dim);
    return VLG_ERR_UNSUPPORTED;
  }
  argmin_kernel<<<dim3((unsigned)cdiv64(n, 4)), 256, (size_t)4 * dim * sizeof(float), st>>>(z, zs_row, zs_c, rows_per_batch, zs_batch, E,
                                                                                             n, n_e, dim, l2norm ? 1 : 0, idx);
  return VLG_OK;
}

}  // namespace vlg
