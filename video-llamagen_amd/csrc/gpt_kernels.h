// Launchers for the decode/prefill kernels of the GPT path (gfx950).  All asynchronous on `st`.
#pragma once
#include "common.h"

namespace vlg {

// Position/step state that lives in device memory so one captured graph serves every decode step.
struct StepState {
  int32_t pos;    // absolute input position of the current step's query rows (first row when Tq > 1)
  int32_t step;   // index of the token being produced (0 = prefill output)
};
// Iteration-level batching (vlg_gpt_session_*): every batch row at its own position / token index.  The per-row arrays are KERNEL
// ARGUMENTS (null = uniform mode): a pointer kept inside StepState would put a dependent load in front of every kernel's first
// address computation (measured: +1 us on the QKV GEMM).

// Block-granular KV cache (vlg_gpt_session_* with option kv_block): per layer the cache is a POOL [n_blocks][H][BS][hd] instead of
// [rows][H][S][hd]; batch row b keeps positions j*BS .. j*BS+BS-1 in pool block table[b * stride + j].  table == nullptr: contiguous.
// (serve/gpt_model.py:181-224 runs on vLLM's paged KV; block 0 is a scratch block every idle row's table points at.)
struct KvPages {
  const int32_t* table = nullptr;
  int stride = 0;   // table entries per batch row (<= 256)
  int shift = 0;    // log2(BS)
};
// index of cache row (b, h, p) in units of hd elements
__host__ __device__ inline size_t kv_row_index(const KvPages& pg, int b, int h, int H, int S, int p) {
  if (pg.table == nullptr) return ((size_t)b * H + h) * S + p;
  const size_t blk = (size_t)pg.table[(size_t)b * pg.stride + (p >> pg.shift)];
  return (((blk * H + h) << pg.shift) | (size_t)(p & ((1 << pg.shift) - 1)));
}

enum Act { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_SILU = 2 };

// ---- GEMM: slabs[s][M][N] (fp32 partial sums over a K slice) = x[M,K] @ w[N,K]^T -----------------
// returns the number of slabs written through *splits (>= 1).  `ws` must hold max_splits*M*N floats.
template <typename T>
int gemm_slabs(const T* x, const T* w, float* ws, int M, int N, int K, int* splits, hipStream_t st, const T* wfm = nullptr);
// FRAGMENT-MAJOR copy of a Linear weight [N][K] (N % 16 == 0, K * sizeof(T) % 64 == 0).  nn.Linear keeps K contiguous per output row,
// and an MFMA B fragment of 16 output columns x 64 bytes of K is then 16 pieces of 64 bytes from 16 rows: every load instruction of a
// wave touches 16 HALF cache lines, and a cold weight stream runs 1.5-1.9x slower than one made of whole lines
// (tools/microbench/frag_lab.hip: GPT-3B w13, 111 MB, 272 workgroups: 39.4 vs 26.1 us).  Here block (t, s) = the fragment of rows
// 16 t .. 16 t + 15, K bytes [64 s, 64 s + 64) is 1 KB contiguous at ((t * nks + s) * 1024), lane l = r + 16 q (row r, 16-byte chunk q)
// at l * 16: one wave instruction = 8 whole lines.  A pure permutation: same values, same MFMA operands, same results.
template <typename T>
int relayout_fragment_major(const T* w, T* out, int N, int K, hipStream_t st);
inline bool fragment_major_ok(int N, int K, int esz) { return N % 16 == 0 && ((long long)K * esz) % 64 == 0; }
int gemm_max_splits();
// fused SwiGLU up-projection: g = rt(rt(silu(rt(x w1^T))) * rt(x w3^T)), w13 = [w1; w3]; false -> use the generic path
template <typename T>
bool gemm_swiglu(const T* x, const T* w13, T* g, int M, int F, int K, hipStream_t st);
size_t gemm_ws_floats(int M, int N, int K, int elem_size);

// ---- fused skinny GEMM (gemm_fused.hip): prologue RMSNorm + epilogue residual / RoPE+scatter / SwiGLU / store -------------
enum FusedEpi { EPI_RESID = 0, EPI_QKV = 1, EPI_SWIGLU = 2, EPI_STORE = 3, EPI_GATED = 4 };
// Activation matrices of the fused decode chain kept A-FRAGMENT-MAJOR ("afm"): element (row, col) of an [M][N] matrix of T sits in the
// 1 KB block (row / 16, K step = col * sizeof(T) / 64) at 16-byte slot (row % 16) + 16 * (chunk of the step), so that the MFMA A fragment of
// 16 rows x 64 bytes - the unit every skinny GEMM of the chain loads - is 1 KB contiguous instead of 16 pieces of 64 bytes.  Measured
// (tools/microbench/act_lab.hip, the 32-row QKV GEMM's load pattern, 240 workgroups, whole-line weights): 8.4 -> 4.4 us per launch.
// Rows are padded to a multiple of 16.  The same values in another order: results do not change.
template <typename T>
__host__ __device__ inline size_t afm_index(int row, int col, int nks) {
  constexpr int EPS = 64 / (int)sizeof(T), EPC = 16 / (int)sizeof(T);
  return (((size_t)(row >> 4) * nks + col / EPS) * 64 + (row & 15) + 16 * ((col % EPS) / EPC)) * EPC + (col % EPC);
}
struct FusedGemm {
  const void* wfm = nullptr;      // fragment-major copy of the weight (relayout_fragment_major): when set, the kernel streams it instead of w
  int a_fm = 0;                   // the A operand x [M, K] is A-fragment-major (afm_index)
  int o_fm = 0;                   // the T-typed result matrix (EPI_RESID / EPI_GATED h, EPI_SWIGLU / EPI_STORE out) [M, N] is A-fragment-major
  const void* norm_w = nullptr;   // PRO: RMSNorm weight [K] (dtype T)
  float eps = 1e-5f;
  void* h = nullptr;              // EPI_RESID / EPI_GATED: residual stream [M,N], updated in place
  const void* gate = nullptr;     // EPI_GATED: h = rt(h + rt(gate[m*gate_stride + n] * rt(acc + bias)))   (diffloss.py:128)
  int gate_stride = 0;
  void* qbuf = nullptr;           // EPI_QKV: q [M,H,hd]; caches [Bp,H,S,hd]
  void* kc = nullptr;
  void* vc = nullptr;
  const float* freqs = nullptr;
  const StepState* state = nullptr;
  const int32_t* row_pos = nullptr;   // EPI_QKV, sessions: position of batch row b instead of state->pos
  int Tq = 1, H = 0, hd = 0, S = 0;
  KvPages pages;                  // EPI_QKV: block-granular cache (sessions)
  void* out = nullptr;            // EPI_SWIGLU: g [M,N]; EPI_STORE: out [M,N] (T)
  float* out_f32 = nullptr;       // EPI_STORE
  const void* bias = nullptr;     // EPI_STORE (dtype T)
  int act = 0;
#ifdef VLG_KTRACE
  unsigned long long* trace = nullptr;   // tools/microbench only: 4 timestamps per workgroup
#endif
};
template <typename T>
bool gemm_fused_ok(int M, int N, int K, bool pro, int epi);
template <typename T>
int gemm_fused(const T* x, const T* w, int M, int N, int K, bool pro, int epi, const FusedGemm& fa, hipStream_t st);

// LayerNorm + adaLN-modulate prologue (DiffLoss ResBlock / FinalLayer, diffloss.py:55-56,120-128,141-148) fused into the skinny GEMM:
//   out = rt(act(rt(bias + W . rt(rt(LN(x) [* ln_w + ln_b]) * (1 + scale[row]) + shift[row]))))
// shift / scale: [M] rows of stride mod_stride; ln_w / ln_b null for elementwise_affine = False.
struct LnGemm {
  const void* ln_w = nullptr;
  const void* ln_b = nullptr;
  const void* shift = nullptr;
  const void* scale = nullptr;
  int mod_stride = 0;
  float eps = 1e-6f;
  void* out = nullptr;
  const void* bias = nullptr;
  int act = 0;
};
template <typename T>
bool gemm_ln_fused_ok(int M, int N, int K);
template <typename T>
int gemm_ln_fused(const T* x, const T* w, int M, int N, int K, const LnGemm& fa, hipStream_t st);

// out[m][n] = rt(act(rt(sum_s slab[s][m][n])));  out_f32 (optional) receives float(rt(sum)) (gpt.py:371)
// bias (optional, dtype T, [N]) is added to the fp32 sum before the first rounding (nn.Linear with bias)
template <typename T>
int reduce_store(const float* ws, int splits, T* out, float* out_f32, int M, int N, int act, hipStream_t st, const T* bias = nullptr);

// h[m] = rt(h[m] + rt(sum slabs));  hn[m] = rmsnorm(h[m]) * w      (gpt.py:257-258 + :146-148)
// if ws == nullptr only the norm is computed (first layer).
template <typename T>
int reduce_residual_rmsnorm(const float* ws, int splits, T* h, const T* w, T* hn, int M, int D, float eps, hipStream_t st);

// g[m][n] = rt(rt(silu(rt(sum a))) * rt(sum b)), slab rows are [w1 | w3] outputs, N2 = 2F   (gpt.py:167)
template <typename T>
int reduce_silu_mul(const float* ws, int splits, T* g, int M, int F, hipStream_t st);

// qkv slabs [s][M][3D] -> rope(q) to qbuf [M,H,hd]; rope(k), v into the caches at pos0 + (m % Tq)
// caches: [Bp, H, S, hd]; rows m = b*Tq + t.  freqs: fp32 [npos, hd/2, 2]          (gpt.py:215-227)
template <typename T>
int qkv_rope_scatter(const float* ws, int splits, T* qbuf, T* kcache, T* vcache, const float* freqs,
                     const StepState* state, int M, int Tq, int H, int hd, int S, hipStream_t st, const int32_t* row_pos = nullptr,
                     KvPages pages = KvPages{});

// attention of every query row m = b*Tq + t (position p = state->pos + t) over keys 0..p of batch b
// (gpt.py:230-237 with the mask of generate.py:156-165).  out [M, H*hd].
// mask: fp32 [Bmask, Tc] or null; batch row b uses mask row b % Bmask.
template <typename T>
int attn_rows(const T* qbuf, T* kcache, T* vcache, T* out, float* partial_ws, const StepState* state,
              int Bp, int Tq, int H, int hd, int S, int max_pos, const float* mask, int Bmask, int Tc,
              hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,  // ev0/ev1 bracket the split-KV kernel
              const int32_t* row_pos = nullptr,    // sessions: batch row b attends keys 0..row_pos[b] (+t)
              KvPages pages = KvPages{},           // sessions with a block-granular cache
              int out_nks = 0);                    // > 0: `out` [M, H hd] is A-fragment-major (afm_index) with this many 64-byte K steps per row
size_t attn_ws_floats(int M, int H, int hd);

// embedding gathers -----------------------------------------------------------------------------
template <typename T>
int gather_rows_i32(const T* table, const int32_t* idx, T* out, int rows, int D, int n_rows, hipStream_t st, int out_nks = 0);   // gpt.py:354; out_nks > 0: out is A-fragment-major
template <typename T>
int gather_rows_i64(const T* table, const int64_t* idx, int n_idx, int null_id, T* out, int rows, int D, int n_rows, hipStream_t st);  // gpt.py:82 + generate.py:131
// cond [B,Tc,cd] fp32 (+ uncond [120,cd] for the CFG half) -> T [Bp*Tc, cd]        (generate.py:138-139)
template <typename T>
int build_text_cond(const float* cond, const T* uncond, T* out, int B, int Bp, int Tc, int cd, hipStream_t st);
// rows (b, Tq-1) of x [Bp*Tq, D] -> [Bp, D]
template <typename T>
int take_last_rows(const T* x, T* out, int Bp, int Tq, int D, hipStream_t st);
template <typename T>
int drop_last_rows(const T* x, T* out, int Bp, int Tq, int D, hipStream_t st);   // [Bp][Tq][D] -> [Bp][Tq - 1][D]
// latent fp32 [B,C] (cur) -> T [Bp, C] (duplicated for CFG)
template <typename T>
int latent_to_rows(const float* cur, T* out, int B, int Bp, int C, hipStream_t st);
// head output T [Bp, C] -> CFG combine (generate_video_diff.py:97-105) -> out_lat[b][step] fp32 and cur [B,C]
template <typename T>
int latent_head_finish(const T* y, float* cur, float* out_lat, float* trace, const StepState* state, int B, int Bp,
                       int C, int N, float cfg_scale, int cfg_interval, hipStream_t st, int b_off = 0, int B_total = 0,
                       const int32_t* row_step = nullptr);   // row_step: sessions - per-row token index instead of state->step

// fused forms for the decode step (one workgroup per row; C <= 16): latent_to_rows + adapter.fc1 + GELU, and adapter2.fc2 + finish
template <typename T>
int latent_in_fc1(const float* cur, const T* w1, T* t1, int B, int Bp, int C, int D, hipStream_t st);
template <typename T>
int latent_out_fc2(const T* t1, const T* w2, float* cur, float* out_lat, float* trace, const StepState* state, int B, int Bp, int C, int D, int N,
                   float cfg_scale, int cfg_interval, hipStream_t st, int b_off = 0, int B_total = 0, const int32_t* row_step = nullptr);

// ---- DiffLoss head pieces (diffloss.hip) ---------------------------------------------------------------------
struct DdpmCoef {   // one respaced reverse step (gaussian_diffusion.py:232-252,288-292,334-339)
  float sqrt_recip, sqrt_recipm1, coef1, coef2, min_log, max_log;
  int nonzero;      // 0 at t == 0: no noise
};
template <typename T>
int dl_make_y(const T* temb, const T* cemb, T* ys, int B, int W, hipStream_t st);
// the same for all S respaced steps: ys [S*B, W], row i*B + b
template <typename T>
int dl_make_y_all(const T* temb, const T* cemb, T* ys, int S, int B, int W, hipStream_t st);
// x_out = x_T (k < 0) or p_sample(x_in, out) (k >= 0); hc = input_proj(x_out) when hc != null.  x_in != x_out.
template <typename T>
int dl_step_proj(const T* x_in, T* x_out, const T* out, const float* noise, const StepState* state, const DdpmCoef& cf, int k, int S, int B, int C,
                 int b_off, int B_total, float temperature, uint64_t seed, const T* wip, const T* bip, T* hc, int W, hipStream_t st,
                 int n_half = 0, float cfg = 1.0f);   // n_half > 0: guidance inside the sampler (forward_with_cfg), rows [cond | uncond]
template <typename T>
int dl_ln_modulate(const T* h, const T* lnw, const T* lnb, const T* shift, const T* scale, int mod_stride, T* g, int B, int W, hipStream_t st);
template <typename T>
int dl_gated_residual(T* h, const T* gate, int gate_stride, const T* g, int B, int W, hipStream_t st);
template <typename T>
int dl_init_x(T* x, const float* noise, const StepState* state, int S, int B, int C, int b_off, int B_total, uint64_t seed, hipStream_t st);
template <typename T>
int dl_ddpm_step(T* x, const T* out, const float* noise, const StepState* state, const DdpmCoef& cf, int k, int S, int B, int C, int b_off,
                 int B_total, float temperature, uint64_t seed, hipStream_t st);
template <typename T>
int dl_finish(const T* x, float* cur, float* out_lat, float* trace, const StepState* state, int B, int C, int N, int b_off, int B_total,
              hipStream_t st);

// ---- persistent DiffLoss sampler (diffloss_persist.hip): all S reverse steps of one token in one launch ----------------------------
struct DlPersist {
  const void* ln_w[8];   // per ResBlock: in_ln weight / bias [W]
  const void* ln_b[8];
  const void* w0[8];     // mlp.0 [W, W], bias [W]
  const void* b0[8];
  const void* w2[8];     // mlp.2 [W, W], bias [W]
  const void* b2[8];
  const void* wf;        // final_layer.linear [2C, W], bias [2C]
  const void* bf;
  const void* wip;       // input_proj [W, C], bias [W]
  const void* bip;
  const void* mod_all;   // [S][B][MR] adaLN modulation vectors of every respaced step (MR = (3 depth + 2) W)
  const DdpmCoef* coef;  // [S] device array
  const float* noise;    // [N][S + 1][B_total][C] or null (Philox)
  const StepState* state;
  void* xbuf;            // exchange units {data, epoch}, dl_persist_xbuf_bytes (zeroed by the launcher)
  float* cur;            // [B][C] next transformer input
  float* out_lat;        // [B][N][C] (already offset to the lane's first sample)
  float* trace;          // optional [N][B_total][C]
  unsigned long long* prof;  // in-kernel time stamps (tools/microbench/dl_persist_lab.hip, -DVLG_DP_PROF builds only)
  int depth, W, C, S, B, MR, N, b_off, B_total;
  float temperature;
  uint64_t seed;
  unsigned* fault;       // host-visible fault word of the handle (vlg_gpt_status): set when an exchange wait runs out
  int spin_max;          // spin bound of every wait (default 1 << 20; option debug_spin_max)
  int n_half;            // 0, or B / 2: DiffLoss.sample's guidance pairs rows (b, b + B/2) as (conditional, unconditional)
  float cfg;             // ... with this scale (eps = u + cfg (c - u))
  const int32_t* row_step;   // sessions: per-row token index (device array [B]) instead of state->step; null otherwise
  int rows;              // rows per group: 0 = 4 while one workgroup per CU holds the batch, else 8; 4 / 8 = that one (option dl_persist)
};
constexpr unsigned kFaultDlPersist = 0x444C0000u;   // 'DL' | reverse step index
constexpr unsigned kFaultDecode = 0x50440000u;      // 'PD' | phase index
size_t dl_persist_xbuf_bytes(int B, int W, int esz);
template <typename T>
bool dl_persist_ok(int B, int W, int C, int depth, int rows = 0);
template <typename T>
int dl_persist(const DlPersist& p, hipStream_t st);

// ---- persistent decode step (pdecode.hip): the L transformer layers of one decode step (Tq = 1) in ONE launch -----------------------
struct PdLayer {   // device array [L]: one layer's weights (handle dtype), as nn.Linear stores them
  const void* wqkv;   // [3D, D]
  const void* wo;     // [D, D]
  const void* w13;    // [2F, D]  (w1 rows, then w3 rows)
  const void* w2;     // [D, F]
  const void* norm1;  // attention_norm.weight [D]
  const void* norm2;  // ffn_norm.weight [D]
};
struct PdArgs {
  const PdLayer* layers;
  void* x;                 // [M][D] residual stream: layer 0's input on entry, the last layer's output on exit (not normed)
  void* kc;                // KV cache [L][M][H][S][hd] (contiguous form), layer stride kv_lstride elements
  void* vc;
  size_t kv_lstride;
  const float* freqs;      // RoPE table
  const StepState* state;  // uniform position of the step
  const float* mask;       // [Bmask][Tc] or null
  int Bmask, Tc;
  void* xbuf;              // exchange granules {4 payload bytes, epoch tag}, PdXbuf layout; zero at allocation, tags only grow
  unsigned xbuf_bytes;
  unsigned* epoch;         // device word: tag base of this launch, advanced by the kernel (so a replayed graph never repeats a tag)
  unsigned* fault;         // host-visible fault word of the handle
  int spin_max;
  int L, M, D, H, hd, F, S;
  float eps;
  int kchunk;              // K elements of the w2 GEMM staged in LDS at a time (a multiple of 8 k-steps)
  int ns_max;              // upper bound of the attention KV splits
  int ksplit;              // K slices of a wo / w2 output tile (units = D / 16 * ksplit <= workgroups)
  int fm;                  // the four weight pointers of PdLayer are FRAGMENT-MAJOR copies (relayout_fragment_major): whole-line fragment loads
#ifdef VLG_PD_PROF
  unsigned long long* prof;   // tools/microbench/pd_lab.hip only: [workgroups][32] wall_clock64 stamps of layer 1
#endif
};
// granule offsets (8-byte units) of the exchange regions, both parities; shared by host and device
struct PdXbuf {
  unsigned nx, nq, ng, nap, nwp, per_parity;
  __host__ __device__ PdXbuf(int M, int D, int F, int H, int hd, int esz, int ns_max, int ksplit) {
    nx = (unsigned)M * D * esz / 4;
    nq = 3 * nx;
    ng = (unsigned)M * F * esz / 4;
    nap = (unsigned)M * H * ns_max * (hd + 2);
    nwp = (unsigned)(D / 16) * ksplit * (M > 16 ? 2 : 1) * 256;
    per_parity = 3 * nx + nq + ng + nap + 2 * nwp;
  }
  __host__ __device__ unsigned X(int par) const { return par * per_parity; }            // layer input rows (w2 output of the layer before)
  __host__ __device__ unsigned Q(int par) const { return X(par) + nx; }                 // q | k | v rows of the current position, RoPE applied
  __host__ __device__ unsigned AO(int par) const { return Q(par) + nq; }                // attention output rows
  __host__ __device__ unsigned HH(int par) const { return AO(par) + nx; }               // residual stream after attention
  __host__ __device__ unsigned G(int par) const { return HH(par) + nx; }                // SwiGLU output rows
  __host__ __device__ unsigned AP(int par) const { return G(par) + ng; }                // attention partials of the non-owner KV splits (fp32)
  __host__ __device__ unsigned WP(int par) const { return AP(par) + nap; }              // wo tile partials of the non-owner K slices (fp32) [tile][slice][mt][256]
  __host__ __device__ unsigned W2P(int par) const { return WP(par) + nwp; }             // w2 tile partials
  __host__ __device__ size_t bytes() const { return (size_t)2 * per_parity * 8; }
};
// can this shape run on the persistent kernel (otherwise: the launch chain of layers_fused)?  cus = compute units of the device
template <typename T>
bool pd_ok(int M, int D, int H, int hd, int F, int S, int cus);
size_t pd_xbuf_bytes(int M, int D, int H, int hd, int F, int esz);
template <typename T>
int pd_layers(PdArgs a, hipStream_t st);

int advance_state(StepState* state, hipStream_t st);
int set_state(StepState* state, int pos, int step, hipStream_t st);
// teacher forcing: cur_tok [Bp] / cur_lat [B][C] <- ids [B_total][N] / lat [B_total][N][C] at token index state->step (rows b_off..)
int force_next_input(const StepState* state, const int32_t* ids, const float* lat, int32_t* cur_tok, float* cur_lat, int B, int Bp, int C, int N,
                     int b_off, hipStream_t st);
template <typename T>
int gather_session_rows(const T* cls_table, int n_cls, const T* tok_table, int n_tok, const int32_t* row_cls, const int32_t* cur_tok,
                        const T* pending, T* out, int rows, int D, hipStream_t st, int out_nks = 0);   // row_cls: >= 0 class id, -3 pending row, else token; out_nks > 0: out is A-fragment-major
template <typename T>
int override_session_rows(const int32_t* row_cls, const T* pending, T* out, int rows, int D, hipStream_t st, int out_nks = 0);   // rows with row_cls = -3 take their pending row

// sampler (sampler.hip) -----------------------------------------------------------------------
// logits fp32 [Bp, V]; writes out_ids[b*N + step] (if out_ids), cur_tok[b] (and [b+B] when cfg_on),
// trace[step][b][V] (if trace), probs[b][V] (if probs).  noise: fp32 [N or 1][B][V] indexed by step.
int sample_rows(const float* logits, int B, int V, bool cfg_on, const vlg_sampling_params& sp, const float* noise,
                const StepState* state, int fixed_step, int N, int32_t* out_ids, int32_t* cur_tok, float* trace,
                float* probs, hipStream_t st, int b_off = 0, int B_total = 0, const int32_t* row_step = nullptr);
// b_off / B_total: the rows are samples b_off.. of a B_total-sample call (batch lanes): noise, trace and the Philox
// counter are indexed by the global sample id so results do not depend on the lane split.

}  // namespace vlg
