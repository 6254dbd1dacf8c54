/* libvlg - MI355X (gfx950) native KV-cached visual-token sampling + VQ / CausalVideoVAE decode.
 *
 * C-ABI drop-in boundary for ONE hot path of BinZhu-ece/Video-LlamaGen.  The reference has no
 * FFI for this path (it sits behind Python callables); each entry point below names
 * the reference callable it replaces (paths relative to the reference root).  No torch types cross
 * this boundary: plain pointers, sizes, opaque handles, int status codes.
 *
 * Conventions
 *  - every function returns VLG_OK (0) or a negative vlg_status; vlg_last_error() gives the text.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is enqueued
 *    asynchronously on it; the caller synchronises.  vlg_gpt_generate runs on handle-owned streams forked from
 *    and joined back into `stream` with events: it returns once the work is enqueued, and anything the caller
 *    enqueues on `stream` afterwards is ordered behind it.  (Exceptions, documented at the function: the
 *    unit entry points and the session calls, which wait for their step.)
 *  - pointers named d_* are DEVICE pointers owned by the caller; weights / KV caches / workspaces are
 *    owned by the handle.  One handle is used by one host thread at a time (the reference's model
 *    object owns its caches and is not re-entrant either, gpt.py:318-332).
 *  - dtype codes: VLG_F32 = 0, VLG_BF16 = 1.
 */
#ifndef VLG_H
#define VLG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  VLG_OK = 0,
  VLG_ERR_BAD_ARG = -1,      /* NULL pointer / negative size / unknown name          */
  VLG_ERR_BAD_SHAPE = -2,    /* tensor shape does not match the configuration        */
  VLG_ERR_UNSUPPORTED = -3,  /* head_dim / dtype / model_type not supported          */
  VLG_ERR_OOM = -4,          /* hipMalloc failed                                     */
  VLG_ERR_HIP = -5,          /* any other HIP runtime error                          */
  VLG_ERR_STATE = -6         /* weights missing / call order violated                */
} vlg_status;

enum { VLG_F32 = 0, VLG_BF16 = 1 };
enum { VLG_C2I = 0, VLG_T2I = 1, VLG_T2V = 2 };
enum { VLG_HEAD_LOGITS = 0, VLG_HEAD_ADAPTER2 = 1, VLG_HEAD_HIDDEN = 2 };

const char* vlg_last_error(void);
/* 3 ints: major, minor, gfx arch (950) */
void vlg_version(int* out3);

/* ------------------------------------------------------------------------------------------
 * GPT (Llama-style decoder)     replaces autoregressive/models/gpt.py:262-371 `Transformer`,
 *                               gpt_video.py:270-431 (t2v, adapter2 head)
 * ------------------------------------------------------------------------------------------ */
typedef struct vlg_gpt vlg_gpt_t;

typedef struct {
  int32_t dim, n_layer, n_head;      /* ModelArgs, gpt.py:23-50                              */
  int32_t vocab_size, block_size, cls_token_num;
  int32_t model_type;                /* VLG_C2I | VLG_T2I | VLG_T2V                          */
  int32_t num_classes, caption_dim;
  int32_t vae_embed_dim, num_frames, t_downsample_size; /* gpt_video.py:58-60                */
  int32_t head;                      /* VLG_HEAD_*                                           */
  int32_t dtype;                     /* compute/storage dtype of weights, activations, KV    */
  int32_t multiple_of;               /* SwiGLU hidden rounding, gpt.py:29,154-159            */
  float norm_eps, rope_base;
  /* VLG_HEAD_HIDDEN only: per-token diffusion head, gpt_video_diff.py:76-78 (DiffLoss)       */
  int32_t diffloss_w, diffloss_d, num_sampling_steps;
} vlg_gpt_config;

int vlg_gpt_create(const vlg_gpt_config* cfg, vlg_gpt_t** out);
int vlg_gpt_destroy(vlg_gpt_t* h);
/* State-dict entry by its reference name ("layers.3.attention.wqkv.weight", ...).
 * `data` is fp32 or bf16 (src_dtype), host (src_on_device=0) or device memory; it is converted to the
 * handle dtype and copied; a device source is read after a device-wide synchronise, so work still
 * pending on any caller stream that produces it is waited for.  Unknown names ("freqs_cis", training-only tensors) return VLG_OK with
 * *consumed = 0 (strict=False semantics of sample_t2i.py:71).                                   */
int vlg_gpt_load_tensor(vlg_gpt_t* h, const char* name, const void* data, const int64_t* shape,
                        int32_t ndim, int32_t src_dtype, int32_t src_on_device, int32_t* consumed);

typedef struct {
  float cfg_scale;       /* generate.py:128; CFG doubles the batch when > 1                    */
  int32_t cfg_interval;  /* generate.py:113-114 (Q7)                                           */
  float temperature;     /* generate.py:58                                                     */
  int32_t top_k;         /* generate.py:32-36; 0 = off                                         */
  float top_p;           /* generate.py:38-53; 1.0 = off                                       */
  int32_t sample_logits; /* 1 = multinomial, 0 = greedy argmax (generate.py:62-65)             */
  uint64_t seed;         /* Philox seed for the on-device Exp(1) noise when d_noise == NULL    */
} vlg_sampling_params;

/* replaces generate() autoregressive/models/generate.py:127-180 (c2i/t2i) and generate_video_diff.py:185-228 (t2v) with
 * the gpt_video.py:431 adapter2 head or the DiffLoss.sample head (diffloss.py:35-52; cfg_scale must be 1 as in the
 * reference's shipped scripts; sp->temperature scales the reverse-step noise).
 *   d_cond      c2i: int64 [B] class ids; t2i/t2v: fp32 [B, cls_token_num, caption_dim] (already * mask)
 *   d_emb_mask  fp32 [B, cls_token_num] (1 = valid, left-padded) or NULL
 *   d_noise     logits head: fp32 [N, B, vocab] Exp(1) draws consumed as argmax(p / q); hidden (DiffLoss) head: fp32
 *               [N, num_sampling_steps + 1, B, C] N(0,1) draws (x_T, then one per reverse step); NULL -> Philox(seed)
 *   d_out_ids   int32 [B, N]            (logits head)
 *   d_out_lat   fp32  [B, N, vae_embed_dim]  (adapter2 / hidden head)
 *   d_trace     optional fp32 [N, B, vocab|C]: the CFG-combined head output fed to the sampler
 * Stream contract: the call returns with the work enqueued behind `stream` and its results ordered on `stream`.  Results are staged in a
 * handle-owned buffer and copied out on `stream` at the end: issue the calls of ONE handle from one stream (or synchronise between
 * streams), as with any torch module - a second call from another stream could overwrite the staging buffer under the first copy. */
int vlg_gpt_generate(vlg_gpt_t* h, const void* d_cond, const float* d_emb_mask, int32_t B, int32_t N,
                     const vlg_sampling_params* sp, const float* d_noise, int32_t* d_out_ids,
                     float* d_out_lat, float* d_trace, void* stream);

/* Iteration-level batching for the request front-end (token models, and the continuous-latent video models below; the role vLLM's model runner plays
 * in the reference's autoregressive/serve/, model_runner.py:845-886 + sampler.py:46-125).  A session owns `rows` KV-cache slots
 * of max_new_tokens + 1 positions; with sp->cfg_scale > 1 every slot carries its unconditional partner internally.
 *   session_step  advances EVERY slot by one token, each slot at its own position.  h_row_class[rows] (host memory):
 *                 >= 0  start a request with this class id in the slot (its first token is sampled by this step),
 *                 -3    (text-conditioned models) start the request whose condition vlg_gpt_session_prefill put into the slot,
 *                 -1    continue the slot's request,   -2  leave the slot idle.
 *                 The call returns once the iteration is enqueued on the session's stream (its inputs are staged in pinned memory); tokens
 *                 stay on the device.  session_read / session_read_latents / session_end wait for the stream; prefill, reserve and release are
 *                 ordered behind earlier iterations by that stream.
 *   session_read  copies the first n_tokens tokens of a slot to host memory (call it when the request is done, before
 *                 the slot is reused).                                                                               */
int vlg_gpt_session_begin(vlg_gpt_t* h, int32_t rows, int32_t max_new_tokens, const vlg_sampling_params* sp);
/* Text-conditioned models (t2i, t2v): puts ONE request's condition into a slot (ending whatever ran there) - d_cond fp32 [cls_token_num, caption_dim]
 * (already * mask, sample_t2i.py:105-119), d_mask fp32 [cls_token_num] (1 = valid, left-padded) or NULL.  Positions 0 .. T-2 are
 * prefilled into the slot's KV rows (and uncond_embedding into its guidance partner's); the request then starts with code -3 in
 * session_step, whose first iteration runs the last condition token at position T-1 and samples token 0.  Waits for the prefill. */
int vlg_gpt_session_prefill(vlg_gpt_t* h, int32_t slot, const float* d_cond, const float* d_mask);
/* The same for n requests that land in CONSECUTIVE slots first_slot .. first_slot + n - 1 (d_cond [n, cls_token_num, caption_dim], d_mask
 * [n, cls_token_num] or NULL): one prefill of n x (cls_token_num - 1) rows instead of n one-request prefills. */
int vlg_gpt_session_prefill_batch(vlg_gpt_t* h, int32_t first_slot, int32_t n, const float* d_cond, const float* d_mask);
int vlg_gpt_session_step(vlg_gpt_t* h, const int32_t* h_row_class);
/* Block-granular KV cache (the role of vLLM's block manager behind serve/gpt_model.py:181-224).  With option "kv_block" = BS > 0
 * (a power of two, 8..1024; set before session_begin) the session's cache is a pool of "kv_pool_blocks" blocks of BS positions
 * (0 = enough for every slot at full length; block 0 is a scratch block for idle rows) instead of one slot of cls_token_num +
 * max_new_tokens positions per row, so a slot holds memory for ITS request's length only.
 *   session_reserve     gives `slot` (and its guidance partner) blocks for cls_token_num + n_tokens positions; call it before the
 *                       request's prefill / start.  VLG_ERR_OOM (nothing changed) when the pool cannot cover it: keep the request
 *                       queued and retry after a release.  Stepping a slot beyond its reservation is VLG_ERR_STATE.
 *   session_release     returns the slot's blocks to the pool and idles the slot (after session_read).
 *   session_free_blocks blocks currently free (-1 when the session has contiguous slots) and the block size.
 * Without "kv_block" reserve / release succeed and change nothing.                                                              */
int vlg_gpt_session_reserve(vlg_gpt_t* h, int32_t slot, int32_t n_tokens);
int vlg_gpt_session_release(vlg_gpt_t* h, int32_t slot);
int vlg_gpt_session_free_blocks(vlg_gpt_t* h, int32_t* n_free, int32_t* block_size);
int vlg_gpt_session_read(vlg_gpt_t* h, int32_t row, int32_t n_tokens, int32_t* h_out);
/* Sessions of the continuous-latent video models (model_type t2v, heads adapter2 and hidden/DiffLoss; beyond the reference, whose serving
 * path stops at class-conditional images): the same calls - session_prefill puts a request's caption features into a slot, code -3
 * starts it, -1 continues - with cfg_scale 1 (and cfg_iter 1): every slot produces ONE latent token [vae_embed_dim] per iteration at its
 * own position (3-D RoPE row, KV rows, DiffLoss noise stream keyed by (seed, slot, token index)).  The hidden head needs the persistent
 * DiffLoss sampler (option dl_persist, `rows` within what it covers).  session_read_latents copies the first n_tokens latents of a slot,
 * fp32 [n_tokens, vae_embed_dim], to host memory.                                                                                      */
int vlg_gpt_session_read_latents(vlg_gpt_t* h, int32_t row, int32_t n_tokens, float* h_out);
int vlg_gpt_session_end(vlg_gpt_t* h);

/* bytes the last generate() call moved algorithmically (weights + KV read/write + logits), for roofline */
int vlg_gpt_last_algorithmic_bytes(vlg_gpt_t* h, double* weight_bytes, double* kv_bytes, double* other_bytes);
/* options: "graph" (default 1) = HIP-graph replay of the decode step; "time_attn" (default 0) = eager decode loop
 * with HIP events (on the stream the kernels run on) around layer 0's split-KV attention kernel of every step.
 * Kernel-selection switches, results unchanged up to fp32 summation order: "fuse_gemm" (1: fused decode
 * GEMMs and fused DiffLoss / latent heads), "fuse_swiglu" (1), "dl_persist" (1: DiffLoss sampler as one persistent launch per
 * token - workgroup groups of 4 rows up to 32 rows at width 1024, of 8 rows up to 64; 4 / 8: that group height, the per-step launch chain when it does not fit; 0: the chain), "pdecode" (1: all transformer layers of a decode step as one persistent launch where the shape allows, small row counts; "pd_rows" = row
 * cap replacing the measured rule), "weights_fm" (1) / "act_fm" (1): the decode GEMMs read fragment-major copies of
 * the Linear weights (one MFMA B fragment = 1 KB contiguous, built at the first call after a load; twice the weight memory for those
 * tensors) and the fused chain keeps its activations A-fragment-major - whole-cache-line requests, bit-identical results; "debug_pos_offset" (0; benchmarks: decode starts `offset` positions after the condition, over zeroed
 * cache rows - the cost of a late-context step without generating up to it), "kv_block" / "kv_pool_blocks" (sessions, see above).  Unknown keys return VLG_ERR_BAD_ARG.  (The measured-slower variants
 * of rounds 1-2 - batch lanes, fuse_qkv, attn_inlaunch, splitk_inlaunch, gemm_lds - were removed in round 3; DESIGN.md section 5
 * keeps their measurements.)                                                                                         */
int vlg_gpt_set_option(vlg_gpt_t* h, const char* key, int64_t value);
/* real-valued options.  "cfg_iter" (default 1.0; hidden / DiffLoss head only): the `cfg` argument of DiffLoss.sample
 * (autoregressive/models/diffloss.py:35-41 -> SimpleMLPAdaLN.forward_with_cfg :240-248, passed as cfg_iter by generate_video_diff.py:89-91):
 * != 1 treats rows [0, B/2) of the batch as the conditional and rows [B/2, B) as the unconditional half of every pair (b, b + B/2): one
 * x_T draw per pair, the network sees the conditional half's x_t twice, eps = u + cfg (c - u); B must be even.                        */
int vlg_gpt_set_option_f64(vlg_gpt_t* h, const char* key, double value);
/* Teacher forcing for evaluation (per-step outputs at positions far into a sequence, without the chaos of a free-running trajectory): while
 * set, the input of decode step i + 1 is forced[b][i] - int32 ids [B][N] for the token heads, fp32 latents [B][N][C] for the latent heads;
 * device memory, caller-owned, alive until cleared - instead of what the head produced at step i; d_out_* and d_trace of vlg_gpt_generate
 * still receive the model's OWN output of every step.  This is the inference-side form of the reference's teacher-forced forward
 * (autoregressive/models/gpt.py:334-347: `idx[:, :-1]` as inputs, targets beside it; gpt_video.py:404-431 for latents).  Both null: off.  */
int vlg_gpt_set_teacher(vlg_gpt_t* h, const int32_t* d_forced_ids, const float* d_forced_latents);
/* Device-side faults.  The persistent kernels of this handle (the DiffLoss sampler, the persistent decode step) exchange data between
 * workgroups inside one launch and bound every wait; a wait that runs out (the grid was not fully resident because something else held
 * compute units, or option "debug_spin_max" forced it) records a fault word in host-visible memory and the kernel drains.  Results of
 * that call are invalid (latents are NaN-poisoned).  vlg_gpt_status reports and clears the fault: VLG_OK, or VLG_ERR_STATE with the
 * fault's kernel / phase in vlg_last_error().  sync != 0 first waits for the handle's stream (vlg_gpt_generate itself returns with the
 * work enqueued); vlg_gpt_generate and the session calls also return VLG_ERR_STATE on entry while an uncollected fault is pending.
 * Option "debug_spin_max" (default 0 = the built-in bound of ~1 s): spin bound of every in-launch wait; 1 makes the first wait that is
 * not satisfied at once give up - how the tests inject a time-out without oversubscribing the chip.
 * EXCLUSIVITY: a persistent launch wants one workgroup on every compute unit of the device.  One process per GPU with one generate() in
 * flight (the deployment this library is built for) always satisfies that; inside one process the decode loops of different handles
 * and host threads are chained on the device (each waits for the previous one's last step), so they cannot starve each other either.  Two PROCESSES that share one GPU and both run persistent
 * launches can split the compute units between them; both then time out (observed with two benchmark ranks on one card: VLG_ERR_STATE on
 * the first decode step, never wrong results).  When a GPU is shared, set options "pdecode" = 0 and "dl_persist" = 0: the per-layer launch
 * chains need no residency.                                                                                                        */
int vlg_gpt_status(vlg_gpt_t* h, int32_t sync);
/* number of decode-step graphs this handle has instantiated so far: vlg_gpt_generate keeps the instantiated graph of its last
 * call and replays it while shape, sampling parameters, options and buffer addresses are unchanged                           */
int vlg_gpt_graphs_built(vlg_gpt_t* h, int64_t* count);
/* host-side counters of this handle (tests prove which decode path a call took): "pd_steps" = decode steps recorded (launched eagerly or
 * captured into a graph) as ONE persistent launch over all layers (csrc/pdecode.hip; option "pdecode", default 1, small row counts);
 * "chain_steps" = decode steps recorded as the per-layer launch chain; "graphs_built".  Unknown keys return VLG_ERR_BAD_ARG.        */
int vlg_gpt_counter(vlg_gpt_t* h, const char* key, int64_t* count);
/* event-timed attention launches of the last generate() with time_attn=1: total ms, total algorithmic KV bytes
 * (2 * Bp * D * (p+1) * elem per launch), number of launches                                                      */
int vlg_gpt_attn_timing(vlg_gpt_t* h, double* ms_sum, double* bytes_sum, int64_t* launches);
/* mean elapsed time (ms) of an EMPTY event pair recorded back to back on the same stream right after that run: what the
 * bracket itself adds to every timed launch                                                                        */
int vlg_gpt_attn_event_overhead(vlg_gpt_t* h, double* ms_per_pair);

/* ------------------------------------------------------------------------------------------
 * Unit entry points (parity tests call the very kernels the handle uses)
 * ------------------------------------------------------------------------------------------ */
/* RMSNorm gpt.py:137-148.  x,out [rows, dim] in `dtype`, w [dim] in `dtype` */
int vlg_rmsnorm(const void* d_x, const void* d_w, void* d_out, int32_t rows, int32_t dim, float eps,
                int32_t dtype, void* stream);
/* bias-free Linear gpt.py:199-200,161-163: out[M,N] = x[M,K] @ w[N,K]^T (fp32 accumulate, output `dtype`) */
int vlg_linear(const void* d_x, const void* d_w, void* d_out, int32_t M, int32_t N, int32_t K,
               int32_t dtype, void* stream);
/* RoPE table gpt.py:407-420 / gpt_video.py:532-552 -> host fp32 [cls + vae_t*grid^2, hd/2, 2] */
int vlg_rope_table(int32_t grid, int32_t vae_t, int32_t head_dim, float base, int32_t cls_token_num,
                   float* host_out);
/* sampler generate.py:16-66 + CFG combine :81-82.  d_logits fp32 [Bp, V] (Bp = 2B if cfg_on),
 * d_noise fp32 [B,V] or NULL, outputs int32 [B] and optional probs fp32 [B,V]                 */
int vlg_sample(const float* d_logits, int32_t B, int32_t V, int32_t cfg_on, const vlg_sampling_params* sp,
               const float* d_noise, uint64_t step, int32_t* d_out_idx, float* d_out_probs, void* stream);
/* single-query attention over a KV cache, gpt.py:226-237 (decode step).
 * q [Bp,H,hd], k/v cache [Bp,H,S,hd] all `dtype`; attends keys 0..pos for every row;
 * d_mask fp32 [Bmask, Tc] or NULL (text padding, generate.py:156-165), out [Bp, H*hd]          */
int vlg_attn_decode(const void* d_q, const void* d_k, const void* d_v, void* d_out, int32_t Bp, int32_t H,
                    int32_t S, int32_t hd, int32_t pos, const float* d_mask, int32_t Bmask, int32_t Tc,
                    int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * VQ-16 / VQ-8 image tokenizer   replaces tokenizer/tokenizer_image/vq_model.py
 * ------------------------------------------------------------------------------------------ */
typedef struct vlg_vq vlg_vq_t;
typedef struct {
  int32_t codebook_size, codebook_embed_dim, z_channels, ch; /* vq_model.py:13-24,129            */
  int32_t n_mult;
  int32_t ch_mult[8];                                        /* decoder_ch_mult                  */
  int32_t num_res_blocks;
  int32_t l2_norm;                                           /* codebook_l2_norm                 */
  int32_t dtype;                                             /* activation/weight compute dtype  */
} vlg_vq_config;
int vlg_vq_create(const vlg_vq_config* cfg, vlg_vq_t** out);
int vlg_vq_destroy(vlg_vq_t* h);
int vlg_vq_load_tensor(vlg_vq_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim,
                       int32_t src_dtype, int32_t src_on_device, int32_t* consumed);
/* VQModel.decode_code vq_model.py:52-55: d_codes int32 [B, gh*gw] -> d_out fp32 [B,3,16gh,16gw] (NCHW) */
int vlg_vq_decode_code(vlg_vq_t* h, const int32_t* d_codes, int32_t B, int32_t gh, int32_t gw, float* d_out,
                       void* stream);
/* VectorQuantizer.forward indices vq_model.py:215-233: d_z fp32 [B,C,H,W] -> int32 [B*H*W]            */
int vlg_vq_argmin(vlg_vq_t* h, const float* d_z, int32_t B, int32_t Hh, int32_t Ww, int32_t* d_idx, void* stream);
/* VQModel.encode vq_model.py:41-45 (Encoder :64-124 -> quant_conv -> VectorQuantizer argmin): d_x fp32 [B,3,H,W] -> int32
 * [B*(H/16)*(W/16)] indices; d_z (optional) receives the pre-quantisation latents fp32 [B, e_dim, H/16, W/16]            */
int vlg_vq_encode(vlg_vq_t* h, const float* d_x, int32_t B, int32_t Hh, int32_t Ww, int32_t* d_idx, float* d_z, void* stream);
/* Codebook.forward argmin, tokenizer_video/vqvae.py:161-170 (== CausalVideoVAE quant.py:42-54):
 * d_z fp32 [n, dim] rows, d_codebook fp32 [n_codes, dim] -> int32 [n] (no normalisation)              */
int vlg_codebook_argmin(const float* d_z, const float* d_codebook, int32_t n, int32_t n_codes, int32_t dim,
                        int32_t* d_idx, void* stream);
/* (vlg_codebook_argmin / vlg_codebook_forward keep their code norms, usage histogram and partial sums in scratch buffers owned by the
 *  library, one set per stream: calls on one stream are ordered by it, calls on different streams or from different threads are
 *  independent.)
 * Codebook.forward in eval mode, tokenizer_video/vqvae.py:161-209 (== CausalVideoVAE quant.py:42-96):
 * d_z fp32 [B, dim, n_pos] (the reference's [b, c, t, h, w] with n_pos = t*h*w), d_codebook fp32 [n_codes, dim] ->
 *   d_encodings        int32 [B, n_pos]        nearest code per position (first minimum)
 *   d_embeddings_st    fp32  [B, dim, n_pos]   (E[idx] - z) + z, the straight-through output        (optional, with the next)
 *   d_loss_perplexity  fp32  [2]               {0.25 * mse(z, E[idx]), exp(-sum p log(p + 1e-10))}                          */
int vlg_codebook_forward(const float* d_z, const float* d_codebook, int32_t B, int32_t dim, int64_t n_pos, int32_t n_codes,
                         int32_t* d_encodings, float* d_embeddings_st, float* d_loss_perplexity, void* stream);

/* ------------------------------------------------------------------------------------------
 * tokenizer_video VQ-VAE decode   replaces VQVAE.decode tokenizer/tokenizer_video/vqvae.py:48-51 (+ Decoder :245-272,
 *                                 AttentionResidualBlock/AxialBlock :89-125, SamePadConv(Transpose)3d :276-319)
 * ------------------------------------------------------------------------------------------ */
typedef struct vlg_vqvae vlg_vqvae_t;
typedef struct {
  int32_t n_hiddens, embedding_dim, n_codes, n_res_layers, n_head; /* vqvae.py:78-86 (240, 256, 2048, 4; AxialBlock n_head 2) */
  int32_t n_upsample;                                              /* transposed 4^3 stride-2 convs (downsample (4,4,4) -> 2)    */
  int32_t dtype;
} vlg_vqvae_config;
int vlg_vqvae_create(const vlg_vqvae_config* cfg, vlg_vqvae_t** out);
int vlg_vqvae_destroy(vlg_vqvae_t* h);
int vlg_vqvae_load_tensor(vlg_vqvae_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim,
                          int32_t src_dtype, int32_t src_on_device, int32_t* consumed);
/* d_codes int32 [B, t, h, w] -> d_out fp32 [B, 3, t*2^n_up, h*2^n_up, w*2^n_up] */
int vlg_vqvae_decode(vlg_vqvae_t* h, const int32_t* d_codes, int32_t B, int32_t t, int32_t hh, int32_t ww, float* d_out,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * CausalVideoVAE decoder   replaces CausalVAEModel.decode modeling_causalvae.py:394-404
 * ------------------------------------------------------------------------------------------ */
typedef struct vlg_vae vlg_vae_t;
typedef struct {
  int32_t hidden_size, z_channels, embed_dim, num_res_blocks; /* modeling_causalvae.py:268-320   */
  int32_t n_mult;
  int32_t hidden_size_mult[8];
  int32_t spatial_upsample[8];   /* per level (index = i_level) 0/1                              */
  int32_t temporal_upsample[8];
  int32_t dtype;
} vlg_vae_config;
int vlg_vae_create(const vlg_vae_config* cfg, vlg_vae_t** out);
int vlg_vae_destroy(vlg_vae_t* h);
int vlg_vae_load_tensor(vlg_vae_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim,
                        int32_t src_dtype, int32_t src_on_device, int32_t* consumed);
/* d_z fp32 [B, embed_dim, t, h, w] (NCTHW) -> d_out fp32 [B, 3, T, 8h, 8w], T = 4(t-1)+1 for the
 * default config.  Tiling never triggers at the benchmark shape.                        */
int vlg_vae_decode(vlg_vae_t* h, const float* d_z, int32_t B, int32_t t, int32_t hh, int32_t ww, float* d_out,
                   void* stream);
int vlg_vae_out_shape(vlg_vae_t* h, int32_t t, int32_t hh, int32_t ww, int32_t* T, int32_t* H, int32_t* W);
/* CausalVAEModel.encode up to the posterior parameters, modeling_causalvae.py:382-392 (Encoder :26-148 -> quant_conv):
 * d_x fp32 [B,3,T,H,W] -> d_moments fp32 [B, 2*embed_dim, (T-1)/4+1, H/8, W/8] = [mean | logvar]                      */
int vlg_vae_encode(vlg_vae_t* h, const float* d_x, int32_t B, int32_t T_, int32_t Hh, int32_t Ww, float* d_moments, void* stream);
/* Spatial tile compositing of the tiled passes: CausalVAEModel.tiled_decode2d / tiled_encode2d with blend_v / blend_h
 * (modeling_causalvae.py:424-443,491-570).  Tiles [planes, th, tw] fp32 (planes = B*C*T) are visited in raster order; the call
 * cross-fades d_tile IN PLACE over its first min(above_h, th, extent) rows against the last rows of the finished tile above
 * (d_above [planes, above_h, tw] or NULL) and then over its first min(left_w, tw, extent) columns against the last columns of the
 * finished tile to its left (d_left [planes, th, left_w] or NULL), and copies rows < keep_h, columns < keep_w of the result to
 * d_canvas [planes, canvas_h, canvas_w] at (y0, x0).                                                                        */
/* Unit entry points of the decoder kernels (per-op parity): planar fp32 tensors in the reference's layouts,
 * computed by the handle kernels in `dtype`.  They synchronise the stream and use process-wide scratch: tests and tools only.
 *   vlg_causal_conv3d    CausalConv3d (modules/conv.py:76-130): x [B,Cin,T,H,W], w [Cout,Cin,kt,kh,kw], bias [Cout] or NULL;
 *                        replicate-first-frame time pad, "same" zero pad in H/W; stride_hw 2 = SpatialDownsample2x's conv (zero pad
 *                        (0,1) bottom/right, updownsample.py:63-93); nearest_up 1 = SpatialUpsample2x (nearest x2 on H,W first, :124-153)
 *   vlg_group_norm       Normalize (GroupNorm 32 groups, normalize.py:14-17) [+ swish, ops.py:14-15]: x [B,C,P]
 *   vlg_time_upsample2x  TimeUpsample2x (updownsample.py:182-194): x [B,C,T,HW] -> [B,C,2T-1,HW]                                     */
int vlg_causal_conv3d(const float* d_x, const float* d_w, const float* d_bias, int32_t B, int32_t Cin, int32_t T_, int32_t H,
                      int32_t W, int32_t Cout, int32_t kt, int32_t kh, int32_t kw, int32_t stride_hw, int32_t nearest_up,
                      int32_t dtype, float* d_out, void* stream);
int vlg_group_norm(const float* d_x, const float* d_gamma, const float* d_beta, int32_t B, int32_t C, int64_t P, float eps,
                   int32_t swish, int32_t dtype, float* d_out, void* stream);
int vlg_time_upsample2x(const float* d_x, int32_t B, int32_t C, int32_t T_, int64_t HW, int32_t dtype, float* d_out, void* stream);
/* Measurement hook (bench.py's MFMA roofline entry): with enable = 1 every halo-tile implicit-GEMM convolution launch of the
 * decoders (conv_halo_kernel, 97 % of the CausalVideoVAE decoder's FLOPs) is bracketed by HIP events on the stream it runs on;
 * read() waits for them and returns total ms, total FLOPs (2 * outputs * taps * Cin * Cout) and the launch count since enable.  */
int vlg_conv_timing(int32_t enable);
int vlg_conv_timing_read(double* ms_sum, double* flop_sum, int64_t* launches);
int vlg_tile_blend(float* d_tile, const float* d_above, const float* d_left, int64_t planes, int32_t th, int32_t tw,
                   int32_t above_h, int32_t left_w, int32_t extent, float* d_canvas, int32_t canvas_h, int32_t canvas_w,
                   int32_t y0, int32_t x0, int32_t keep_h, int32_t keep_w, void* stream);

/* ---- T5 text encoder: the conditioning step in front of t2i / t2v (language/t5.py:60-81 -> transformers.T5EncoderModel) ------
 * State-dict names are transformers' ("shared.weight", "encoder.block.{i}.layer.0.SelfAttention.{q,k,v,o}.weight",
 * "...layer.0.SelfAttention.relative_attention_bias.weight" (block 0), "...layer.{0,1}.layer_norm.weight",
 * "...layer.1.DenseReluDense.{wi_0,wi_1,wo}.weight", "encoder.final_layer_norm.weight").
 * encode: input_ids int64 [B,T], attention_mask fp32 [B,T] (1 = token, language/t5.py:66-74) -> last_hidden_state fp32 [B,T,d_model] */
typedef struct vlg_t5 vlg_t5_t;
typedef struct {
  int32_t d_model, d_kv, num_heads, d_ff, num_layers, vocab_size;
  int32_t relative_attention_num_buckets, relative_attention_max_distance;
  int32_t gated_gelu;            /* 1: feed_forward_proj = "gated-gelu" (flan-t5, t5-v1_1); the only form built */
  int32_t dtype;
  float layer_norm_epsilon;
} vlg_t5_config;
int vlg_t5_create(const vlg_t5_config* cfg, vlg_t5_t** out);
int vlg_t5_destroy(vlg_t5_t* h);
int vlg_t5_load_tensor(vlg_t5_t* h, const char* name, const void* data, const int64_t* shape, int32_t ndim,
                       int32_t src_dtype, int32_t src_on_device, int32_t* consumed);
int vlg_t5_encode(vlg_t5_t* h, const int64_t* d_input_ids, const float* d_attention_mask, int32_t B, int32_t T,
                  float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLG_H */
