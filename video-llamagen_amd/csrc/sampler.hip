// Token sampler: one 1024-thread workgroup per batch row, the whole vocabulary (<= 16384) in registers / LDS.
//
// Reference: autoregressive/models/generate.py:16-66 (top_k_top_p_filtering, sample) and the CFG combine of
// :81-82 / :96-99.  Order: CFG -> / temperature -> top-k (ties kept, threshold = k-th largest value)
// -> top-p (descending sort, softmax, cumsum, shift-right, scatter back) -> softmax -> multinomial / argmax.
// torch.multinomial(probs, 1) == argmax(probs / q), q ~ Exp(1) (verified in SURVEY.md §7): q is an input
// tensor (parity tests) or a Philox4x32-10 stream (bench).
//
// top-k uses a 4-pass 8-bit radix select on order-preserving keys (exact threshold, no sort);
// top-p runs a bitonic sort of (key, index) pairs in 128 KiB of the CU's 160 KiB LDS.
#include "gpt_kernels.h"

namespace vlg {

namespace {

constexpr int NT = 1024;   // threads per row
constexpr int PER = 16;    // vocabulary entries per thread
constexpr int VMAX = NT * PER;

__device__ __forceinline__ uint32_t fkey(float x) {  // larger float -> larger key
  uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float exp1_noise(uint64_t seed, uint32_t step, uint32_t b, uint32_t v) {
  uint32_t c[4] = {v, b, step, 0x5eedu};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  return -logf(u);
}

struct Smem {
  float redf[16];
  int redi[16];
  double redd[16];
  uint32_t hist[256];
  uint32_t bcast[4];
};

__device__ __forceinline__ float block_max(float v, Smem& sm) {
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm.redf[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sm.redf[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) r = fmaxf(r, sm.redf[i]);
  return r;
}
__device__ __forceinline__ float block_sumf(float v, Smem& sm) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm.redf[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += sm.redf[i];
  return r;
}

template <bool TOPP>
__global__ __launch_bounds__(NT) void sample_kernel(const float* __restrict__ logits, int B, int V, int cfg_on, float cfg_scale,
                                                    int cfg_interval, float temperature, int top_k, float top_p,
                                                    int sample_logits, uint64_t seed, const float* __restrict__ noise,
                                                    const StepState* __restrict__ state, int fixed_step, int N,
                                                    int32_t* __restrict__ out_ids, int32_t* __restrict__ cur_tok,
                                                    float* __restrict__ trace, float* __restrict__ probs_out, int b_off, int B_total,
                                                    const int32_t* __restrict__ row_step) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  __shared__ Smem sm;
  const int b = blockIdx.x, t = threadIdx.x;
  const int step = row_step ? row_step[b] : (state ? state->step : fixed_step);

  // ---- 1. load, CFG combine, temperature ------------------------------------------------------------
  const bool cfg_flag = cfg_on && !(cfg_interval > -1 && (step - 1) > cfg_interval);
  const float tdiv = fmaxf(temperature, 1e-5f);
  float x[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int v = t + i * NT;
    float val = -INFINITY;
    if (v < V) {
      val = logits[(size_t)b * V + v];
      if (cfg_flag) {
        const float u = logits[(size_t)(b + B) * V + v];
        val = __fadd_rn(u, __fmul_rn(__fsub_rn(val, u), cfg_scale));   // generate.py:82
      }
      if (trace) trace[((size_t)step * B_total + b_off + b) * V + v] = val;
      val = __fdiv_rn(val, tdiv);                                        // generate.py:58
    }
    x[i] = val;
  }

  // ---- 2. top-k: exact k-th largest by radix select; ties kept (generate.py:35) -----------------------
  if (top_k > 0) {
    int k = top_k < 1 ? 1 : top_k;
    if (k > V) k = V;
    if (k < V) {
      uint32_t prefix = 0, pmask = 0;
      int kk = k;
      for (int shift = 24; shift >= 0; shift -= 8) {
        if (t < 256) sm.hist[t] = 0;
        __syncthreads();
        // Histogram of the digit at `shift` over the keys that still match the prefix.  In the first passes nearly all keys of a wave share
        // one digit (sign + exponent): the lanes that agree with the wave's first active lane are counted with one ballot and ONE atomic,
        // only the others go to the LDS atomic unit one by one (64 same-address atomics per wave instruction otherwise).
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const uint32_t key = fkey(x[i]);
          const bool act = (t + i * NT < V) && ((key & pmask) == prefix);
          const uint32_t dig = (key >> shift) & 255u;
          const unsigned long long am = __ballot(act);
          if (am != 0ull) {
            const int first = __ffsll((long long)am) - 1;
            const uint32_t d0 = (uint32_t)__shfl((int)dig, first);
            const unsigned long long same = __ballot(act && dig == d0);
            if ((int)(threadIdx.x & 63) == first) atomicAdd(&sm.hist[d0], (uint32_t)__popcll(same));
            if (act && dig != d0) atomicAdd(&sm.hist[dig], 1u);
          }
        }
        __syncthreads();
        // digit = the first d (from 255 down) with count(digits > d) + hist[d] >= kk, else 0; by wave 0: lane l owns digits 255 - 4l .. 252 - 4l,
        // an inclusive scan over the lanes gives every lane the count of all larger digits (the serial form walked 256 dependent LDS reads)
        if (t < 64) {
          int c[4], own = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            c[j] = (int)sm.hist[255 - 4 * t - j];
            own += c[j];
          }
          int incl = own;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (t >= o) incl += up;
          }
          int above = incl - own;     // keys whose digit is larger than every digit of this lane
          int dsel = -1, csel = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int d = 255 - 4 * t - j;
            if (dsel < 0 && d > 0 && above + c[j] >= kk) {
              dsel = d;
              csel = above;
            }
            above += c[j];
          }
          const unsigned long long hit = __ballot(dsel >= 0);
          if (hit != 0ull) {
            if (t == __ffsll((long long)hit) - 1) {
              sm.bcast[0] = (uint32_t)dsel;
              sm.bcast[1] = (uint32_t)(kk - csel);
            }
          } else if (t == 63) {          // no digit above 0 reaches kk: digit 0, everything above it counted (`above` now includes digit 0's lane mates 3..1)
            sm.bcast[0] = 0u;
            sm.bcast[1] = (uint32_t)(kk - (above - c[3]));
          }
        }
        __syncthreads();
        prefix |= sm.bcast[0] << shift;
        pmask |= 255u << shift;
        kk = (int)sm.bcast[1];
        __syncthreads();
      }
      const float kth = funkey(prefix);
#pragma unroll
      for (int i = 0; i < PER; ++i)
        if (x[i] < kth) x[i] = -INFINITY;
    }
  }

  // ---- 3. top-p (generate.py:38-53) ----------------------------------------------------------------------
  if constexpr (TOPP) {
    uint64_t* arr = reinterpret_cast<uint64_t*>(dyn);                 // VMAX pairs, 128 KiB
    double* scan = reinterpret_cast<double*>(dyn + (size_t)VMAX * 8);  // NT doubles
    uint32_t* flags = reinterpret_cast<uint32_t*>(dyn + (size_t)VMAX * 8 + NT * 8);  // VMAX bits
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int v = t + i * NT;
      const uint32_t dk = (v < V) ? ~fkey(x[i]) : 0xFFFFFFFFu;       // ascending dk == descending value
      arr[v] = ((uint64_t)dk << 32) | (uint32_t)v;
    }
    if (t < VMAX / 32 / 1) {
      for (int i = t; i < VMAX / 32; i += NT) flags[i] = 0;
    }
    __syncthreads();
    for (int ksz = 2; ksz <= VMAX; ksz <<= 1) {
      for (int j = ksz >> 1; j > 0; j >>= 1) {
#pragma unroll
        for (int i = 0; i < PER / 2; ++i) {
          const int id = t + i * NT;                   // VMAX/2 compare-exchanges per pass
          const int lo = ((id & ~(j - 1)) << 1) | (id & (j - 1));
          const int hi = lo | j;
          const bool up = (lo & ksz) == 0;
          const uint64_t a = arr[lo], c = arr[hi];
          if ((a > c) == up) {
            arr[lo] = c;
            arr[hi] = a;
          }
        }
        __syncthreads();
      }
    }
    // thread t owns sorted positions [PER*t, PER*t + PER)
    const float smax = funkey(~(uint32_t)(arr[0] >> 32));
    float e[PER];
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint64_t pr = arr[PER * t + i];
      const bool real = (uint32_t)pr < (uint32_t)V && (PER * t + i) < V;
      const float sv = funkey(~(uint32_t)(pr >> 32));
      e[i] = real ? expf(sv - smax) : 0.f;
      lsum += e[i];
    }
    const float S = block_sumf(lsum, sm);
    double run = 0.0;
    float cum[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      run += (double)(e[i] / S);
      cum[i] = (float)run;                      // local inclusive prefix (offset added below)
      e[i] = e[i] / S;
    }
    scan[t] = run;
    __syncthreads();
    // exclusive scan of the per-thread totals (Hillis-Steele on doubles, NT = 1024)
    for (int off = 1; off < NT; off <<= 1) {
      const double add = (t >= off) ? scan[t - off] : 0.0;
      __syncthreads();
      scan[t] += add;
      __syncthreads();
    }
    const double base = (t > 0) ? scan[t - 1] : 0.0;
    // remove sorted position j when cum[j-1] > top_p (shifted right by one, position 0 always kept)
    double prevd = base;
    {
      double r2 = base;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int j = PER * t + i;
        const bool rem = (j >= 1) && ((float)prevd > top_p);
        r2 += (double)e[i];
        prevd = r2;
        if (rem) {
          const uint32_t v = (uint32_t)arr[j];
          if (v < (uint32_t)V) atomicOr(&flags[v >> 5], 1u << (v & 31));
        }
      }
    }
    (void)cum;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int v = t + i * NT;
      if (v < V && (flags[v >> 5] >> (v & 31)) & 1u) x[i] = -INFINITY;
    }
    __syncthreads();
  }

  // ---- 4. softmax (generate.py:61) ---------------------------------------------------------------------------
  float lmax = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER; ++i) lmax = fmaxf(lmax, x[i]);
  const float gmax = block_max(lmax, sm);
  float ls = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    x[i] = (t + i * NT < V) ? expf(x[i] - gmax) : 0.f;
    ls += x[i];
  }
  const float gs = block_sumf(ls, sm);

  // ---- 5. pick: argmax(p / q) (multinomial) or argmax(p) (topk(probs,1)); first max wins ---------------------------
  float best = -1.f;
  int besti = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int v = t + i * NT;
    if (v < V) {
      const float p = __fdiv_rn(x[i], gs);
      if (probs_out) probs_out[(size_t)b * V + v] = p;
      float sc = p;
      if (sample_logits) {
        const float q = noise ? noise[((size_t)step * B_total + b_off + b) * V + v]
                              : exp1_noise(seed, (uint32_t)step, (uint32_t)(b_off + b), (uint32_t)v);
        sc = __fdiv_rn(p, q);
      }
      if (sc > best || (sc == best && v < besti)) {
        best = sc;
        besti = v;
      }
    }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oi = __shfl_xor(besti, o);
    if (ob > best || (ob == best && oi < besti)) {
      best = ob;
      besti = oi;
    }
  }
  __syncthreads();
  if ((t & 63) == 0) {
    sm.redf[t >> 6] = best;
    sm.redi[t >> 6] = besti;
  }
  __syncthreads();
  if (t == 0) {
    for (int i = 1; i < 16; ++i) {
      if (sm.redf[i] > best || (sm.redf[i] == best && sm.redi[i] < besti)) {
        best = sm.redf[i];
        besti = sm.redi[i];
      }
    }
    if (besti < 0 || besti >= V) besti = 0;   // all-NaN row: never hand an invalid id to the next step's gather
    if (out_ids) out_ids[(size_t)b * N + step] = besti;
    if (cur_tok) {
      cur_tok[b] = besti;
      if (cfg_on) cur_tok[b + B] = besti;
    }
  }
}

}  // namespace

int sample_rows(const float* logits, int B, int V, bool cfg_on, const vlg_sampling_params& sp, const float* noise,
                const StepState* state, int fixed_step, int N, int32_t* out_ids, int32_t* cur_tok, float* trace, float* probs,
                hipStream_t st, int b_off, int B_total, const int32_t* row_step) {
  if (B_total <= 0) B_total = B;
  if (V > VMAX || V < 1) {
    set_error("sampler: vocab %d not supported (max %d)", V, VMAX);
    return VLG_ERR_UNSUPPORTED;
  }
  const bool topp = sp.top_p < 1.0f;
  if (topp) {
    const size_t dyn = (size_t)VMAX * 8 + NT * 8 + VMAX / 8;
    static LdsAttrOnce attr_once;
    VLG_TRY(set_max_dynamic_lds(attr_once, {reinterpret_cast<const void*>(&sample_kernel<true>)}, (int)dyn));
    sample_kernel<true><<<B, NT, dyn, st>>>(logits, B, V, cfg_on ? 1 : 0, sp.cfg_scale, sp.cfg_interval, sp.temperature, sp.top_k,
                                            sp.top_p, sp.sample_logits, sp.seed, noise, state, fixed_step, N, out_ids, cur_tok, trace,
                                            probs, b_off, B_total, row_step);
  } else {
    sample_kernel<false><<<B, NT, 0, st>>>(logits, B, V, cfg_on ? 1 : 0, sp.cfg_scale, sp.cfg_interval, sp.temperature, sp.top_k,
                                           sp.top_p, sp.sample_logits, sp.seed, noise, state, fixed_step, N, out_ids, cur_tok, trace,
                                           probs, b_off, B_total, row_step);
  }
  return VLG_OK;
}

}  // namespace vlg
