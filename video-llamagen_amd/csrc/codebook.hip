// Video codebook (n_codes x dim, default 2048 x 256): Codebook.forward in eval mode.
//
// Replaces: Codebook.forward, tokenizer/tokenizer_video/vqvae.py:161-209 == CausalVideoVAE/causalvideovae/model/modules/quant.py:42-96
//   flat = 'b c t h w -> (b t h w) c';  d = (|x|^2 - 2 x E^T) + |E|^2;  idx = argmin_j d (first minimum);
//   embeddings_st = (E[idx] - z) + z;  commitment_loss = 0.25 * mean((z - E[idx])^2);
//   perplexity = exp(-sum_j p_j log(p_j + 1e-10)), p_j = count_j / n.
//
// The distance matrix is a [n, dim] x [dim, n_codes] contraction with fp32 inputs: it runs on the matrix cores with the exact-fp32
// v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, no reduced precision), so nearest-neighbour decisions differ from a CPU BLAS only
// by fp32 summation order.  HBM/L2 traffic: z once, E re-read once per 64-row workgroup from L2 (2 MB, resident).
//
// Workgroup = 4 waves = 64 z rows.  The z tile sits in LDS ([64][dim + 4] fp32: the 16-byte pad makes the 16 rows of a ds_read_b128
// lane group fall on 16 different 16-byte slots); wave w owns the codes {32 (4 t + w) .. + 31}, t = 0 .. n_codes/128 - 1, and for
// each of them runs two 32x32 accumulators (row tiles 0-31 and 32-63) over K: one E fragment (a 16-byte load per lane, straight
// from L2 into registers - each code row is consumed by exactly one wave of the workgroup) feeds 8 MFMAs.  The K index is
// permuted so that lane half h owns the contiguous K range [h dim/2, (h+1) dim/2): the fragment of MFMA step s is element s of that
// range for both operands.  The epilogue applies the reference's rounding sequence fl(fl(zz - 2 dot) + ee), keeps a running
// (min, first index) per row in registers, and the 32 lanes x 4 waves are merged by shuffles and LDS at the end.
#include <algorithm>
#include <map>
#include <mutex>

#include "conv_kernels.h"

namespace vlg {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ E, int n_e, int dim, float* __restrict__ ee) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_e) return;
  float s = 0.f;
  for (int c = lane; c < dim; c += 64) {
    const float v = E[(size_t)row * dim + c];
    s = fmaf(v, v, s);
  }
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) ee[row] = s;
}

// z row r = (batch b, position pos): element c at z + b*zs_batch + pos*zs_row + c*zs_c
__global__ __launch_bounds__(256) void codebook_mfma_kernel(const float* __restrict__ z, long long zs_row, long long zs_c, long long rows_per_batch,
                                                            long long zs_batch, const float* __restrict__ E, const float* __restrict__ ee,
                                                            long long n, int n_e, int dim, int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ld = dim + 4;                                  // padded row stride (floats)
  float* zt = reinterpret_cast<float*>(smem_raw);          // [64][ld]
  float* zz = zt + (size_t)64 * ld;                        // [64] |z|^2
  float* bestv = zz + 64;                                  // [4][64]
  int* besti = reinterpret_cast<int*>(bestv + 256);        // [4][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long row0 = (long long)blockIdx.x * 64;

  // ---- stage the z tile (rows beyond n are zero-filled: their results are never stored) ----
  if (zs_c == 1) {                                         // row-major z [n, dim]: consecutive threads walk a row
    for (int e = tid; e < 64 * dim; e += 256) {
      const int i = e / dim, c = e - i * dim;
      const long long r = row0 + i;
      zt[(size_t)i * ld + c] = r < n ? z[(r / rows_per_batch) * zs_batch + (r % rows_per_batch) * zs_row + c] : 0.f;
    }
  } else {                                                 // planar [b, c, pos]: consecutive threads = consecutive positions
    const int i = tid & 63;
    const long long r = row0 + i;
    const bool ok = r < n;
    const float* zp = ok ? z + (r / rows_per_batch) * zs_batch + (r % rows_per_batch) * zs_row : z;
    for (int c = tid >> 6; c < dim; c += 4) zt[(size_t)i * ld + c] = ok ? zp[(size_t)c * zs_c] : 0.f;
  }
  __syncthreads();
  if (tid < 64) {
    float s = 0.f;
    for (int c = 0; c < dim; ++c) {
      const float v = zt[(size_t)tid * ld + c];
      s = fmaf(v, v, s);
    }
    zz[tid] = s;
  }
  __syncthreads();

  const int i32 = lane & 31, h = lane >> 5;
  const int half = dim >> 1;                               // host guarantees dim % 8 == 0
  const float* a0p = zt + (size_t)i32 * ld + h * half;          // row tile 0
  const float* a1p = zt + (size_t)(32 + i32) * ld + h * half;   // row tile 1
  float best[2][16];
  int bidx[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) best[t][r] = INFINITY, bidx[t][r] = 0x7fffffff;
  float zrow[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) zrow[t][r] = zz[t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];

  const int ntile = (n_e + 31) / 32;
  for (int tile = wave; tile < ntile; tile += 4) {
    const int j = tile * 32 + i32;                         // this lane's code (column of both accumulators)
    const int jc = j < n_e ? j : n_e - 1;
    const f32x4v_t* ep = reinterpret_cast<const f32x4v_t*>(E + (size_t)jc * dim + h * half);
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
    f32x4v_t bn = ep[0];
    for (int s4 = 0; s4 < half / 4; ++s4) {
      const f32x4v_t b = bn;
      if (s4 + 1 < half / 4) bn = ep[s4 + 1];
      const f32x4v_t a0 = *reinterpret_cast<const f32x4v_t*>(a0p + 4 * s4);
      const f32x4v_t a1 = *reinterpret_cast<const f32x4v_t*>(a1p + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b[e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b[e], acc1, 0, 0, 0);
      }
    }
    const float eej = ee[jc];
    if (j < n_e) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d0 = __fadd_rn(__fsub_rn(zrow[0][r], __fmul_rn(2.0f, acc0[r])), eej);
        const float d1 = __fadd_rn(__fsub_rn(zrow[1][r], __fmul_rn(2.0f, acc1[r])), eej);
        if (d0 < best[0][r]) best[0][r] = d0, bidx[0][r] = j;    // tiles are visited in ascending j: strict < keeps the first minimum
        if (d1 < best[1][r]) best[1][r] = d1, bidx[1][r] = j;
      }
    }
  }
  // ---- merge: 32 columns of one lane half (xor 1..16 stays inside the half), then the 4 waves ----
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float bv = best[t][r];
      int bi = bidx[t][r];
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov < bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
      }
      if (i32 == 0) {
        const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        bestv[wave * 64 + row] = bv;
        besti[wave * 64 + row] = bi;
      }
    }
  __syncthreads();
  if (tid < 64 && row0 + tid < n) {
    float bv = bestv[tid];
    int bi = besti[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = bestv[w * 64 + tid];
      const int oi = besti[w * 64 + tid];
      if (ov < bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
    }
    idx[row0 + tid] = bi;
  }
}

// embeddings_st, squared-error partial sums (fixed order, double) and the usage histogram (integer atomics: exact, order-free)
__global__ __launch_bounds__(256) void codebook_gather_kernel(const float* __restrict__ z, const float* __restrict__ E, const int32_t* __restrict__ idx,
                                                              long long n_pos, int B, int dim, float* __restrict__ emb_st,
                                                              double* __restrict__ partial, int* __restrict__ hist) {
  // z / emb_st planar [B][dim][n_pos]; thread = one (b, pos), loops over c (coalesced over pos)
  const long long total = (long long)B * n_pos;
  double acc = 0.0;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (long long)gridDim.x * blockDim.x) {
    const long long b = r / n_pos, pos = r % n_pos;
    const int code = idx[r];
    atomicAdd(hist + code, 1);
    const float* e = E + (size_t)code * dim;
    const float* zp = z + (size_t)b * dim * n_pos + pos;
    float* op = emb_st + (size_t)b * dim * n_pos + pos;
    for (int c = 0; c < dim; ++c) {
      const float zv = zp[(size_t)c * n_pos], ev = e[c];
      op[(size_t)c * n_pos] = __fadd_rn(__fsub_rn(ev, zv), zv);     // (embeddings - z).detach() + z
      const float df = __fsub_rn(zv, ev);
      acc += (double)__fmul_rn(df, df);
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void codebook_stats_kernel(const double* __restrict__ partial, int nblk, const int* __restrict__ hist, int n_e,
                                                             long long n_rows, long long n_elem, float* __restrict__ out2) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) a += partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double sq = red[0];
  __syncthreads();
  double ent = 0.0;
  for (int j = threadIdx.x; j < n_e; j += 256) {
    const float p = (float)hist[j] / (float)n_rows;                 // torch.mean of the one-hot column (exact count / n in fp32)
    ent += (double)__fmul_rn(p, logf(__fadd_rn(p, 1e-10f)));
  }
  red[threadIdx.x] = ent;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = 0.25f * (float)(sq / (double)n_elem);                 // 0.25 * F.mse_loss(z, embeddings)
    out2[1] = expf(-(float)red[0]);                                 // perplexity
  }
}

// Scratch (code norms, histogram, partial sums) is kept per STREAM: calls on one stream are ordered by it and may share buffers, calls
// on different streams - two Codebook objects, two threads - must not.  cb_mu covers the pool and a call's host side (enqueue only).
struct CbScratch {
  DevBuf ee, idx, partial, hist;
};
std::mutex cb_mu;
CbScratch& cb_scratch(hipStream_t st) {   // under cb_mu; std::map keeps references valid across insertions
  static std::map<hipStream_t, CbScratch> pool;
  return pool[st];
}

}  // namespace

bool codebook_mfma_ok(int n_e, int dim) { return dim % 8 == 0 && dim >= 8 && dim <= 512 && n_e >= 1; }

// idx[r] = argmin_j (|z_r|^2 - 2 z_r.e_j) + |e_j|^2 on the matrix cores; ee_scratch: n_e floats
int codebook_argmin_mfma(const float* z, long long zs_row, long long zs_c, long long rows_per_batch, long long zs_batch, const float* E,
                         float* ee_scratch, long long n, int n_e, int dim, int32_t* idx, hipStream_t st) {
  row_sumsq_kernel<<<cdiv(n_e, 4), 256, 0, st>>>(E, n_e, dim, ee_scratch);
  const size_t lds = ((size_t)64 * (dim + 4) + 64 + 256) * sizeof(float) + 256 * sizeof(int);
  static LdsAttrOnce attr_once;
  VLG_TRY(set_max_dynamic_lds(attr_once, {reinterpret_cast<const void*>(codebook_mfma_kernel)}, 160 * 1024));
  codebook_mfma_kernel<<<(unsigned)cdiv64(n, 64), 256, lds, st>>>(z, zs_row, zs_c, rows_per_batch, zs_batch, E, ee_scratch, n, n_e, dim, idx);
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}

}  // namespace vlg

using namespace vlg;

extern "C" int vlg_codebook_forward(const float* d_z, const float* d_codebook, int32_t B, int32_t dim, int64_t n_pos, int32_t n_codes,
                                    int32_t* d_encodings, float* d_embeddings_st, float* d_loss_perplexity, void* stream) {
  VLG_CHECK(d_z && d_codebook && d_encodings && B > 0 && dim > 0 && n_pos > 0 && n_codes > 0, VLG_ERR_BAD_ARG, "vlg_codebook_forward: bad argument");
  VLG_CHECK((d_embeddings_st != nullptr) == (d_loss_perplexity != nullptr), VLG_ERR_BAD_ARG,
            "vlg_codebook_forward: embeddings and loss/perplexity outputs come together");
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(cb_mu);
  CbScratch& s = cb_scratch(st);
  const long long n = (long long)B * n_pos;
  const int nblk = (int)std::min<long long>(cdiv64(n, 256), 1024);
  if (s.ee.bytes < (size_t)n_codes * sizeof(float) || s.partial.bytes < (size_t)nblk * sizeof(double) || s.hist.bytes < (size_t)n_codes * sizeof(int)) {
    VLG_HIP(hipStreamSynchronize(st));
    VLG_TRY(s.ee.reserve((size_t)n_codes * sizeof(float)));
    VLG_TRY(s.partial.reserve((size_t)1024 * sizeof(double)));
    VLG_TRY(s.hist.reserve((size_t)n_codes * sizeof(int)));
  }
  // z planar [B][dim][n_pos]: row (b, pos) element c at b*dim*n_pos + c*n_pos + pos   ('b c t h w -> (b t h w) c')
  if (codebook_mfma_ok(n_codes, dim))
    VLG_TRY(codebook_argmin_mfma(d_z, 1, n_pos, n_pos, (long long)dim * n_pos, d_codebook, s.ee.as<float>(), n, n_codes, dim, d_encodings, st));
  else
    VLG_TRY(codebook_argmin(d_z, 1, n_pos, n_pos, (long long)dim * n_pos, d_codebook, n, n_codes, dim, false, d_encodings, st));
  if (!d_embeddings_st) return VLG_OK;
  VLG_HIP(hipMemsetAsync(s.hist.p, 0, (size_t)n_codes * sizeof(int), st));
  codebook_gather_kernel<<<nblk, 256, 0, st>>>(d_z, d_codebook, d_encodings, n_pos, B, dim, d_embeddings_st, s.partial.as<double>(), s.hist.as<int>());
  codebook_stats_kernel<<<1, 256, 0, st>>>(s.partial.as<double>(), nblk, s.hist.as<int>(), n_codes, n, n * dim, d_loss_perplexity);
  VLG_HIP(hipGetLastError());
  return VLG_OK;
}

extern "C" int vlg_codebook_argmin(const float* d_z, const float* d_codebook, int32_t n, int32_t n_codes, int32_t dim, int32_t* d_idx,
                                   void* stream) {
  VLG_CHECK(d_z && d_codebook && d_idx && n > 0 && n_codes > 0 && dim > 0, VLG_ERR_BAD_ARG, "vlg_codebook_argmin: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!codebook_mfma_ok(n_codes, dim)) return codebook_argmin(d_z, dim, 1, n, 0, d_codebook, n, n_codes, dim, false, d_idx, st);
  std::lock_guard<std::mutex> lk(cb_mu);
  CbScratch& s = cb_scratch(st);
  if (s.ee.bytes < (size_t)n_codes * sizeof(float)) {
    VLG_HIP(hipStreamSynchronize(st));
    VLG_TRY(s.ee.reserve((size_t)n_codes * sizeof(float)));
  }
  return codebook_argmin_mfma(d_z, dim, 1, n, 0, d_codebook, s.ee.as<float>(), n, n_codes, dim, d_idx, st);
}
