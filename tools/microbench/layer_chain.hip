// Micro-benchmark of the decode layer's four fused skinny GEMMs as a hipGraph chain over L distinct weight sets (cold weights, as in
// the real step), with in-kernel timestamps.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVLG_KTRACE -I video-llamagen_amd/csrc -I include tools/microbench/layer_chain.hip \
//         video-llamagen_amd/csrc/core.hip -o gpurun_out/layer_chain && gpurun_out/layer_chain
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../video-llamagen_amd/csrc/gemm_fused.hip"

using namespace vlg;

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    const float v = ((h & 0xffff) / 65536.0f - 0.5f) * scale;
    p[i] = f32_to_bf16(v);
  }
}

// experiment: pull a weight matrix through the caches ahead of the kernel that uses it (plain loads, results folded into one dead store)
__global__ __launch_bounds__(256) void prefetch_kernel(const uint4* p, size_t n16, unsigned* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n16; i += stride) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// fragment-major copy of a [N][K] bf16 matrix (gpt_kernels.h: relayout_fragment_major)
__global__ __launch_bounds__(256) void lab_fm_kernel(const uint4* __restrict__ w, uint4* __restrict__ out, long long chunks, int nks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= chunks) return;
  const int l = (int)(i & 63);
  const long long blk = i >> 6;
  const long long t = blk / nks;
  const int s = (int)(blk - t * nks);
  out[i] = w[(t * 16 + (l & 15)) * (long long)(nks * 4) + s * 4 + (l >> 4)];
}
static bf16* fm_copy(const bf16* w, int N, int K) {
  bf16* o;
  CK(hipMalloc(&o, (size_t)N * K * 2));
  const long long chunks = (long long)N * K / 8;
  lab_fm_kernel<<<dim3((unsigned)((chunks + 255) / 256)), 256>>>((const uint4*)w, (uint4*)o, chunks, K / 32);
  return o;
}

static bf16* alloc_fill(size_t n, uint32_t seed, float scale) {
  bf16* p;
  CK(hipMalloc(&p, n * 2));
  fill_kernel<<<1024, 256>>>((uint16_t*)p, n, seed, scale);
  return p;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32;
  const int L = 36, D = 1280, F = 3584, H = 20, hd = 64, S = 8;
  const int reps = 30;
  const bool static_a = argc > 2 && atoi(argv[2]) == 1;   // experiment: prologue kernels read a never-written activation buffer
  const int prefetch = argc > 3 ? atoi(argv[3]) : 0;      // 1: read each layer's weights right before its four kernels (warm MALL / L2)
  const int fmode = argc > 4 ? atoi(argv[4]) : 2;         // 0: row-major everything, 1: fragment-major weights, 2: + A-fragment-major activations
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  std::vector<bf16*> wqkv(L), wo(L), w13(L), w2(L), nw1(L), nw2(L);
  for (int l = 0; l < L; ++l) {
    wqkv[l] = alloc_fill((size_t)3 * D * D, 11 * l + 1, 0.04f);
    wo[l] = alloc_fill((size_t)D * D, 11 * l + 2, 0.04f);
    w13[l] = alloc_fill((size_t)2 * F * D, 11 * l + 3, 0.04f);
    w2[l] = alloc_fill((size_t)D * F, 11 * l + 4, 0.04f);
    if (fmode >= 1) {
      bf16* t;
      t = fm_copy(wqkv[l], 3 * D, D); CK(hipDeviceSynchronize()); CK(hipFree(wqkv[l])); wqkv[l] = t;
      t = fm_copy(wo[l], D, D); CK(hipDeviceSynchronize()); CK(hipFree(wo[l])); wo[l] = t;
      t = fm_copy(w13[l], 2 * F, D); CK(hipDeviceSynchronize()); CK(hipFree(w13[l])); w13[l] = t;
      t = fm_copy(w2[l], D, F); CK(hipDeviceSynchronize()); CK(hipFree(w2[l])); w2[l] = t;
    }
    nw1[l] = alloc_fill(D, 11 * l + 5, 2.0f);
    nw2[l] = alloc_fill(D, 11 * l + 6, 2.0f);
  }
  bf16* x = alloc_fill((size_t)M * D, 777, 2.0f);
  bf16* x0 = alloc_fill((size_t)M * D, 777, 2.0f);
  bf16* q = alloc_fill((size_t)M * D, 778, 1.0f);
  bf16* ao = alloc_fill((size_t)M * D, 779, 1.0f);
  bf16* g = alloc_fill((size_t)M * F, 780, 1.0f);
  bf16* kc = alloc_fill((size_t)M * H * S * hd, 781, 1.0f);
  bf16* vc = alloc_fill((size_t)M * H * S * hd, 782, 1.0f);
  float* freqs;
  CK(hipMalloc(&freqs, 64 * hd * 4));
  CK(hipMemset(freqs, 0, 64 * hd * 4));
  StepState* state;
  CK(hipMalloc(&state, sizeof(StepState)));
  CK(hipMemset(state, 0, sizeof(StepState)));
  const int NK = 4 * L, MAXG = 512;
  unsigned long long* trace;
  CK(hipMalloc(&trace, (size_t)NK * MAXG * 4 * 8));
  CK(hipMemset(trace, 0, (size_t)NK * MAXG * 4 * 8));
  CK(hipDeviceSynchronize());

  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<dim3> grids(NK);
  auto enqueue = [&](bool tr) {
    int k = 0;
    for (int l = 0; l < L; ++l) {
      if (prefetch) {
        prefetch_kernel<<<1024, 256, 0, st>>>((const uint4*)wqkv[l], (size_t)3 * D * D / 8, sink);
        prefetch_kernel<<<1024, 256, 0, st>>>((const uint4*)wo[l], (size_t)D * D / 8, sink);
        prefetch_kernel<<<1024, 256, 0, st>>>((const uint4*)w13[l], (size_t)2 * F * D / 8, sink);
        prefetch_kernel<<<1024, 256, 0, st>>>((const uint4*)w2[l], (size_t)D * F / 8, sink);
      }
      FusedGemm fa;
      fa.norm_w = nw1[l];
      fa.qbuf = q; fa.kc = kc; fa.vc = vc; fa.freqs = freqs; fa.state = state; fa.Tq = 1; fa.H = H; fa.hd = hd; fa.S = S;
      fa.trace = tr ? trace + (size_t)k * MAXG * 4 : nullptr; ++k;
      if (fmode >= 1) fa.wfm = wqkv[l];
      fa.a_fm = fmode >= 2;
      if (gemm_fused<bf16>(static_a ? x0 : x, wqkv[l], M, 3 * D, D, true, EPI_QKV, fa, st)) { fprintf(stderr, "qkv fail\n"); exit(1); }
      FusedGemm fb;
      fb.h = x;
      if (fmode >= 1) fb.wfm = wo[l];
      fb.a_fm = fb.o_fm = fmode >= 2;
      fb.trace = tr ? trace + (size_t)k * MAXG * 4 : nullptr; ++k;
      if (gemm_fused<bf16>(ao, wo[l], M, D, D, false, EPI_RESID, fb, st)) exit(1);
      FusedGemm fc;
      fc.norm_w = nw2[l];
      fc.out = g;
      if (fmode >= 1) fc.wfm = w13[l];
      fc.a_fm = fc.o_fm = fmode >= 2;
      fc.trace = tr ? trace + (size_t)k * MAXG * 4 : nullptr; ++k;
      if (gemm_fused<bf16>(static_a ? x0 : x, w13[l], M, F, D, true, EPI_SWIGLU, fc, st)) exit(1);
      FusedGemm fd;
      fd.h = x;
      if (fmode >= 1) fd.wfm = w2[l];
      fd.a_fm = fd.o_fm = fmode >= 2;
      fd.trace = tr ? trace + (size_t)k * MAXG * 4 : nullptr; ++k;
      if (gemm_fused<bf16>(g, w2[l], M, D, F, false, EPI_RESID, fd, st)) exit(1);
    }
  };
  for (int tr = 0; tr < 2; ++tr) {
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue(tr);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) {
      CK(hipMemcpyAsync(x, x0, (size_t)M * D * 2, hipMemcpyDeviceToDevice, st));
      CK(hipGraphLaunch(exec, st));
    }
    CK(hipStreamSynchronize(st));
    float tot = 0;
    for (int i = 0; i < reps; ++i) {
      CK(hipMemcpyAsync(x, x0, (size_t)M * D * 2, hipMemcpyDeviceToDevice, st));
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(exec, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      tot += ms;
    }
    printf("M=%d trace=%d: %.2f us per layer (4 GEMMs), %.1f us per step-chain\n", M, tr, tot / reps * 1e3 / L, tot / reps * 1e3);
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
  }
  // ---- trace analysis (last replay) ----
  std::vector<unsigned long long> tr((size_t)NK * MAXG * 4);
  CK(hipMemcpy(tr.data(), trace, tr.size() * 8, hipMemcpyDeviceToHost));
  const char* names[4] = {"qkv(norm,rope)", "wo(resid)", "w13(norm,swiglu)", "w2(resid)"};
  double acc[4][8] = {};
  unsigned long long prev_end = 0;
  for (int k = 0; k < NK; ++k) {
    const unsigned long long* t = tr.data() + (size_t)k * MAXG * 4;
    int nwg = 0;
    unsigned long long s_min = ~0ull, s_max = 0, e_max = 0;
    double body = 0, red = 0, epi = 0, bmax = 0;
    for (int w = 0; w < MAXG; ++w) {
      if (!t[w * 4]) continue;
      ++nwg;
      s_min = std::min(s_min, t[w * 4]);
      s_max = std::max(s_max, t[w * 4]);
      e_max = std::max(e_max, t[w * 4 + 3]);
      body += (double)(t[w * 4 + 1] - t[w * 4]);
      bmax = std::max(bmax, (double)(t[w * 4 + 1] - t[w * 4]));
      red += (double)(t[w * 4 + 2] - t[w * 4 + 1]);
      epi += (double)(t[w * 4 + 3] - t[w * 4 + 2]);
    }
    if (!nwg) continue;
    double* a = acc[k % 4];
    if (k >= 4) {   // skip the first layer (no predecessor)
      a[0] += 1;
      a[1] += (double)(s_min - prev_end);          // boundary: previous kernel's last end -> first start
      a[2] += (double)(s_max - s_min);             // launch ramp
      a[3] += body / nwg;
      a[4] += bmax;
      a[5] += red / nwg;
      a[6] += epi / nwg;
      a[7] += (double)(e_max - s_min);             // span
    }
    prev_end = e_max;
    if (k < 4) printf("%-18s workgroups %d\n", names[k], nwg);
  }
  printf("%-18s %9s %9s %9s %9s %9s %9s %9s   (us, mean over layers; 10 ns clock)\n", "kernel", "boundary", "ramp", "body", "body_max", "reduce", "epilogue", "span");
  for (int j = 0; j < 4; ++j) {
    const double n = acc[j][0] * 100.0;   // 100 ticks per us
    printf("%-18s %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f\n", names[j], acc[j][1] / n, acc[j][2] / n, acc[j][3] / n, acc[j][4] / n, acc[j][5] / n,
           acc[j][6] / n, acc[j][7] / n);
  }
  return 0;
}
