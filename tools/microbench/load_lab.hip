// Load-path lab for the decode GEMMs (M = 32 rows): how fast can one workgroup per 16-column tile pull its weight tile (HBM, cold,
// read once) and the activation rows (L2, read by every workgroup) through the CU, by access shape?  36 distinct weight sets in a
// hipGraph chain like the real step.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/load_lab.hip -o gpurun_out/load_lab && gpurun_out/load_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t _e = (e);                                                        \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sink(u32x4 v, unsigned* out) {
  const unsigned s = v[0] ^ v[1] ^ v[2] ^ v[3];
  if (s == 0x12345678u) out[threadIdx.x] = s;   // never true for our data; keeps the loads alive
}

// MODE 0: fragment-shaped loads straight to VGPRs (16 rows x 64 B per wave-instruction), the shape of gemm_fused_kernel
// MODE 1: row-contiguous loads straight to VGPRs (1 KB of one row per wave-instruction)
// MODE 2: row-contiguous LDS-DMA (global_load_lds_dwordx4), then one ds_read_b128 per lane of what landed
// WHAT bit 0: weights, bit 1: activations
template <int MODE, int WHAT, int NW>
__global__ __launch_bounds__(64 * NW) void load_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w, int M, int K,
                                                       unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16;
  u32x4 acc = {0u, 0u, 0u, 0u};
  if constexpr (MODE == 0) {
    const int r = lane & 15, q = lane >> 4;
    const int nkb = K / 128;
    for (int kb = wave; kb < nkb; kb += NW) {
      u32x4 a[2][4], b[4];
      if (WHAT & 2) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) a[mt][s] = reinterpret_cast<const u32x4*>(x + (size_t)(mt * 16 + r) * K + kb * 128)[q + 4 * s];
      }
      if (WHAT & 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + (size_t)(n0 + r) * K + kb * 128) + q + 4 * s);
      }
      if (WHAT & 2) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc ^= a[mt][s];
      }
      if (WHAT & 1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc ^= b[s];
      }
    }
  } else if constexpr (MODE == 1) {
    // 1 KB pieces: piece p of a row-major [rows][K] matrix; pieces dealt round-robin to the waves
    const int ppr = K / 512;   // pieces per row
    if (WHAT & 2) {
      for (int p = wave; p < M * ppr; p += NW) acc ^= reinterpret_cast<const u32x4*>(x + (size_t)(p / ppr) * K + (p % ppr) * 512)[lane];
    }
    if (WHAT & 1) {
      for (int p = wave; p < 16 * ppr; p += NW)
        acc ^= __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + (size_t)(n0 + p / ppr) * K + (p % ppr) * 512) + lane);
    }
  } else if constexpr (MODE == 5 || MODE == 6 || MODE == 7) {
    // 4 x 256 B pieces; MODE 5: all X pieces, then all W pieces (per wave), K order rotated by the workgroup index;
    // MODE 6: the same without rotation; MODE 7: dedicated waves - waves 0..2 stream W, waves 3..7 stream X (rotated)
    const int nkg = K / 128;
    const int px = (WHAT & 2) ? M / 4 : 0, pw = (WHAT & 1) ? 4 : 0;
    const int lr = lane >> 4, lc = lane & 15;
    const int rot = (MODE == 6) ? 0 : (int)(blockIdx.x % nkg);
    auto ldx = [&](int p, int slot) {   // X piece p = kg * px + g
      const int kg = (p / px + rot) % nkg, g = p % px;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(x + (size_t)(g * 4 + lr) * K + (size_t)kg * 128 + lc * 8),
                                       (__attribute__((address_space(3))) void*)(lds + (size_t)slot * 1024), 16, 0, 0);
    };
    auto ldw = [&](int p, int slot) {
      const int kg = (p / pw + rot) % nkg, g = p % pw;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(w + (size_t)(n0 + g * 4 + lr) * K + (size_t)kg * 128 + lc * 8),
                                       (__attribute__((address_space(3))) void*)(lds + (size_t)slot * 1024), 16, 0, 2);
    };
    const int nxp = nkg * px, nwp = nkg * pw;
    if constexpr (MODE == 7) {
      const int WW = 3, XW = NW - WW;
      if (wave < WW) {
        for (int p = wave; p < nwp; p += WW) ldw(p, nxp + p);
      } else {
        for (int p = wave - WW; p < nxp; p += XW) ldx(p, p);
      }
    } else {
      for (int p = wave; p < nxp; p += NW) ldx(p, p);
      for (int p = wave; p < nwp; p += NW) ldw(p, nxp + p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s2 = wave; s2 < nxp + nwp; s2 += NW) acc ^= reinterpret_cast<const u32x4*>(lds + (size_t)s2 * 1024)[lane];
  } else if constexpr (MODE == 3 || MODE == 4) {
    // the GEMM kernel's order: per K group, the X pieces then the W pieces; a piece = RP rows x (1024 / RP) bytes
    constexpr int RP = MODE == 3 ? 4 : 2;          // rows per piece
    constexpr int SEG = 1024 / RP;                 // bytes per row segment
    const int nseg = K * 2 / SEG;                  // K groups
    const int px = (WHAT & 2) ? M / RP : 0, pw = (WHAT & 1) ? 16 / RP : 0, pp = px + pw;
    const int lr = lane / (SEG / 16), lc = lane % (SEG / 16);
    int slot = wave;
    for (int p = wave; p < nseg * pp; p += NW, slot += NW) {
      const int kg = p / pp, g = p - kg * pp;
      const unsigned short* src = g < px ? x + (size_t)(g * RP + lr) * K + (size_t)kg * (SEG / 2) + lc * 8
                                         : w + (size_t)(n0 + (g - px) * RP + lr) * K + (size_t)kg * (SEG / 2) + lc * 8;
      if (g < px)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(src), (__attribute__((address_space(3))) void*)(lds + (size_t)slot * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(src), (__attribute__((address_space(3))) void*)(lds + (size_t)slot * 1024), 16, 0, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s2 = wave; s2 < nseg * pp; s2 += NW) acc ^= reinterpret_cast<const u32x4*>(lds + (size_t)s2 * 1024)[lane];
  } else {
    const int ppr = K / 512;
    int slot = wave;   // LDS piece index; NW pieces are written per round
    const int nx = (WHAT & 2) ? M * ppr : 0, nw_ = (WHAT & 1) ? 16 * ppr : 0;
    for (int p = wave; p < nx; p += NW, slot += NW)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(x + (size_t)(p / ppr) * K + (p % ppr) * 512) + lane,
                                       (__attribute__((address_space(3))) void*)(lds + (size_t)slot * 1024), 16, 0, 0);
    for (int p = wave; p < nw_; p += NW, slot += NW)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(w + (size_t)(n0 + p / ppr) * K + (p % ppr) * 512) + lane,
                                       (__attribute__((address_space(3))) void*)(lds + (size_t)slot * 1024), 16, 0, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = wave; s < nx + nw_; s += NW) acc ^= reinterpret_cast<const u32x4*>(lds + (size_t)s * 1024)[lane];
  }
  sink(acc, out);
}

__global__ void fill_kernel(unsigned short* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    p[i] = (unsigned short)(h >> 16) | 1;
  }
}

template <int MODE, int WHAT, int NW>
float run(const char* name, const std::vector<unsigned short*>& ws, unsigned short* x, int M, int N, int K, unsigned* out, hipStream_t st) {
  const size_t ldsb = MODE >= 2 ? (size_t)(((WHAT & 2) ? M : 0) + ((WHAT & 1) ? 16 : 0)) * K * 2 : 0;
  if (ldsb > 160 * 1024) {
    printf("%-44s skipped (%zu KB LDS)\n", name, ldsb / 1024);
    return 0;
  }
  if (ldsb > 64 * 1024)
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(load_kernel<MODE, WHAT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipGraph_t g;
  hipGraphExec_t e;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (size_t l = 0; l < ws.size(); ++l) load_kernel<MODE, WHAT, NW><<<N / 16, 64 * NW, ldsb, st>>>(x, ws[l], M, K, out);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(e, st));
  CK(hipStreamSynchronize(st));
  float tot = 0;
  const int reps = 20;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(e, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms;
  }
  const float us = tot / reps * 1e3f / ws.size();
  printf("%-44s %7.2f us per launch  (weights %.1f MB -> %.2f TB/s)\n", name, us, (double)N * K * 2 / 1e6, (double)N * K * 2 / us / 1e6);
  CK(hipGraphExecDestroy(e));
  CK(hipGraphDestroy(g));
  return us;
}

int main(int argc, char** argv) {
  const int M = 32, L = 36;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* out;
  CK(hipMalloc(&out, 4096));
  struct Shape { const char* n; int N, K; } shapes[] = {{"qkv", 3840, 1280}, {"wo", 1280, 1280}, {"w13", 7168, 1280}, {"w2", 1280, 3584}};
  for (auto& sh : shapes) {
    std::vector<unsigned short*> ws(L);
    for (int l = 0; l < L; ++l) {
      CK(hipMalloc(&ws[l], (size_t)sh.N * sh.K * 2));
      fill_kernel<<<1024, 256>>>(ws[l], (size_t)sh.N * sh.K, 17 * l + 3);
    }
    // pad between sets with other allocations so that chains do not sit in one hot region
    unsigned short* x;
    CK(hipMalloc(&x, (size_t)M * sh.K * 2));
    fill_kernel<<<64, 256>>>(x, (size_t)M * sh.K, 99);
    CK(hipDeviceSynchronize());
    printf("---- %s: N %d, K %d, M %d, %d workgroups (\"rows\" modes cover %d of %d K elements) ----\n", sh.n, sh.N, sh.K, M, sh.N / 16, sh.K / 512 * 512, sh.K);
    char nm[96];
#define RUN(MODE, WHAT, NW, label)                                              \
  snprintf(nm, sizeof nm, "%s [%s] %d waves", label, #WHAT, NW);                \
  run<MODE, WHAT, NW>(nm, ws, x, M, sh.N, sh.K, out, st);
    RUN(0, 3, 4, "fragment loads -> VGPR, W+X");
    RUN(0, 1, 4, "fragment loads -> VGPR, W only");
    RUN(0, 2, 4, "fragment loads -> VGPR, X only");
    RUN(0, 3, 8, "fragment loads -> VGPR, W+X");
    RUN(1, 3, 4, "row loads -> VGPR, W+X");
    RUN(1, 1, 4, "row loads -> VGPR, W only");
    RUN(1, 2, 4, "row loads -> VGPR, X only");
    RUN(1, 3, 8, "row loads -> VGPR, W+X");
    RUN(2, 3, 4, "LDS-DMA rows, W+X");
    RUN(2, 1, 4, "LDS-DMA rows, W only");
    RUN(2, 2, 4, "LDS-DMA rows, X only");
    RUN(2, 3, 8, "LDS-DMA rows, W+X");
    RUN(3, 3, 8, "LDS-DMA 4x256B pieces, K-major, W+X");
    RUN(3, 1, 8, "LDS-DMA 4x256B pieces, K-major, W only");
    RUN(3, 2, 8, "LDS-DMA 4x256B pieces, K-major, X only");
    RUN(6, 3, 8, "LDS-DMA 4x256B, X then W, W+X");
    RUN(5, 3, 8, "LDS-DMA 4x256B, X then W, rotated, W+X");
    RUN(5, 2, 8, "LDS-DMA 4x256B, rotated, X only");
    RUN(6, 2, 8, "LDS-DMA 4x256B, X only");
    RUN(7, 3, 8, "LDS-DMA 4x256B, dedicated waves 3W+5X, rotated");
    RUN(4, 3, 8, "LDS-DMA 2x512B pieces, K-major, W+X");
    RUN(4, 1, 8, "LDS-DMA 2x512B pieces, K-major, W only");
    RUN(2, 1, 8, "LDS-DMA rows, W only");
    RUN(2, 2, 8, "LDS-DMA rows, X only");
    for (auto p : ws) CK(hipFree(p));
    CK(hipFree(x));
  }
  // an empty kernel chain: the boundary floor
  return 0;
}
