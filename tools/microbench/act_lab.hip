// Lab: with the weights arriving as whole lines (fragment-major copies), does the SHAPE of the activation loads matter?
// 240 workgroups x 256 threads (the 32-row QKV decode GEMM of GPT-XL): each streams its own 40 KB of cold weights (1 KB per wave
// instruction) and reads the SAME [32][1280] bf16 activation rows (80 KB, L2-resident after the first touch) either as MFMA A fragments
// from the row-major matrix (a wave instruction = 16 rows x 64 bytes: 16 half lines) or from a fragment-major copy (1 KB contiguous).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) * gptr16;

template <int AMODE, bool WITH_W>
__global__ __launch_bounds__(256) void k(const char* w, const char* x, int M, int K, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = K * 2 / 64;           // 40
  const int mtiles = M / 16;               // 2
  unsigned acc = 0;
  // weights: this workgroup's 16-column tile, ksteps KB, K steps dealt to the 4 waves
  u32x4 wv[10];
  if (WITH_W) {
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int s = wave + 4 * j;
      wv[j] = __builtin_nontemporal_load((gptr16)(uintptr_t)(w + ((size_t)blockIdx.x * ksteps + (s < ksteps ? s : 0)) * 1024 + lane * 16));
    }
  }
  // activations: this wave's K steps of every m-tile
  u32x4 av[2][10];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int s = wave + 4 * j, sc = s < ksteps ? s : 0;
      const char* p = AMODE == 0 ? x + ((size_t)(mt * 16 + r) * K * 2 + (size_t)sc * 64 + q * 16)      // row-major fragment
                                 : x + (((size_t)mt * ksteps + sc) * 64 + lane) * 16;                  // fragment-major
      av[mt][j] = *(gptr16)(uintptr_t)p;
    }
  if (WITH_W)
#pragma unroll
    for (int j = 0; j < 10; ++j) acc ^= wv[j][0] ^ wv[j][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 10; ++j) acc ^= av[mt][j][1] ^ av[mt][j][3];
  if (acc == 0x12345679u) sink[0] = acc;
  (void)mtiles;
}

int main() {
  const int M = 32, K = 1280, grid = 240, NB = 40;
  const size_t wbytes = (size_t)grid * (K * 2 / 64) * 1024;
  std::vector<char*> wb(NB);
  for (auto& b : wb) { CK(hipMalloc(&b, wbytes)); CK(hipMemset(b, 1, wbytes)); }
  std::vector<char*> xb(NB);
  for (auto& b : xb) { CK(hipMalloc(&b, (size_t)M * K * 2)); CK(hipMemset(b, 2, (size_t)M * K * 2)); }
  unsigned* sink; CK(hipMalloc(&sink, 64));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int amode, bool with_w) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < NB; ++i) {   // a chain of launches on fresh weights and fresh activation rows, as the layers of a decode step
        if (amode == 0 && with_w) k<0, true><<<grid, 256, 0, st>>>(wb[i], xb[i], M, K, sink);
        else if (amode == 1 && with_w) k<1, true><<<grid, 256, 0, st>>>(wb[i], xb[i], M, K, sink);
        else if (amode == 0) k<0, false><<<grid, 256, 0, st>>>(wb[i], xb[i], M, K, sink);
        else k<1, false><<<grid, 256, 0, st>>>(wb[i], xb[i], M, K, sink);
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    return best * 1e3 / NB;
  };
  printf("per launch (chain of %d, incl. the launch boundary):\n", NB);
  printf("  activations only, row-major fragments : %.2f us\n", run(0, false));
  printf("  activations only, fragment-major      : %.2f us\n", run(1, false));
  printf("  weights + row-major fragment rows     : %.2f us\n", run(0, true));
  printf("  weights + fragment-major rows         : %.2f us\n", run(1, true));
  return 0;
}
