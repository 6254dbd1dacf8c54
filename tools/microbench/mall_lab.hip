// Lab: can the 256 MB Infinity Cache (memory-side, "MALL") be filled with the NEXT layer's K/V while a latency-bound phase leaves HBM idle,
// and does the following HBM-bound read then run faster than the HBM rate?
//
// Per "layer": stream_read(buf[l % NB], bytes) on stream 0 (stand-in for the decode attention: every byte once, non-temporal, 2560 workgroups),
// then an HBM-idle gap of `gap_us` (stand-in for the decode GEMM phase: one spinning workgroup).  With prefetch: when stream_read(l) has
// finished, stream 1 runs prefetch(buf[(l + 1) % NB], pf_bytes) - plain loads whose values are dropped - beside the gap.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mall_lab.hip -o tools/microbench/bin/mall_lab && mall_lab [MB per layer] [gap us]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) * gptr16;

template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const char* p, size_t bytes, unsigned* sink) {
  const size_t n16 = bytes / 16;
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (NT)
        v[u] = __builtin_nontemporal_load((gptr16)(uintptr_t)(p + (i + u * stride) * 16));
      else
        v[u] = *(gptr16)(uintptr_t)(p + (i + u * stride) * 16);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
  }
  for (; i < n16; i += stride) acc ^= (*(gptr16)(uintptr_t)(p + i * 16))[1];
  if (acc == 0x12345679u) sink[0] = acc;
}

__global__ void gap_kernel(long long ticks) {   // wall_clock64: 100 MHz
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atol(argv[1]) : 440;
  const int gap_us = argc > 2 ? atoi(argv[2]) : 35;
  const int NB = 8, layers = 72;
  const size_t bytes = mb << 20;
  std::vector<char*> buf(NB);
  for (int i = 0; i < NB; ++i) {
    CK(hipMalloc(&buf[i], bytes));
    CK(hipMemset(buf[i], i + 1, bytes));
  }
  unsigned* sink;
  CK(hipMalloc(&sink, 64));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(layers), evp(layers);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : evp) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%zu MB per layer, gap %d us, %d buffers\n", mb, gap_us, NB);
  // how fast is a buffer that was just read (temporal or non-temporal), by size?  (does a read allocate in the Infinity Cache at all?)
  for (size_t hmb : {(size_t)32, (size_t)64, (size_t)128, (size_t)192, (size_t)256, (size_t)440}) {
    if (hmb > mb) continue;
    for (int mode = 0; mode < 3; ++mode) {   // 0: NT reads after NT reads, 1: NT reads after plain reads, 2: plain after plain
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        read_kernel<true><<<2560, 256, 0, s0>>>(buf[1], bytes, sink);   // evict
        if (mode == 0) read_kernel<true><<<2560, 256, 0, s0>>>(buf[0], hmb << 20, sink);
        else read_kernel<false><<<2560, 256, 0, s0>>>(buf[0], hmb << 20, sink);
        CK(hipEventRecord(e0, s0));
        if (mode == 2) read_kernel<false><<<2560, 256, 0, s0>>>(buf[0], hmb << 20, sink);
        else read_kernel<true><<<2560, 256, 0, s0>>>(buf[0], hmb << 20, sink);
        CK(hipEventRecord(e1, s0));
        CK(hipStreamSynchronize(s0));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("re-read %3zu MB mode %d: %.1f us = %.2f TB/s\n", hmb, mode, best * 1e3, (hmb << 20) / (best * 1e-3) / 1e12);
    }
  }
  for (int pf_grid : {256, 512, 1024}) {
    for (size_t pf_mb : {(size_t)0, (size_t)64, (size_t)128, (size_t)192, (size_t)256}) {
      if (pf_mb > mb) continue;
      if (pf_mb == 0 && pf_grid != 256) continue;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s0));
        for (int l = 0; l < layers; ++l) {
          if (pf_mb && l > 0) CK(hipStreamWaitEvent(s0, evp[l - 1], 0));   // the prefetch of this buffer is over (it always is: it ran beside the gap)
          read_kernel<true><<<2560, 256, 0, s0>>>(buf[l % NB], bytes, sink);
          if (pf_mb) {
            CK(hipEventRecord(ev[l], s0));
            CK(hipStreamWaitEvent(s1, ev[l], 0));
            read_kernel<false><<<pf_grid, 256, 0, s1>>>(buf[(l + 1) % NB], pf_mb << 20, sink);
            CK(hipEventRecord(evp[l], s1));
          }
          gap_kernel<<<1, 64, 0, s0>>>((long long)gap_us * 100);
        }
        CK(hipEventRecord(e1, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double per = best * 1e3 / layers;
      printf("prefetch %3zu MB (grid %4d): %.1f us per layer -> read phase %.1f us = %.2f TB/s effective\n", pf_mb, pf_grid, per, per - gap_us,
             bytes / ((per - gap_us) * 1e-6) / 1e12);
    }
  }
  return 0;
}
