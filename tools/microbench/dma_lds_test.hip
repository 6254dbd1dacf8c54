// Does global_load_lds_dwordx4 (inline-asm form, M0 = wave-uniform LDS byte address) reach LDS offsets beyond 64 KB on gfx950 (160 KB LDS)?
// One workgroup, wave 1 DMA-copies a 1 KB block (lane-linear) to each probed LDS offset, waits vmcnt(0), barrier, wave 0 reads it back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(128) void k(const unsigned* src, unsigned* out, const unsigned* offs, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 159 * 1024 / 4; i += 128) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
  __syncthreads();
  if (wave == 1) {
    for (int j = 0; j < n; ++j) {
      const unsigned base = (unsigned)(uintptr_t)smem;   // LDS byte address of the dynamic segment (0 here)
      const unsigned dst = __builtin_amdgcn_readfirstlane(base + offs[j]);
      glds16(src + (size_t)j * 256 + lane * 4, dst);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (wave == 0)
    for (int j = 0; j < n; ++j) {
      u32x4 v = *reinterpret_cast<u32x4*>(smem + offs[j] + lane * 16);
      *reinterpret_cast<u32x4*>(out + (size_t)j * 256 + lane * 4) = v;
    }
}
int main() {
  std::vector<unsigned> offs = {0, 1024, 60 * 1024, 65 * 1024, 70 * 1024, 100 * 1024, 130 * 1024, 158 * 1024};
  const int n = (int)offs.size();
  std::vector<unsigned> h(n * 256), o(n * 256, 0);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (unsigned)i;
  unsigned *ds, *dout, *doffs;
  hipMalloc(&ds, h.size() * 4); hipMalloc(&dout, h.size() * 4); hipMalloc(&doffs, n * 4);
  hipMemcpy(ds, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(doffs, offs.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  k<<<1, 128, 159 * 1024>>>(ds, dout, doffs, n);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  for (int j = 0; j < n; ++j) {
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += o[j * 256 + i] != h[j * 256 + i];
    printf("offset %6u: %s (first word 0x%08x, want 0x%08x)\n", offs[j], bad ? "MISMATCH" : "ok", o[j * 256], h[j * 256]);
  }
  return 0;
}
