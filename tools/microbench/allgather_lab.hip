// Feasibility lab for a persistent DiffLoss head: how long does one all-gather phase take between the workgroups of one launch?
// G groups x P producers (one workgroup per CU).  Every phase each workgroup publishes a TILE-byte tile (16-byte sc1 stores, drain,
// workgroup barrier, one relaxed agent-scope flag store = epoch) and gathers the P tiles of its group (one wave polls the P flags with
// a single 64-lane sc1 load per try, bounded; then every wave reads the payload with sc1 16-byte loads).  Payload buffers are
// double-buffered by phase parity.  Checks every gathered word.  cdna_hip_programming.md Guideline 16, form R1 / table row 1.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/allgather_lab.hip -o tools/microbench/bin/allgather_lab
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

template <int P, int TILE>   // producers per group, bytes per published tile
__global__ __launch_bounds__(256) void allgather_kernel(u32x4* __restrict__ pay, unsigned* __restrict__ flags, unsigned* __restrict__ err, int phases,
                                                        unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // pads the workgroup to one per CU
  __shared__ int ok_sm;
  const int wg = blockIdx.x, grp = wg / P, me = wg % P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CH = TILE / 16;                 // 16-byte chunks per tile
  unsigned check = 0;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(pay, 0, (int)(2 * gridDim.x * TILE), 0x00020000);
  for (int ph = 0; ph < phases; ++ph) {
    const unsigned epoch = (unsigned)ph + 1u;
    // ---- publish: value encodes (phase, producer, chunk); 16-byte write-through (sc1) stores ----
    if (tid < CH) {
      const unsigned v = (epoch << 16) ^ ((unsigned)wg << 8) ^ (unsigned)tid;
      u32x4 val = {v, v + 1, v + 2, v + 3};
      __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (int)(((size_t)(ph & 1) * gridDim.x + wg) * TILE + tid * 16), 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + (size_t)(ph & 1) * gridDim.x + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- consume: wave 0 polls the group's P flags (one lane per producer), bounded ----
    if (wave == 0) {
      const unsigned* f = flags + (size_t)(ph & 1) * gridDim.x + (size_t)grp * P;
      bool done = false;
      for (int spin = 0; spin < (1 << 20); ++spin) {
        bool mine = true;
        for (int j = lane; j < P; j += 64) mine = mine && (__hip_atomic_load(f + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
        if (__all(mine)) {
          done = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) ok_sm = done ? 1 : 0;
    }
    __syncthreads();
    if (!ok_sm) {
      if (tid == 0) atomicAdd(err, 1u);
      return;   // give up: never hang
    }
    // payload: every load sc1 (bypasses this CU's L1)
    constexpr int NL = P * CH / 256;
    u32x4 got[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i)
      got[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((size_t)(ph & 1) * gridDim.x + (size_t)grp * P) * TILE + (tid + 256 * i) * 16), 0, 16);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int c = tid + 256 * i;
      const u32x4 v = got[i];
      const int prod = grp * P + c / CH, chunk = c % CH;
      const unsigned want = (epoch << 16) ^ ((unsigned)prod << 8) ^ (unsigned)chunk;
      if (v[0] != want || v[3] != want + 3) atomicAdd(err + 1, 1u);
      check += v[1];
    }
    __syncthreads();   // the tile buffers of this parity are free again two phases on: see the header of the lab
  }
  if (check == 0x12345u) sink[tid] = check;
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 600;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned *flags, *err, *sink;
  u32x4* pay;
  const int maxwg = 256;
  CK(hipMalloc(&flags, 2 * maxwg * sizeof(unsigned)));
  CK(hipMalloc(&err, 16));
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&pay, (size_t)2 * maxwg * 1024));
  auto run = [&](const char* name, auto kern, int grid, size_t ldsb) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    unsigned herr[2] = {0, 0};
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemsetAsync(flags, 0, 2 * maxwg * sizeof(unsigned), st));
      CK(hipMemsetAsync(err, 0, 16, st));
      CK(hipEventRecord(e0, st));
      kern<<<grid, 256, ldsb, st>>>(pay, flags, err, phases, sink);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
      if (herr[0] || herr[1]) break;
    }
    printf("%-44s grid %3d: %8.2f us per phase  (timeouts %u, wrong words %u)\n", name, grid, best * 1e3f / phases, herr[0], herr[1]);
  };
  run("2 groups x 64 producers, 512 B tiles", allgather_kernel<64, 512>, 128, 100 * 1024);
  run("4 groups x 64 producers, 256 B tiles", allgather_kernel<64, 256>, 256, 100 * 1024);
  run("1 group x 64 producers, 1024 B tiles", allgather_kernel<64, 1024>, 64, 100 * 1024);
  run("8 groups x 32 producers, 256 B tiles", allgather_kernel<32, 256>, 256, 100 * 1024);
  return 0;
}
