// Shared device helpers of the persistent decode kernels (pdecode.hip, pdecode2.hip): vector types, DPP reductions, the LDS-only
// workgroup barrier, the MFMA wrapper.  Internal linkage: every translation unit gets its own copy.
#pragma once
#include <type_traits>

#include "gpt_kernels.h"

namespace vlg {
namespace {

typedef __bf16 pd_bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 pd_bf2_t __attribute__((ext_vector_type(2)));
typedef float pd_f2_t __attribute__((ext_vector_type(2)));
typedef float pd_f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned pd_u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned pd_u32x2_t __attribute__((ext_vector_type(2)));
// explicit address spaces: a pointer read out of a struct is GENERIC to hipcc and loads through it become flat_load, which counts in
// lgkmcnt as well as vmcnt - every LDS wait behind a weight prefetch would then wait for the HBM stream
typedef const pd_u32x4_t __attribute__((address_space(1))) * pd_gptr16;
typedef const pd_u32x2_t __attribute__((address_space(1))) * pd_gptr8;
// the layer table is never written while the kernel runs: constant address space -> scalar loads (no vector load + vmcnt(0) per pointer)
typedef const PdLayer __attribute__((address_space(4))) * pd_layer_cptr;

constexpr int PD_NW = 8;           // waves per workgroup (all of them compute; K steps are dealt to them round-robin)
constexpr int PD_NTHR = PD_NW * 64;
constexpr int PD_NF = 10;          // weight fragments (16 bytes per lane = one 16-column x 64-byte-of-K tile per wave) per register set
constexpr int PD_UB = 8;           // granule-pair loads in flight per thread in a sweep
constexpr int PD_MAXNS = 8;        // attention KV splits
constexpr int PD_MAXKS = 8;        // K slices of a wo / w2 tile

template <typename T, int VEC>
struct alignas((VEC * sizeof(T)) > 16 ? 16 : (VEC * sizeof(T))) PdPack {
  T v[VEC];
};
template <typename T, int VEC>
__device__ __forceinline__ PdPack<T, VEC> pd_load_stream(const char* p) {
  static_assert(sizeof(PdPack<T, VEC>) == 16 || sizeof(PdPack<T, VEC>) == 8, "16- or 8-byte K/V chunks");
  if constexpr (sizeof(PdPack<T, VEC>) == 16)
    return __builtin_bit_cast(PdPack<T, VEC>, __builtin_nontemporal_load((pd_gptr16)(uintptr_t)p));
  else
    return __builtin_bit_cast(PdPack<T, VEC>, __builtin_nontemporal_load((pd_gptr8)(uintptr_t)p));
}
template <int CTRL>
__device__ __forceinline__ float pd_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float pd_group_sum(float s) {
  if (N >= 2) s += pd_dpp<0xB1>(s);
  if (N >= 4) s += pd_dpp<0x4E>(s);
  if (N >= 8) s += pd_dpp<0x141>(s);
  if (N >= 16) s += pd_dpp<0x140>(s);
  if (N >= 32) s += __shfl_xor(s, 16);
  if (N >= 64) s += __shfl_xor(s, 32);
  return s;
}
// wave-wide sum on the DPP network only (ds_bpermute-based shuffles cost several times more latency); the total comes back to every lane
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float pd_dpp_rm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float pd_wave_sum(float v) {
  v += pd_dpp_rm<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += pd_dpp_rm<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += pd_dpp_rm<0x141, 0xf>(v);   // row_half_mirror
  v += pd_dpp_rm<0x140, 0xf>(v);   // row_mirror: every lane holds the sum of its row of 16
  v += pd_dpp_rm<0x142, 0xa>(v);   // row_bcast15 into rows 1, 3
  v += pd_dpp_rm<0x143, 0xc>(v);   // row_bcast31 into rows 2, 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float pd_silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ unsigned pd_pack2(float a, float b) {   // two RNE bf16 (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(pd_f2_t{a, b}, pd_bf2_t));
}
__device__ __forceinline__ float pd_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float pd_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() carries a workgroup-scope release, which on gfx950 (one counter for
// vector loads and stores) drains vmcnt: placed behind a weight prefetch it would stall every wave until the HBM stream has landed.
__device__ __forceinline__ void pd_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <typename T>
__device__ __forceinline__ void pd_mfma(const pd_u32x4_t& a, const pd_u32x4_t& b, pd_f32x4_t& acc) {
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pd_bf16x8_t, a), __builtin_bit_cast(pd_bf16x8_t, b), acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
  }
}

}  // namespace
}  // namespace vlg
