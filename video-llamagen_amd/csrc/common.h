// Shared host/device helpers for libvlg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <initializer_list>
#include <map>
#include <string>
#include <vector>

#include "../../include/vlg.h"

namespace vlg {

void set_error(const char* fmt, ...);

#define VLG_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      vlg::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return _e == hipErrorOutOfMemory ? VLG_ERR_OOM : VLG_ERR_HIP;                        \
    }                                                                                      \
  } while (0)

#define VLG_TRY(expr)            \
  do {                           \
    int _s = (expr);             \
    if (_s != VLG_OK) return _s; \
  } while (0)

#define VLG_CHECK(cond, code, ...)   \
  do {                               \
    if (!(cond)) {                   \
      vlg::set_error(__VA_ARGS__);   \
      return code;                   \
    }                                \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel's code object, and a code object is loaded per DEVICE: a process-wide
// "already set" flag leaves the second device of a process at the 64 KB default, and its launch then fails (or, unchecked, silently does
// nothing).  `done` is one static array per call site (= per kernel instantiation).
struct LdsAttrOnce {
  bool done[64] = {};
};
inline int set_max_dynamic_lds(LdsAttrOnce& once, std::initializer_list<const void*> kernels, int bytes) {
  int dev = 0;
  VLG_HIP(hipGetDevice(&dev));
  const bool track = dev >= 0 && dev < 64;
  if (track && once.done[dev]) return VLG_OK;
  for (const void* k : kernels) VLG_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (track) once.done[dev] = true;
  return VLG_OK;
}

// ---- storage dtypes -------------------------------------------------------------------------
struct bf16 {
  uint16_t v;
};

__host__ __device__ inline float bf16_to_f32(uint16_t b) {
  union {
    uint32_t u;
    float f;
  } c;
  c.u = (uint32_t)b << 16;
  return c.f;
}
// round-to-nearest-even; NaN stays NaN (MI355X_MICROARCH correctness table)
__host__ __device__ inline uint16_t f32_to_bf16(float f) {
  union {
    uint32_t u;
    float f;
  } c;
  c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((c.u >> 16) | 0x40);
  uint32_t r = ((c.u >> 16) & 1u) + 0x7fffu;
  return (uint16_t)((c.u + r) >> 16);
}

template <typename T>
struct DT;
template <>
struct DT<float> {
  static constexpr int code = VLG_F32;
  __host__ __device__ static inline float ld(const float* p) { return *p; }
  __host__ __device__ static inline void st(float* p, float v) { *p = v; }
  __host__ __device__ static inline float rt(float v) { return v; }  // round-trip through storage
};
template <>
struct DT<bf16> {
  static constexpr int code = VLG_BF16;
  __host__ __device__ static inline float ld(const bf16* p) { return bf16_to_f32(p->v); }
  __host__ __device__ static inline void st(bf16* p, float v) { p->v = f32_to_bf16(v); }
  __host__ __device__ static inline float rt(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

inline size_t dtype_size(int code) { return code == VLG_BF16 ? 2 : 4; }

// ---- device buffer ---------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  // grow-only
  int reserve(size_t n) {
    if (n <= bytes) return VLG_OK;
    release();
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
      p = nullptr;
      set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
      return VLG_ERR_OOM;
    }
    bytes = n;
    return VLG_OK;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// A named weight tensor in the handle dtype
struct Tensor {
  DevBuf buf;
  std::vector<int64_t> shape;
  bool loaded = false;
  DevBuf fm;             // fragment-major copy of a Linear weight (gpt_kernels.h: relayout_fragment_major), built lazily
  bool fm_stale = true;  // set by every (partial) load of the tensor
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

// converts + uploads `data` (host or device, fp32 or bf16) into dst (device, dst_dtype)
int upload_convert(void* dst, int dst_dtype, const void* data, int src_dtype, int src_on_device, int64_t n,
                   hipStream_t stream);

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

}  // namespace vlg
