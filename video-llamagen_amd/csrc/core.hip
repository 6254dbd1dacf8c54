// libvlg core: error text, dtype conversion on upload, version.
#include <stdarg.h>

#include "common.h"

namespace vlg {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

template <typename TS, typename TD>
__global__ void convert_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) DT<TD>::st(dst + i, DT<TS>::ld(src + i));
}

int upload_convert(void* dst, int dst_dtype, const void* data, int src_dtype, int src_on_device, int64_t n, hipStream_t stream) {
  if (n <= 0) return VLG_OK;
  const size_t sbytes = (size_t)n * dtype_size(src_dtype);
  const void* dsrc = data;
  DevBuf staging;
  // a device source may still be being produced on a stream this library does not know: loading is not hot, order against everything
  if (src_on_device) VLG_HIP(hipDeviceSynchronize());
  if (!src_on_device) {
    if (src_dtype == dst_dtype) {
      VLG_HIP(hipMemcpy(dst, data, sbytes, hipMemcpyHostToDevice));
      return VLG_OK;
    }
    VLG_TRY(staging.reserve(sbytes));
    VLG_HIP(hipMemcpy(staging.p, data, sbytes, hipMemcpyHostToDevice));
    dsrc = staging.p;
  } else if (src_dtype == dst_dtype) {
    VLG_HIP(hipMemcpyAsync(dst, data, sbytes, hipMemcpyDeviceToDevice, stream));
    VLG_HIP(hipStreamSynchronize(stream));
    return VLG_OK;
  }
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (src_dtype == VLG_F32 && dst_dtype == VLG_BF16)
    convert_kernel<float, bf16><<<blocks, 256, 0, stream>>>((const float*)dsrc, (bf16*)dst, n);
  else if (src_dtype == VLG_BF16 && dst_dtype == VLG_F32)
    convert_kernel<bf16, float><<<blocks, 256, 0, stream>>>((const bf16*)dsrc, (float*)dst, n);
  else {
    set_error("upload_convert: bad dtype pair %d -> %d", src_dtype, dst_dtype);
    return VLG_ERR_BAD_ARG;
  }
  VLG_HIP(hipStreamSynchronize(stream));  // staging goes out of scope
  return VLG_OK;
}

}  // namespace vlg

extern "C" {
const char* vlg_last_error(void) { return vlg::g_err; }
void vlg_version(int* out3) {
  if (!out3) return;
  out3[0] = 0;
  out3[1] = 1;
  out3[2] = 950;
}
}
