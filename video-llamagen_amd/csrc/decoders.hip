// placeholder entry points for the VQ / CausalVideoVAE decoders (replaced by the real implementation)
#include "common.h"
using namespace vlg;
#define NOTYET(name) set_error(name ": not implemented in this build"); return VLG_ERR_UNSUPPORTED
extern "C" {
int vlg_vq_create(const vlg_vq_config*, vlg_vq_t**) { NOTYET("vlg_vq_create"); }
int vlg_vq_destroy(vlg_vq_t*) { return VLG_OK; }
int vlg_vq_load_tensor(vlg_vq_t*, const char*, const void*, const int64_t*, int32_t, int32_t, int32_t, int32_t*) { NOTYET("vlg_vq_load_tensor"); }
int vlg_vq_decode_code(vlg_vq_t*, const int32_t*, int32_t, int32_t, int32_t, float*, void*) { NOTYET("vlg_vq_decode_code"); }
int vlg_vq_argmin(vlg_vq_t*, const float*, int32_t, int32_t, int32_t, int32_t*, void*) { NOTYET("vlg_vq_argmin"); }
int vlg_codebook_argmin(const float*, const float*, int32_t, int32_t, int32_t, int32_t*, void*) { NOTYET("vlg_codebook_argmin"); }
int vlg_vae_create(const vlg_vae_config*, vlg_vae_t**) { NOTYET("vlg_vae_create"); }
int vlg_vae_destroy(vlg_vae_t*) { return VLG_OK; }
int vlg_vae_load_tensor(vlg_vae_t*, const char*, const void*, const int64_t*, int32_t, int32_t, int32_t, int32_t*) { NOTYET("vlg_vae_load_tensor"); }
int vlg_vae_decode(vlg_vae_t*, const float*, int32_t, int32_t, int32_t, int32_t, float*, void*) { NOTYET("vlg_vae_decode"); }
int vlg_vae_out_shape(vlg_vae_t*, int32_t, int32_t, int32_t, int32_t*, int32_t*, int32_t*) { NOTYET("vlg_vae_out_shape"); }
}
