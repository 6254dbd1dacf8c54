// Kernels shared by the VQ-16 image decoder and the CausalVideoVAE decoder (gfx950).
// Activations are channels-last [B, T, H, W, C] (T = 1 for images) in the handle dtype.
#pragma once
#include "common.h"

namespace vlg {

struct ConvDesc {
  int B, Ti, Hi, Wi, Cin;   // input (before the optional nearest 2x upsample)
  int To, Ho, Wo, Cout;     // output
  int kt, kh, kw;           // kernel; zero pad kh/2, kw/2 in H/W; causal replicate pad (kt-1) frames in front (conv.py:124-130)
  int up;                   // 1: nearest 2x on H and W applied to the input first (vq_model.py:375; updownsample.py:146-153)
  int tmode = 0;            // time padding: 0 = causal replicate (CausalConv3d), 1 = symmetric zero pad (SamePadConv3d, vqvae.py:276-296)
  int sh = 1;               // spatial stride (1 or 2; encoders' Downsample / SpatialDownsample2x)
  int ph0 = -1, pw0 = -1;   // leading zero pad in H / W; -1 = kh/2, kw/2 ("same").  Downsample pads (0,1): ph0 = pw0 = 0
};

// out = conv(in) + bias (+ residual).  w: [Cout][taps][Cin] (re-laid out at load time), bias fp32 [Cout].
// out_cl (channels-last, dtype T) or out_planar (fp32 [B, Cout, To, Ho, Wo], the API output layout); one must be null.
// gn_part (optional): the GroupNorm (32 groups) statistics of the OUTPUT are accumulated in the epilogue (halo-tile path, channels-last output)
// as per-tile partial sums [B][*gn_nblk][32][{sum, sumsq}] doubles, fixed order, no atomics; *gn_nblk = 0 when the launch that ran cannot
// provide them (the caller then lets group_norm() make its own statistics pass).  Buffer: conv_gn_part_doubles(d) doubles.
template <typename T>
int conv_forward(const ConvDesc& d, const T* in, const T* w, const float* bias, const T* residual, T* out_cl,
                 float* out_planar, hipStream_t st, double* gn_part = nullptr, int* gn_nblk = nullptr);
size_t conv_gn_part_doubles(const ConvDesc& d);

// HIP-event bracket around every halo-tile (MFMA) conv launch on its own stream; read() waits for the events and returns the sums
int conv_timing_enable(bool on);
int conv_timing_read(double* ms_sum, double* flop_sum, long long* launches);

// GroupNorm(32 groups, eps) statistics + apply (+ swish) on channels-last data (vq_model.py:354-364, normalize.py:14-17)
// stats: scratch of group_norm_scratch_bytes(B, P) bytes.  Statistics are reduced in a fixed order (no atomics): deterministic.
constexpr int kGnPosPerBlock = 512;
size_t group_norm_scratch_bytes(int B, long long P);
template <typename T>
int group_norm(const T* x, T* y, const float* gamma, const float* beta, double* stats, int B, long long P, int C, float eps,
               bool swish, hipStream_t st, const double* given_part = nullptr, int given_nblk = 0);

// single-head spatial self-attention per frame (vq_model.py:335-347, attention.py:60-70): q,k,v,out [NF][HW][C], scale C^-0.5
template <typename T>
int spatial_attention(const T* q, const T* k, const T* v, T* out, int NF, int HW, int C, hipStream_t st);

// Q12 reinterpretation of AttnBlock3D (attention.py:60-63,69): channels-last view of the reference's
// [b,c,t,h,w] -> [b*t, c, h*w] reshape.  inverse = false: x[b][t][hw][c] -> y[b][t'][hw][c'];  inverse = true: back.
template <typename T>
int q12_permute(const T* x, T* y, int B, int T_, int HW, int C, bool inverse, hipStream_t st);

// TimeUpsample2x (updownsample.py:189-194): [B,T,HW,C] -> [B,2T-1,HW,C]
template <typename T>
int time_upsample2x(const T* x, T* y, int B, int T_, long long HWC, hipStream_t st);

// TimeDownsample2x (updownsample.py:163-180): replicate frame 0 twice in front, AvgPool3d((3,1,1), stride (2,1,1)): T -> (T-1)/2 + 1
template <typename T>
int time_downsample2x(const T* x, T* y, int B, int T_, long long HWC, hipStream_t st);
// [B,P,C] channels-last T -> planar fp32 [B,C,P]
template <typename T>
int cl_to_planar_f32(const T* x, float* y, int B, int C, long long P, hipStream_t st);

// ---- tokenizer_video VQ-VAE decoder pieces (tokenizer/tokenizer_video/vqvae.py, attention.py) ----------------------------
// y = relu(batchnorm_eval(x)) on channels-last data; rm/rv/gamma/beta fp32 [C], eps 1e-5
template <typename T>
int bn_relu(const T* x, T* y, const float* gamma, const float* beta, const float* rm, const float* rv, long long n_pos, int C, bool relu,
            hipStream_t st);
// SamePadConvTranspose3d(k=4, stride 2) (vqvae.py:299-319): in [B,T,H,W,Cin] -> out [B,2T,2H,2W,Cout]; w [Cout][64][Cin]
template <typename T>
int conv_transpose_k4s2(const T* in, const T* w, const float* bias, T* out_cl, float* out_planar, int B, int Ti, int Hi, int Wi, int Cin,
                        int Cout, bool relu, hipStream_t st);
// axial multi-head attention along one of the (t,h,w) axes (attention.py:228-247): q,k,v,out [B,T,H,W,nh*dk]
template <typename T>
int axial_attention(const T* q, const T* k, const T* v, T* out, int B, int T_, int H, int W, int nh, int dk, int axis, hipStream_t st);
// out = r + a + b + c (elementwise)
template <typename T>
int add4(const T* r, const T* a, const T* b, const T* c, T* out, long long n, hipStream_t st);
// ConvTranspose3d weight [Cin, Cout, taps] -> [Cout][taps][Cin]
template <typename T>
int relayout_convt_weight(const float* src, T* dst, int Cin, int Cout, int taps, hipStream_t st);

// layout / dtype glue
template <typename T>
int planar_f32_to_cl(const float* x, T* y, int B, int C, long long P, hipStream_t st);   // [B,C,P] fp32 -> [B,P,C] T
// conv weight [Cout, Cin, taps] (reference layout, any float dtype already converted to fp32) -> [Cout][taps][Cin] T
template <typename T>
int relayout_conv_weight(const float* src, T* dst, int Cout, int Cin, int taps, hipStream_t st);
// codebook lookup vq_model.py:261-276: out[b][pos][:] = normalize(E)[codes[b][pos]] (channels-last), E fp32 [n_e, e_dim]
template <typename T>
int codebook_lookup(const float* E, const int32_t* codes, T* out, long long n, int n_e, int e_dim, bool l2norm, hipStream_t st);

// nearest-neighbour search: idx[i] = argmin_j |z_i|^2 + |e_j|^2 - 2 z_i.e_j  (first minimum).
// l2norm: both sides row-normalised first (vq_model.py:221-232); z row i at z + i*z_stride_row + c*z_stride_c
int codebook_argmin(const float* z, long long z_stride_row, long long z_stride_c, long long rows_per_batch, long long z_stride_batch,
                    const float* E, long long n, int n_e, int dim, bool l2norm, int32_t* idx, hipStream_t st);

// the same search on the matrix cores (exact-fp32 MFMA) for the video codebook's shapes (csrc/codebook.hip); ee_scratch: n_e floats
bool codebook_mfma_ok(int n_e, int dim);
int codebook_argmin_mfma(const float* z, long long z_stride_row, long long z_stride_c, long long rows_per_batch, long long z_stride_batch,
                         const float* E, float* ee_scratch, long long n, int n_e, int dim, int32_t* idx, hipStream_t st);

}  // namespace vlg
