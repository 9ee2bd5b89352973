/*
 * cer_hip.h -- C-ABI of libcer_hip.so, the MI355X (gfx950) hot path for
 * feature-based compound emotion recognition.
 *
 * The reference (sbelharbi/feature-vs-text-compound-emotion) is pure Python /
 * torch.nn and has NO native interface.  Each entry point below therefore
 * replaces the stock torch op(s) that the reference executes at the cited
 * file:line; the Python host in feature_vs_text_compound_emotion_amd/ binds
 * them with ctypes and mirrors the reference's nn.Module surface on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors keep
 *     them alive); the library never allocates, frees or retains them;
 *   - activations are channels-last: images [N,H,W,C], sequences [B,L,C];
 *   - all launches are asynchronous on the `stream` argument (a hipStream_t
 *     passed as void*; NULL = the default stream); no hidden device sync;
 *   - every function returns 0 on success, a negative cer_status otherwise;
 *     cer_last_error() returns a thread-local description.  No exception or
 *     abort crosses the boundary.
 */
#ifndef CER_HIP_H
#define CER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum cer_status {
    CER_OK = 0,
    CER_ERR_INVALID_ARG = -1,
    CER_ERR_UNSUPPORTED = -2,
    CER_ERR_HIP = -3,
    CER_ERR_WORKSPACE = -4
};

enum cer_act { CER_ACT_NONE = 0, CER_ACT_PRELU = 1, CER_ACT_LEAKY = 2, CER_ACT_RELU = 3, CER_ACT_GELU = 4 };

const char *cer_last_error(void);
int cer_version(void);

/* ------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer on the fp32 matrix cores.
 *
 *   y = act2( mask * act1( conv(affine_in(x), w) + bias ) + residual )
 *
 * Replaces nn.Conv2d + BatchNorm2d + PReLU + residual add
 * (reference models/arcface_model.py:44-60, :130-132), nn.Conv2d + ReLU
 * (models/backbone.py:41-52), the causal dilated nn.Conv1d + Chomp1d +
 * LeakyReLU + residual of models/temporal_convolutional_model.py:21-56, and
 * every nn.Linear on the path (a 1x1 "conv" on an [M,1,1,K] image).
 *
 *   x        [N,H,W,Cin] (or [N,Cin,H,W] when x_nchw != 0; small-Cin path only)
 *   w        [Cout][Kpad], K index = (kh*KW + kw)*Cin + c, zero padded to a
 *            multiple of 32 (cer_conv_kpad)
 *   in_scale/in_shift  [Cin] or NULL: per-channel affine applied to in-bounds
 *            input pixels only (zero padding stays zero) -- the pre-conv
 *            BatchNorm of bottleneck_IR in eval mode
 *   bias     [Cout] or NULL;  alpha [Cout] PReLU slopes (act1 == CER_ACT_PRELU)
 *   residual [N,Hr,Wr,Cout] or NULL, sampled at (ho*res_stride, wo*res_stride)
 *            (MaxPool2d(1,stride) shortcut == spatial subsample)
 *   mask     [N,Ho,Wo,Cout] or NULL (pre-scaled dropout mask)
 *   split_k  >= 1; > 1 needs `workspace` of cer_conv2d_workspace_bytes()
 * ---------------------------------------------------------------------- */
typedef struct cer_conv_desc {
    int32_t N, H, W, Cin;
    int32_t Ho, Wo, Cout;
    int32_t KH, KW, stride, dil_h, dil_w, pad_t, pad_l;
    int32_t x_nchw;
    int32_t res_stride, Hr, Wr;
    int32_t act1, act2;
    float slope;          /* LeakyReLU slope */
    int32_t split_k;
    int32_t tile;         /* 0 = auto; else forces a tile config (testing / tuning) */
} cer_conv_desc;

int cer_conv_kpad(int KH, int KW, int Cin);
size_t cer_conv2d_workspace_bytes(const cer_conv_desc *d);
int cer_conv2d_fwd(const cer_conv_desc *d, const float *x, const float *w,
                   const float *in_scale, const float *in_shift,
                   const float *bias, const float *alpha,
                   const float *residual, const float *mask,
                   float *y, void *workspace, size_t workspace_bytes, void *stream);

/* Pack an OIHW (torch) conv weight into [Cout][Kpad] with optional per-output
 * scale (BatchNorm fold).  w_oihw [Cout,Cin,KH,KW]; out_scale [Cout] or NULL. */
int cer_pack_conv_weight(const float *w_oihw, const float *out_scale, float *w_packed,
                         int Cout, int Cin, int KH, int KW, int flip, void *stream);

/* rows x / ||x||_2, no epsilon (reference models/arcface_model.py:17-20). */
int cer_l2norm_rows(const float *x, float *y, int rows, int cols, void *stream);

/* max-pool 2x2 stride 2 on NHWC (reference models/backbone.py:45-46). */
int cer_maxpool2x2_nhwc(const float *x, float *y, int N, int H, int W, int C, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CER_HIP_H */
