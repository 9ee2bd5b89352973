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

/* Storage type of the 16-bit tensors of a launch (cer_conv_desc.storage and the *_n16 entry points):
 * 0 = none (fp32 tensors / split hi+lo bf16 pairs, as the pointers say); otherwise every 16-bit tensor is ONE plane of
 * bf16 or IEEE half -- the reference's own GPU recipe is fp16 autocast (trainer.py:14-15,341,367), BASELINE cfg5 asks
 * for "bf16 storage / fp32 accumulate". */
enum cer_storage { CER_STORE_NONE = 0, CER_STORE_BF16 = 1, CER_STORE_F16 = 2 };

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
 *   aux      [N,Ho,Wo,Cout] or NULL: receives mask*act1(conv+bias), the value before
 *            the residual add (saved for the backward pass of a TemporalBlock)
 *   stats    [cer_conv2d_stats_tiles(d, bf16x3)][2][Cout] or NULL: per-tile sum and sum of squares
 *            of the RAW conv result over valid pixels (deterministic partials for the
 *            train-mode BatchNorm that follows; reduce with cer_bn_finalize)
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
    int32_t x_ld, y_ld;   /* pitch in floats between pixels of x / rows of y; 0 = dense (Cin / Cout).
                             Lets a layer read or write a column slice of a wider [rows, ld] buffer. */
    int32_t storage;      /* cer_storage: narrow (single 16-bit plane) operands / outputs, see cer_conv_io */
    /* Space-to-depth hand-over between the two 3x3 convs of a stride-2 bottleneck_IR unit (models/arcface_model.py:38-41):
     *   y_s2d != 0: the 16-bit output planes (y_hi / y_lo) are stored as [N, Ho/2, Wo/2, 4*Cout] with
     *       ys[n][i][j][blk*Cout + c] = y[n][2i+py][2j+px][c], blk = 3 - 2*py - px -- same bytes, the stores permuted.
     *       Window / patch kernels only (3x3 / stride 1 / pad 1), even Ho and Wo, no fp32 / second output, no residual.
     *   x_s2d != 0: x_hi / x_lo are such a tensor standing for the [N, H, W, Cin] input of THIS 3x3 / stride 2 / pad 1
     *       conv (H, W even, Cin % 64 == 0 -- narrow storage: % 128 --, Wo <= 126, x_ld = 0), and the K columns of w are
     *       in the kernel's step order (cer_conv_s2d_k_order).  The conv then runs window-resident (conv_b3_s2d_kernel /
     *       conv_n16_s2d_kernel) instead of gathering nine tap tiles per channel chunk.
     *   Split (bf16x3) and narrow operands only. */
    int32_t x_s2d, y_s2d;
} cer_conv_desc;

int cer_conv_kpad(int KH, int KW, int Cin);
/* K-column order of the weights of an x_s2d conv: order[j] (j < 9*Cin) = the column (kh*3 + kw)*Cin + c of the ordinary
 * [Cout][9*Cin] layout that column j of the permuted matrix holds.  chunk = 32 (bf16x3 operands) or 64 (narrow storage):
 * the channels of one kernel step; Cin % (2*chunk) == 0.  Host array of 9*Cin ints. */
int cer_conv_s2d_k_order(int Cin, int chunk, int32_t *order);
size_t cer_conv2d_workspace_bytes(const cer_conv_desc *d);
/* rows of `stats` the launch will write; kernel_family: 0 = fp32 kernel, 1 = bf16x3 (split operands), 2 = narrow (n16) */
int cer_conv2d_stats_tiles(const cer_conv_desc *d, int kernel_family);
int cer_conv2d_fwd(const cer_conv_desc *d, const float *x, const float *w,
                   const float *in_scale, const float *in_shift,
                   const float *bias, const float *alpha,
                   const float *residual, const float *mask,
                   float *y, float *aux, float *stats, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------
 * Unified entry point: every operand and output of one conv / linear launch.
 *
 * bf16x3 mode (x_hi != NULL): operands are SPLIT tensors, value = hi + lo with hi = bf16(v),
 * lo = bf16(v - hi) (two bf16 planes, NHWC / [Cout][Kpad] like the fp32 layouts).  Products are
 * evaluated as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation: <= 2^-15 relative error per product
 * (|logit error| 1.3e-6 on the full model) at a 5.3x higher matrix ceiling than the fp32 MFMA path.
 * Outputs (any subset): y fp32; (y_hi, y_lo) split; (y2_hi, y2_lo) = split(out*s2[c]+t2[c]), the next
 * layer's eval-mode pre-conv BatchNorm applied by the producer (the zero padding of the consumer
 * then stays exactly zero).  The residual may be fp32 (`residual`) or split (`res_hi`, `res_lo`).
 *
 * Narrow mode (desc.storage = CER_STORE_BF16 / CER_STORE_F16): x_hi, w_hi, res_hi, y_hi are single planes of that type
 * and every *_lo pointer must be NULL; ONE v_mfma_f32_16x16x32_{bf16,f16} per 32-deep product, fp32 accumulate, fp32
 * epilogue (bias / bias9 / PReLU / residual / batch statistics), one rounding at the store.  This is what the
 * reference's --amp recipe computes (fp16 autocast) and BASELINE cfg5's "bf16 storage".  With fp32 operands
 * (x, w given; the Cin = 3 stem) and storage != 0 the fp32 kernel runs and y_hi receives the narrow plane.
 * ---------------------------------------------------------------------- */
typedef struct cer_conv_io {
    const float *x, *w;                       /* fp32 mode operands */
    const uint16_t *x_hi, *x_lo, *w_hi, *w_lo;  /* bf16x3 mode operands */
    const float *in_scale, *in_shift, *bias, *alpha, *residual, *mask;
    const uint16_t *res_hi, *res_lo;
    float *y, *aux, *stats;
    uint16_t *y_hi, *y_lo;
    const float *s2, *t2;
    uint16_t *y2_hi, *y2_lo;
    /* [9][Cout] border-dependent bias (replaces `bias`) of a stride-1 "same" conv whose INPUT BatchNorm was folded
     * into the weights: row 3*ry + rx with ry / rx = 0 on the first, 2 on the last, 1 on the other output rows /
     * columns -- the input shift only contributes through the taps that fall inside the image. */
    const float *bias9;
} cer_conv_io;

int cer_conv2d_run(const cer_conv_desc *d, const cer_conv_io *io, void *workspace, size_t workspace_bytes, void *stream);
/* The bf16x3 kernel variant cer_conv2d_run launches for this descriptor (desc.tile, or the automatic choice when it is 0):
 * 41/42/44/45 = conv_b3_dma16_kernel<128x128 / 128x64 / 64x128 / 64x64>, see csrc/conv_b3.hip.  0 = invalid. */
int cer_conv2d_b3_tile(const cer_conv_desc *d);
/* The narrow kernel variant for this descriptor: 61..67 = conv_n16_kernel tiles (csrc/conv_n16.hip).  0 = invalid. */
int cer_conv2d_n16_tile(const cer_conv_desc *d);

/* v' = v*scale[c]+shift[c] (channels-last; scale/shift may be NULL) -> one 16-bit plane of `storage` (bf16 / half),
 * round-to-nearest-even; and back (tests, fp32 consumers). */
int cer_to_n16(const float *x, const float *scale, const float *shift, int C, uint16_t *out, size_t n, int storage,
               void *stream);
int cer_from_n16(const uint16_t *x, float *out, size_t n, int storage, void *stream);

/* Elementwise passes of the released encoder units that hand a SPLIT tensor (hi / lo bf16 planes) straight to the matrix-core
 * kernels -- replace "torch op -> fp32 tensor -> cer_split_bf16" pairs in the backward of a bottleneck_IR unit
 * (reference models/arcface_model.py:44-60; PReLU :54, BatchNorm2d :53,57):
 *   cer_prelu_split        t = prelu(x) -> split
 *   cer_prelu_bwd_split    torch's PReLU backward; dx as fp32 (dx) and / or split (dx_hi, dx_lo); dalpha_terms as cer_prelu_bwd
 *   cer_bn_rows_bwd_split  cer_bn_rows_bwd (train mode, dense rows) with dx as a split tensor; workspace as cer_col_sum */
int cer_prelu_split(const float *x, const float *alpha, uint16_t *hi, uint16_t *lo, size_t rows, int C, void *stream);
int cer_prelu_bwd_split(const float *dy, const float *x, const float *alpha, float *dx, uint16_t *dx_hi, uint16_t *dx_lo,
                        float *dalpha_terms, size_t rows, int C, void *stream);
int cer_bn_rows_bwd_split(const float *dy, const float *x, const float *save_mean, const float *save_invstd, const float *w,
                          uint16_t *dx_hi, uint16_t *dx_lo, float *dw, float *db, int R, int C, void *workspace,
                          size_t workspace_bytes, void *stream);
/* cer_bn_rows_bwd (train mode, dense rows, C % 4 == 0) with an addend: dx = BatchNorm-backward(dy) + add (add may be NULL; it may
 * alias dx).  The gradient of a unit's input is the sum of its residual branch (through the unit's first BatchNorm) and its
 * shortcut branch (models/arcface_model.py:58-60): one pass instead of a BatchNorm-backward pass and an add pass. */
int cer_bn_rows_bwd_add(const float *dy, const float *x, const float *save_mean, const float *save_invstd, const float *w,
                        const float *add, float *dx, float *dw, float *db, int R, int C, void *workspace, size_t workspace_bytes,
                        void *stream);

/* v' = v*scale[c]+shift[c] (channels-last, C channels; scale/shift may be NULL) -> (bf16(v'), bf16(v' - bf16(v'))),
 * round-to-nearest-even on both parts. */
int cer_split_bf16(const float *x, const float *scale, const float *shift, int C, uint16_t *hi, uint16_t *lo, size_t n,
                   void *stream);

/* Pack an OIHW (torch) conv weight into [Cout][Kpad] with optional per-output
 * scale (BatchNorm fold).  w_oihw [Cout,Cin,KH,KW]; out_scale [Cout] or NULL.
 * transpose != 0 writes the data-gradient filter instead: [Cin][Kpad'] with
 * k = tap*Cout + o (taps flipped when flip != 0), so that
 * dX = cer_conv2d_fwd(dY, packed_T) with the padding mirrored. */
int cer_pack_conv_weight(const float *w_oihw, const float *out_scale, float *w_packed,
                         int Cout, int Cin, int KH, int KW, int flip, int transpose, void *stream);

/* rows x / ||x||_2, no epsilon (reference models/arcface_model.py:17-20). */
int cer_l2norm_rows(const float *x, float *y, int rows, int cols, void *stream);

/* Backward of cer_l2norm_rows: x is the forward INPUT; dx = (dy - y (y.dy)) / ||x|| with y = x/||x||.  Used when the
 * encoder head is released for training (reference base/parameter_control.py:55-103, first release group). */
int cer_l2norm_rows_bwd(const float *dy, const float *x, float *dx, int rows, int cols, void *stream);

/* max-pool 2x2 stride 2 on NHWC (reference models/backbone.py:45-46). */
/* ------------------------------------------------------------------------
 * IR-50 input layer: Conv2d(3, 64, 3x3, stride 1, pad 1) [+ BatchNorm2d + PReLU] (reference models/arcface_model.py:130-132,
 * :148) as a direct convolution on the vector ALUs -- the layer is write-bound (27 multiply-adds per output value), so in
 * model.train() it is cheaper to run it twice than to store its raw result for the batch-statistics BatchNorm:
 *   statistics pass (y, y_hi, y_lo, scale, shift, alpha all NULL): stats [cer_stem_conv3x3_stats_rows(N, H)][2][64] receives the
 *       per-block sum / sum of squares of the RAW conv result (reduce with cer_bn_finalize);
 *   apply pass (an output given): out = prelu(conv * scale[c] + shift[c], alpha[c]) (NULL scale / shift / alpha = 1 / 0 / none)
 *       stored as fp32 (y) and / or split bf16 planes (y_hi, y_lo; storage = 0) or ONE narrow plane (y_hi; storage =
 *       CER_STORE_BF16 / CER_STORE_F16); stats (optional) then receives the partial statistics of `out`.
 * x [N, 3, H, W] fp32 (NCHW frames as the reference feeds them); w [64][Kpad] packed like every conv weight
 * (K index = (kh*3 + kw)*3 + c, Kpad = cer_conv_kpad(3, 3, 3)); outputs NHWC [N, H, W, 64]. */
int cer_stem_conv3x3_stats_rows(int N, int H);
int cer_stem_conv3x3(const float *x, const float *w, int Kpad, const float *scale, const float *shift, const float *alpha,
                     float *y, uint16_t *y_hi, uint16_t *y_lo, int storage, float *stats, int N, int H, int W, void *stream);

int cer_maxpool2x2_nhwc(const float *x, float *y, int N, int H, int W, int C, void *stream);

/* Video-frame input transform of the reference Dataset, fused (base/dataset.py:487-508,
 * base/transforms3D.py:33-143): uint8 frames [n][H][W][3] -> GroupScale(size) (== PIL
 * Image.resize BILINEAR, Pillow's 8-bit fixed-point antialiased triangle filter, reproduced
 * bit for bit) -> crop (x1, y1) to crop x crop -> optional horizontal flip -> /255 -> (v - mean) / std,
 * written as fp32 [n][3][crop][crop] (and optionally the uint8 HWC image before the float stage).
 * hbounds/vbounds [size][2] = (first input index, tap count); hcoef/vcoef [size][ksize] = Pillow's
 * integer coefficients (2^22 scale), computed by the host.  crop_xyf [groups][3] = (x1, y1, flip), one
 * triple per `frames_per_group` consecutive frames (GroupRandomCrop / GroupRandomHorizontalFlip draw
 * once per clip).  max_band_rows = the largest number of input rows any band of cer_frames_band_rows()
 * output rows needs (sizes the LDS). */
int cer_frames_band_rows(void);
int cer_frames_transform(const uint8_t *frames, int n_frames, int H, int W, const int32_t *hbounds, const int32_t *hcoef,
                         int hksize, const int32_t *vbounds, const int32_t *vcoef, int vksize, int size, int crop,
                         const int32_t *crop_xyf, int frames_per_group, int max_band_rows, float mean, float stdv,
                         float *out, uint8_t *out_u8, void *stream);

/* ------------------------------------------------------------------------
 * Trainable tail (rows = B*L frames, channels-last).  Forward AND backward,
 * because this is the part of the model the reference actually trains
 * (models/model.py:432-433 freezes the encoders).
 * ---------------------------------------------------------------------- */

/* torch.nn.utils.weight_norm(dim=0): w = g*v/||v|| per output row of E = Cin*k
 * elements (reference models/temporal_convolutional_model.py:24,30).  norm [rows] is saved
 * for the backward, which returns dv [rows,E] and dg [rows]. */
int cer_weight_norm_fwd(const float *v, const float *g, float *w, float *norm, int rows, int E, void *stream);
int cer_weight_norm_bwd(const float *dw, const float *v, const float *g, const float *norm,
                        float *dv, float *dg, int rows, int E, void *stream);
/* The same forward for a [Cout][Cin][k] 1-D filter, writing in the same launch the two packed layouts the conv kernels read
 * (cer_pack_conv_weight's): wp_fwd [Cout][cer_conv_kpad(k,1,Cin)] and the flipped, transposed data-gradient filter
 * wp_dgrad [Cin][cer_conv_kpad(k,1,Cout)] -- the TCN (temporal_convolutional_model.py:24-38) re-normalises every step. */
int cer_weight_norm_fwd_packed(const float *v, const float *g, float *w, float *norm, float *wp_fwd, float *wp_dgrad, int Cout,
                               int Cin, int k, void *stream);
/* The backward on the weight-gradient kernel's split partial slabs [splits][rows * E] (summed in the order 0 .. splits-1). */
int cer_weight_norm_bwd_partials(const float *dw_parts, int splits, const float *v, const float *g, const float *norm, float *dv,
                                 float *dg, int rows, int E, void *stream);

/* Weight gradient of the causal dilated conv1d / linear layers:
 * dW[co][ci][j] = sum_r dZ[r][co] * X[r-(k-1-j)*dil][ci], rows never cross a length-L sequence.
 * dz [R,dz_ld], x [R,x_ld], dw [Cout,Cin,k] (torch layout).  Linear: k = 1. */
/* workspace (optional; both kernels below): with cer_conv_wgrad_workspace_bytes(R, Cout, Cin, k) bytes the rows are split over
 * blocks and the partial results added in a fixed order (the tail's layers have R = B * L = 1024 rows and few output tiles);
 * without it (NULL / too small) one block per tile and tap walks all rows. */
size_t cer_conv_wgrad_workspace_bytes(long long R, int Cout, int Cin, int k);
int cer_conv1d_wgrad(const float *dz, int dz_ld, const float *x, int x_ld, float *dw,
                     int R, int L, int Cout, int Cin, int k, int dil, void *workspace, size_t workspace_bytes, void *stream);
/* cer_conv1d_wgrad of a weight-normed conv followed by its weight-norm backward, in two launches: (dv, dg) instead of dW.
 * workspace: max(cer_conv_wgrad_workspace_bytes(R, Cout, Cin, k), Cout * Cin * k * 4) bytes. */
int cer_conv1d_wgrad_weight_norm_bwd(const float *dz, int dz_ld, const float *x, int x_ld, int R, int L, int Cout, int Cin, int k,
                                     int dil, const float *v, const float *g, const float *norm, float *dv, float *dg,
                                     void *workspace, size_t workspace_bytes, void *stream);

/* 2-D weight gradient (encoder units released for training, reference base/parameter_control.py:55-103):
 * dW[co][ci][kh][kw] = sum over output pixels (n,ho,wo) of dZ[n,ho,wo,co] * X[n, ho*stride-pad_t+kh, wo*stride-pad_l+kw, ci]
 * (zero outside the image).  dz dense [N*Ho*Wo, Cout], x dense NHWC, dw in torch's OIHW layout. */
int cer_conv2d_wgrad(const float *dz, const float *x, float *dw, int N, int H, int W, int Ho, int Wo, int Cout, int Cin,
                     int KH, int KW, int stride, int pad_t, int pad_l, void *workspace, size_t workspace_bytes, void *stream);
/* The same on the bf16 matrix cores with split hi/lo operands ("bf16x3": three MFMAs per product, <= 2^-15 relative per
 * product, fp32 accumulation), the pixel range split over blocks and reduced in a fixed order: the weight gradient of the
 * reference's released encoder units (base/parameter_control.py:55-103 un-freezes IR-50 stage 4 and half of stage 3; their
 * backward is torch autograd's conv2d weight gradient, models/arcface_model.py:44-60).  Cout and Cin must be multiples of 4
 * (CER_ERR_UNSUPPORTED otherwise: use cer_conv2d_wgrad); the output tile is 128 x 128, so channel counts below 128 waste
 * matrix-core work but stay correct. */
size_t cer_conv2d_wgrad_b3_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW);
int cer_conv2d_wgrad_b3(const float *dz, const float *x, float *dw, int N, int H, int W, int Ho, int Wo, int Cout, int Cin,
                        int KH, int KW, int stride, int pad_t, int pad_l, void *workspace, size_t workspace_bytes, void *stream);
/* The same on operands that are split tensors already (hi / lo bf16 planes, cer_split_bf16): the loader copies instead of
 * converting -- twice as fast when the caller has the split tensors anyway (the released units' convs consume them). */
int cer_conv2d_wgrad_b3s(const uint16_t *dz_hi, const uint16_t *dz_lo, const uint16_t *x_hi, const uint16_t *x_lo, float *dw,
                         int N, int H, int W, int Ho, int Wo, int Cout, int Cin, int KH, int KW, int stride, int pad_t, int pad_l,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Channels-last PReLU with per-channel slopes (arcface_model.py:54).  Backward: dx = x > 0 ? dy : alpha*dy, and
 * dalpha_terms = x > 0 ? 0 : x*dy, whose column sum (cer_col_sum) is the slope gradient. */
int cer_prelu_fwd(const float *x, const float *alpha, float *y, size_t rows, int C, void *stream);
int cer_prelu_bwd(const float *dy, const float *x, const float *alpha, float *dx, float *dalpha_terms, size_t rows, int C,
                  void *stream);

/* out[c] = sum_r a[r][c] * (b ? (b[r][c]-mean[c])*invstd[c] : 1); a == NULL (with b and mean): the centred second moment
 * sum_r ((b[r][c]-mean[c])*invstd[c])^2; deterministic tree.
 * Bias gradients and the BatchNorm / LayerNorm parameter gradients. */
size_t cer_col_sum_workspace_bytes(int R, int C);
/* db = sum dy, dw = sum dy * (x - mean) * invstd over dense rows [R, C] -- the two reductions of the train-mode BatchNorm
 * backward -- in ONE pass over dy and x when C % 4 == 0 and R spans several slabs (two passes otherwise); workspace as cer_col_sum */
int cer_bn_bwd_sums(const float *dy, const float *x, const float *save_mean, const float *save_invstd, float *db, float *dw,
                    int R, int C, void *workspace, size_t workspace_bytes, void *stream);
int cer_col_sum(const float *a, int a_ld, const float *b, int b_ld, const float *mean, const float *invstd,
                float *out, int R, int C, void *workspace, size_t workspace_bytes, void *stream);

/* dz = dy * mask * leaky'(y) for y = mask*leaky(z)  (Dropout + LeakyReLU backward). */
int cer_act_mask_bwd(const float *dy, const float *y, const float *mask, float *dz, size_t n, float slope, void *stream);
/* TemporalBlock tail out = leaky(a2 + res), a2 = mask2*leaky(z2):
 * du = dout*leaky'(out)  (gradient of res and of a2),  dz2 = du*mask2*leaky'(a2). */
int cer_tblock_tail_bwd(const float *dout, const float *out, const float *a2, const float *mask2,
                        float *du, float *dz2, size_t n, float slope, void *stream);

/* BatchNorm1d over rows (reference models/model.py:475,515).  train != 0: batch statistics,
 * running stats updated in place (unbiased variance, `momentum`), save_mean/save_invstd written.
 * train == 0: running statistics. */
size_t cer_bn_rows_fwd_workspace_bytes(int R, int C);   /* 0 for R <= 2048; larger R reduce the statistics in two column sums */
/* y == NULL with train != 0: statistics pass alone (save_mean / save_invstd, running buffers updated), nothing is written --
 * the released encoder units apply the normalisation inside the pass that follows (affine + split, or cer_bn_apply_nhwc
 * with the shortcut add), so the separate apply pass over the activation would be read-once-write-once traffic for nothing. */
int cer_bn_rows_fwd(const float *x, int x_ld, const float *w, const float *b, float *running_mean,
                    float *running_var, float *save_mean, float *save_invstd, float *y, int y_ld,
                    int R, int C, int train, float eps, float momentum, void *workspace, size_t workspace_bytes, void *stream);
int cer_bn_rows_bwd(const float *dy, int dy_ld, const float *x, int x_ld, const float *save_mean,
                    const float *save_invstd, const float *w, float *dx, float *dw, float *db,
                    int R, int C, int train, void *workspace, size_t workspace_bytes, void *stream);

/* LFAN cross-modal attention core (reference models/transformer.py:11-19,133-159): for each
 * (row, head) an M x M softmax over MODALITIES, vals = softmax(q k^T/sqrt(hd)) v + v.
 * qkv[m] [R, H*3*hd] rows laid out [head][q|k|v]; vals [R, H*M*hd]; probs [R,H,M,M] saved. */
int cer_lfan_attn_fwd(const float *const *qkv, float *vals, float *probs, int R, int H, int M, int hd, void *stream);
int cer_lfan_attn_bwd(const float *const *qkv, const float *dvals, const float *probs, float *const *dqkv,
                      int R, int H, int M, int hd, void *stream);

/* y = LayerNorm(x*mask)*gamma+beta (mask = pre-scaled dropout mask or NULL), rows of C.
 * Backward returns dx w.r.t. the un-masked x; scratch holds R*C floats. */
int cer_layernorm_fwd(const float *x, const float *mask, const float *gamma, const float *beta, float *y, int y_ld,
                      float *save_mean, float *save_rstd, int R, int C, float eps, void *stream);
int cer_layernorm_bwd(const float *dy, int dy_ld, const float *x, const float *mask, const float *gamma,
                      const float *save_mean, const float *save_rstd, float *dx, float *dgamma, float *dbeta,
                      float *scratch, int R, int C, void *workspace, size_t workspace_bytes, void *stream);

/* nn.CrossEntropyLoss(reduction='mean') on [R,C] logits with FLOAT labels cast to long
 * (reference experiment.py:133, trainer.py:380-383); dlogits may be NULL.  Label semantics as torch: -100 (ignore_index)
 * rows are skipped (zero gradient, mean over the other rows); any other value outside [0, C), or NaN, is an error:
 * the row is never dereferenced, loss and that row's gradient become NaN and *bad_labels (device int, may be NULL)
 * receives the number of such rows, which the host wrapper turns into the exception torch raises. */
int cer_cross_entropy(const float *logits, const float *labels, float *loss, float *dlogits, int *bad_labels, int R, int C,
                      void *stream);

/* Train-mode BatchNorm2d inside the vision encoder (the reference's model.train() also puts
 * the frozen IR-50's BatchNorms in batch-statistics mode).
 * cer_bn_finalize: reduce `tiles` partial rows [tiles][2][C] (sum, sum of squares over `count`
 * elements per channel) in double precision -> scale = gamma*invstd, shift = beta - mean*scale;
 * updates running_mean/var in place (unbiased variance, `momentum`) when non-NULL.
 * cer_bn_apply_nhwc: out = mask * prelu(y*scale+shift) + (res*res_scale+res_shift), residual
 * sampled with res_stride like the conv epilogue; optionally emits the partial statistics of
 * `out` ([cer_bn_apply_stats_tiles(P)][2][C]) for the next BatchNorm. */
size_t cer_bn_finalize_workspace_bytes(int tiles, int C);
int cer_bn_finalize(const float *partials, int tiles, int C, double count, const float *gamma, const float *beta,
                    float *running_mean, float *running_var, float momentum, float eps,
                    float *scale, float *shift, void *workspace, size_t workspace_bytes, void *stream);
int cer_bn_apply_stats_tiles(int P);
int cer_bn_apply_nhwc(const float *y, const float *scale, const float *shift, const float *alpha,
                      const float *res, const float *res_scale, const float *res_shift, const float *mask,
                      float *out, float *stats, int N, int Ho, int Wo, int C, int res_stride, int Hr, int Wr,
                      void *stream);
/* The same pass for the bf16x3 encoder: the residual may come as a split tensor (res_hi/res_lo instead of res) and
 * the result is stored split (out_hi/out_lo) -- what the next conv reads directly -- and/or as fp32 (`out`); the
 * statistics are taken from the fp32 value before the split. */
int cer_bn_apply_nhwc_b3(const float *y, const float *scale, const float *shift, const float *alpha, const float *res,
                         const uint16_t *res_hi, const uint16_t *res_lo, const float *res_scale, const float *res_shift,
                         const float *mask, float *out, uint16_t *out_hi, uint16_t *out_lo, float *stats, int N, int Ho,
                         int Wo, int C, int res_stride, int Hr, int Wr, void *stream);

/* The same pass on NARROW tensors (cer_storage: one bf16 / half plane each): the conv result comes as fp32 (`y`) or narrow
 * (`y16`, exactly one of the two), the residual as fp32 or narrow, the result goes to `out16` and/or fp32 `out`;
 * arithmetic and statistics are fp32 (taken before the final rounding).  16-byte accesses: C % 8 == 0. */
int cer_bn_apply_nhwc_n16(const float *y, const uint16_t *y16, const float *scale, const float *shift, const float *alpha,
                          const float *res, const uint16_t *res16, const float *res_scale, const float *res_shift,
                          const float *mask, float *out, uint16_t *out16, float *stats, int N, int Ho, int Wo, int C,
                          int res_stride, int Hr, int Wr, int storage, void *stream);

/* Fold the per-input-channel affine (scale, shift) of a BatchNorm in FRONT of a 3x3 / stride 1 / pad 1 conv into the conv
 * (reference arcface_model.py:53-54: BatchNorm2d -> Conv2d; in model.train() the affine only exists after the batch
 * reduction): w_packed [Cout][Kpad] (cer_pack_conv_weight) -> w * scale[c] as fp32 (w_f32), split hi/lo (w_hi, w_lo,
 * storage 0) or one narrow plane (w_hi, storage bf16 / f16), plus bias9 [9][Cout] (see cer_conv_io.bias9).  One launch. */
int cer_fold_bn_3x3(const float *w_packed, const float *scale, const float *shift, int Cout, int Cin, float *w_f32,
                    uint16_t *w_hi, uint16_t *w_lo, int storage, float *bias9, void *stream);

/* Nesterov SGD over flat buffers (reference instantiators.py:74-92: torch.optim.SGD(momentum .9, nesterov, wd 1e-4);
 * trainer.py:385-391): d = grad + wd*p; buf = first_step ? d : mu*buf + (1-damp)*d; d = nesterov ? d + mu*buf : buf;
 * p -= lr*d -- torch's _single_tensor_sgd operation by operation, one launch for the whole model.  n % 4 == 0,
 * 16-byte aligned buffers. */
int cer_sgd_nesterov_flat(float *param, const float *grad, float *momentum_buf, size_t n, float lr, float momentum,
                          float dampening, float weight_decay, int nesterov, int first_step, void *stream);

/* out[i][:] = src[index[i]][:], zeros where index[i] < 0 or >= n_src: the token -> frame spreading of the BERT rows
 * (abaw5_pre_processing/base/speech.py:690-738) and the edge-padded frame indexing of VGGish rows
 * (base/preprocessing.py:992-1018); the index plan is host logic.  cols % 4 == 0. */
int cer_gather_rows(const float *src, const int64_t *index, float *out, int n_out, int cols, int64_t n_src, void *stream);

/* ------------------------------------------------------------------------
 * Audio / text encoder front ends and attention.
 * ---------------------------------------------------------------------- */

/* VGGish log-mel front end (reference abaw5_pre_processing/base/vggish/mel_features.py:75-114,
 * 207-236; vggish_input.py:84-98): pcm [clips][num_samples] int16 at 16 kHz -> /32768 -> `pad_samples`
 * of edge padding -> periodic-Hann 400/160 frames -> |DFT 512| -> mel_matrix [257][64] (float64,
 * host-computed constant) -> log(x + log_offset).  logmel [clips][cer_logmel_num_frames()][64] fp32;
 * the DFT and mel product run in float64 like the reference's numpy. */
int cer_logmel_num_frames(int num_samples, int pad_samples);
int cer_logmel_fwd(const int16_t *pcm, int clips, int num_samples, int pad_samples, const double *mel_matrix,
                   float log_offset, float *logmel, void *stream);
/* examples[c][e][f][:] = logmel[c][starts[e]+f][:] (my_frame, mel_features.py:21-49; the caller
 * computes `starts` with the reference's round-half-to-even rule). */
int cer_frame_examples(const float *logmel, const int *starts, float *examples, int clips, int frames_per_clip,
                       int n_examples, int win, void *stream);

/* BERT embeddings: y[b][s] = LayerNorm(word[ids[b][s]] + pos[s] + type[0]) (transformers
 * BertEmbeddings; reference call site abaw5_pre_processing/base/speech.py:603-606). ids int64. */
int cer_bert_embed_ln(const long long *ids, const float *word, const float *pos, const float *type,
                      const float *gamma, const float *beta, float *y, int B, int S, int Hd, int vocab,
                      int max_pos, float eps, void *stream);

/* out = softmax(q k^T * scale + key mask) v on the fp32 matrix cores (flash style, exact fp32).
 * Element (b, s, h, d) of a tensor sits at ptr + b*strides[0] + s*strides[1] + h*strides[2] + d.
 * key_mask [B][Sk] (1 = attend) or NULL.  D in {32, 64, 128}.  Replaces BertSelfAttention and the
 * nn.MultiheadAttention of the JMT/MT heads (reference models/model.py:716-750, 967-972). */
int cer_attention_fwd(const float *q, const float *k, const float *v, const int *key_mask, float *out, float *lse,
                      int B, int H, int Sq, int Sk, int D, const long long *q_strides, const long long *k_strides,
                      const long long *v_strides, const long long *o_strides, float scale, void *stream);
/* Backward of cer_attention_fwd (JMT/MT heads train through nn.MultiheadAttention).  lse [B,H,Sq] is the
 * forward's optional output; delta [B,H,Sq] is scratch.  Two recompute passes (one owning queries, one
 * owning keys), no atomics: deterministic. */
int cer_attention_bwd(const float *q, const float *k, const float *v, const float *out, const float *dout,
                      const float *lse, const int *key_mask, float *delta, float *dq, float *dk, float *dv,
                      int B, int H, int Sq, int Sk, int D, const long long *q_strides, const long long *k_strides,
                      const long long *v_strides, const long long *o_strides, const long long *do_strides,
                      const long long *dq_strides, const long long *dk_strides, const long long *dv_strides,
                      float scale, void *stream);

/* ------------------------------------------------------------------------
 * Evaluation path on the device (reference trainer.py:436-523, 832-892; metrics.py:43-193).
 * ---------------------------------------------------------------------- */

/* Stitch the outputs of the sliding inference windows of ONE video (trainer.py:832-892): out[f][c] = sum over the windows
 * covering frame f, in window order, of win_out[w][f - starts[w]][c], divided by the number of such windows.
 * win_out [nw][Lw][C], starts [nw] (device int32), out [total][C]. */
int cer_window_stitch(const float *win_out, const int *starts, int nw, int Lw, int C, int total, float *out, void *stream);

/* Accumulate (+=) the confusion counts of a batch of V videos: logits [R][C] (the videos' frames concatenated), labels [R]
 * (float class ids), video_offsets [V+1] (device int32 row offsets).  frame_cm [C][C]: counts[label][argmax]; video_cm
 * [3][C][C]: the same per video for the reference's three frame -> video decisions (majority vote / mean logits / mean
 * softmax probabilities, metrics.py:118-139); ignore_class >= 0 drops the LAST logit column and skips frames / videos
 * labelled ignore_class (C-EXPR-DB's 'Other', metrics.py:62-84); video_pred [V][3] (optional) receives the decisions;
 * *bad counts labels outside [0, C) and videos with mixed labels (the reference asserts on those).  F1 / accuracy / the
 * normalised confusion matrix are functions of these counts (host side: one [4][C][C] copy per evaluation). */
int cer_eval_accumulate(const float *logits, const float *labels, const int *video_offsets, int V, int R, int C,
                        int ignore_class, unsigned long long *frame_cm, unsigned long long *video_cm, int *video_pred, int *bad,
                        void *stream);

/* y += x */
int cer_add_inplace(float *y, const float *x, size_t n, void *stream);

/* Pre-scaled dropout keep-mask, a pure function of (seed, offset + i). */
int cer_dropout_mask(float *mask, size_t n, float p, uint64_t seed, uint64_t offset, void *stream);

/* y = x >= 0 ? x : slope*x  (F.leaky_relu between fc1/bn1 and fc2 of the CAN / JMT heads,
 * reference models/model.py:676,1160; backward = cer_act_mask_bwd). */
int cer_leaky_relu_fwd(const float *x, float *y, size_t n, float slope, void *stream);

/* CAN attention-fusion gate (reference models/model.py:561-566): out = softmax(z) * c over rows of C;
 * prob is saved for the backward, which returns dz and dc. */
int cer_softmax_gate_fwd(const float *z, const float *c, float *out, float *prob, int R, int C, void *stream);
int cer_softmax_gate_bwd(const float *dout, const float *prob, const float *c, float *dz, float *dc, int R, int C,
                         void *stream);

/* y[r, 0:C] (pitch y_ld) = x[r, 0:C] (pitch x_ld). */
int cer_copy_cols(const float *x, int x_ld, float *y, int y_ld, int R, int C, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CER_HIP_H */
