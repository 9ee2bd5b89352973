// unit_passes.hip -- elementwise passes of the RELEASED encoder units (reference models/arcface_model.py:44-60 in train mode, with
// a backward) that hand their result to the matrix-core kernels in the layout those read: PReLU forward / backward and the
// BatchNorm-backward apply pass write Split tensors (hi / lo bf16 planes) directly, and the BatchNorm-1 backward adds the
// shortcut branch's gradient in the same pass.  HBM-bound: 8-16 bytes per element and pass, 4 elements per thread.
#include "conv_b3.h"

namespace cer {

// ---- released encoder units: elementwise passes that hand a SPLIT tensor straight to the matrix-core kernels (round 3).
// The backward of a unit used to write each of these results as fp32 and re-read it in a separate split pass (5 split passes
// per unit: 7.5 % of the whole-encoder training step at 1024 frames of 224x224); the arithmetic is unchanged.
// t = prelu(x) -> split (the conv input rebuilt from the raw conv result z1)
__global__ void prelu_split_kernel(const float4 *__restrict__ x, const float *__restrict__ alpha, ushort4 *__restrict__ hi,
                                   ushort4 *__restrict__ lo, size_t n4, int C4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = x[i], a = *reinterpret_cast<const float4 *>(alpha + (i % C4) * 4);
    const float t[4] = {v.x > 0.f ? v.x : a.x * v.x, v.y > 0.f ? v.y : a.y * v.y, v.z > 0.f ? v.z : a.z * v.z, v.w > 0.f ? v.w : a.w * v.w};
    ushort4 h, l;
    split_bf16(t[0], h.x, l.x); split_bf16(t[1], h.y, l.y); split_bf16(t[2], h.z, l.z); split_bf16(t[3], h.w, l.w);
    hi[i] = h;
    lo[i] = l;
}

// torch's prelu backward with dx as a split tensor (or fp32 when dx is given); t = the slope-gradient terms (column-summed by the caller)
__global__ void prelu_bwd_split_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ alpha,
                                       float4 *__restrict__ dx, ushort4 *__restrict__ hi, ushort4 *__restrict__ lo,
                                       float4 *__restrict__ t, size_t n4, int C4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 g = dy[i], v = x[i], a = *reinterpret_cast<const float4 *>(alpha + (i % C4) * 4);
    const float d[4] = {v.x > 0.f ? g.x : a.x * g.x, v.y > 0.f ? g.y : a.y * g.y, v.z > 0.f ? g.z : a.z * g.z, v.w > 0.f ? g.w : a.w * g.w};
    if (dx) dx[i] = make_float4(d[0], d[1], d[2], d[3]);
    if (hi) {
        ushort4 h, l;
        split_bf16(d[0], h.x, l.x); split_bf16(d[1], h.y, l.y); split_bf16(d[2], h.z, l.z); split_bf16(d[3], h.w, l.w);
        hi[i] = h;
        lo[i] = l;
    }
    t[i] = make_float4(v.x > 0.f ? 0.f : v.x * g.x, v.y > 0.f ? 0.f : v.y * g.y, v.z > 0.f ? 0.f : v.z * g.z, v.w > 0.f ? 0.f : v.w * g.w);
}

// train-mode BatchNorm backward, apply pass (the two column sums come from cer_col_sum): dx = w * invstd * (dy - (sum_dy + x_hat *
// sum_dy_xhat) / R) as a split tensor, 4 channels per thread
__global__ void bn_rows_bwd_split_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ mean,
                                         const float *__restrict__ invstd, const float *__restrict__ w, const float *__restrict__ sum_dy,
                                         const float *__restrict__ sum_dy_xhat, ushort4 *__restrict__ hi, ushort4 *__restrict__ lo,
                                         size_t n4, int C4, float inv_r) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    const float4 g = dy[i], v = x[i];
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
    const float4 ww = *reinterpret_cast<const float4 *>(w + c), s1 = *reinterpret_cast<const float4 *>(sum_dy + c);
    const float4 s2 = *reinterpret_cast<const float4 *>(sum_dy_xhat + c);
    const float d[4] = {ww.x * is.x * (g.x - (s1.x + (v.x - mu.x) * is.x * s2.x) * inv_r),
                        ww.y * is.y * (g.y - (s1.y + (v.y - mu.y) * is.y * s2.y) * inv_r),
                        ww.z * is.z * (g.z - (s1.z + (v.z - mu.z) * is.z * s2.z) * inv_r),
                        ww.w * is.w * (g.w - (s1.w + (v.w - mu.w) * is.w * s2.w) * inv_r)};
    ushort4 h, l;
    split_bf16(d[0], h.x, l.x); split_bf16(d[1], h.y, l.y); split_bf16(d[2], h.z, l.z); split_bf16(d[3], h.w, l.w);
    hi[i] = h;
    lo[i] = l;
}

// the same apply pass with an fp32 result and an optional addend: dx = BatchNorm-backward(dy) + add -- the unit input's gradient
// is the sum of the residual branch (through BatchNorm 1) and the shortcut branch (dout itself, or the projection's data
// gradient), which used to be a separate read-modify-write pass over dx
__global__ void bn_rows_bwd_add_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ mean,
                                       const float *__restrict__ invstd, const float *__restrict__ w, const float *__restrict__ sum_dy,
                                       const float *__restrict__ sum_dy_xhat, const float4 *__restrict__ add, float4 *__restrict__ dx,
                                       size_t n4, int C4, float inv_r) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    const float4 g = dy[i], v = x[i];
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
    const float4 ww = *reinterpret_cast<const float4 *>(w + c), s1 = *reinterpret_cast<const float4 *>(sum_dy + c);
    const float4 s2 = *reinterpret_cast<const float4 *>(sum_dy_xhat + c);
    float4 d = make_float4(ww.x * is.x * (g.x - (s1.x + (v.x - mu.x) * is.x * s2.x) * inv_r),
                           ww.y * is.y * (g.y - (s1.y + (v.y - mu.y) * is.y * s2.y) * inv_r),
                           ww.z * is.z * (g.z - (s1.z + (v.z - mu.z) * is.z * s2.z) * inv_r),
                           ww.w * is.w * (g.w - (s1.w + (v.w - mu.w) * is.w * s2.w) * inv_r));
    if (add) {
        const float4 a = add[i];
        d = make_float4(d.x + a.x, d.y + a.y, d.z + a.z, d.w + a.w);
    }
    dx[i] = d;
}

}  // namespace cer

using namespace cer;


extern "C" int cer_prelu_split(const float *x, const float *alpha, uint16_t *hi, uint16_t *lo, size_t rows, int C, void *stream) {
    if (!x || !alpha || !hi || !lo || rows == 0 || C <= 0 || (C & 3)) return cer_set_error(CER_ERR_INVALID_ARG, "prelu_split: C % 4 == 0");
    const size_t n4 = rows * (C / 4);
    CER_LAUNCH(prelu_split_kernel, dim3(cer_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, alpha, (ushort4 *)hi,
               (ushort4 *)lo, n4, C / 4);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_prelu_bwd_split(const float *dy, const float *x, const float *alpha, float *dx, uint16_t *dx_hi, uint16_t *dx_lo,
                                   float *dalpha_terms, size_t rows, int C, void *stream) {
    if (!dy || !x || !alpha || (!dx && !dx_hi) || ((dx_hi == nullptr) != (dx_lo == nullptr)) || !dalpha_terms || rows == 0 || C <= 0 ||
        (C & 3))
        return cer_set_error(CER_ERR_INVALID_ARG, "prelu_bwd_split: needs dx and / or both split planes, C % 4 == 0");
    const size_t n4 = rows * (C / 4);
    CER_LAUNCH(prelu_bwd_split_kernel, dim3(cer_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)dy,
               (const float4 *)x, alpha, (float4 *)dx, (ushort4 *)dx_hi, (ushort4 *)dx_lo, (float4 *)dalpha_terms, n4, C / 4);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_bn_rows_bwd_split(const float *dy, const float *x, const float *save_mean, const float *save_invstd,
                                     const float *w, uint16_t *dx_hi, uint16_t *dx_lo, float *dw, float *db, int R, int C,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    if (!dy || !x || !save_mean || !save_invstd || !w || !dx_hi || !dx_lo || !dw || !db || R <= 0 || C <= 0 || (C & 3))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_rows_bwd_split: bad argument (dense rows, C % 4 == 0)");
    const int rc = cer_bn_bwd_sums(dy, x, save_mean, save_invstd, db, dw, R, C, workspace, workspace_bytes, stream);
    if (rc) return rc;
    const size_t n4 = (size_t)R * (C / 4);
    CER_LAUNCH(bn_rows_bwd_split_kernel, dim3(cer_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)dy,
               (const float4 *)x, save_mean, save_invstd, w, (const float *)db, (const float *)dw, (ushort4 *)dx_hi, (ushort4 *)dx_lo, n4,
               C / 4, 1.0f / (float)R);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_bn_rows_bwd_add(const float *dy, const float *x, const float *save_mean, const float *save_invstd, const float *w,
                                   const float *add, float *dx, float *dw, float *db, int R, int C, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    if (!dy || !x || !save_mean || !save_invstd || !w || !dx || !dw || !db || R <= 0 || C <= 0 || (C & 3))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_rows_bwd_add: bad argument (dense rows, C % 4 == 0)");
    const int rc = cer_bn_bwd_sums(dy, x, save_mean, save_invstd, db, dw, R, C, workspace, workspace_bytes, stream);
    if (rc) return rc;
    const size_t n4 = (size_t)R * (C / 4);
    CER_LAUNCH(bn_rows_bwd_add_kernel, dim3(cer_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)dy,
               (const float4 *)x, save_mean, save_invstd, w, (const float *)db, (const float *)dw, (const float4 *)add, (float4 *)dx, n4,
               C / 4, 1.0f / (float)R);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
