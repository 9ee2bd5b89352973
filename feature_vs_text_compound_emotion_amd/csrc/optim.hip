// optim.hip -- the optimiser update and the feature-to-frame row gather of the host loop.
//
// sgd_nesterov_flat: the reference trains with torch.optim.SGD(momentum=0.9, nesterov=True, weight_decay=1e-4)
// (/root/reference/instantiators.py:74-92, trainer.py:385-391).  All trainable parameters, their gradients and the
// momentum buffers live in three flat fp32 buffers (data_parallel.py), so the whole update is ONE bandwidth-bound
// launch (16 B read + 8 B written per parameter) instead of ~100 multi-tensor launches.  Arithmetic follows torch's
// _single_tensor_sgd operation by operation (grad + wd*p; buf = mu*buf + (1-damp)*d, or d on the first step;
// d + mu*buf; p - lr*d) so the result is bit-identical to it.
//
// gather_rows: out[i] = src[index[i]] (zeros for index < 0): token -> frame spreading of the BERT features
// (abaw5_pre_processing/base/speech.py:690-738) and the edge-padded frame indexing of VGGish rows
// (base/preprocessing.py:992-1018); the index plan is host logic (feature_extractor.py).
#include <stdint.h>

#include "cer_internal.h"

namespace cer {

__global__ void sgd_nesterov_flat_kernel(float4 *__restrict__ p, const float4 *__restrict__ g, float4 *__restrict__ buf,
                                         size_t n4, float lr, float mu, float damp, float wd, int nesterov, int first) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 pv = p[i];
    const float4 gv = g[i];
    float4 bv = first ? make_float4(0, 0, 0, 0) : buf[i];
    float *pp = reinterpret_cast<float *>(&pv), *bb = reinterpret_cast<float *>(&bv);
    const float *gg = reinterpret_cast<const float *>(&gv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float d = wd != 0.f ? __fmaf_rn(wd, pp[e], gg[e]) : gg[e];          // grad.add(param, alpha=wd)
        if (mu != 0.f) {
            bb[e] = first ? d : __fmaf_rn(1.f - damp, d, __fmul_rn(mu, bb[e]));  // buf.mul_(mu).add_(d, alpha=1-damp)
            d = nesterov ? __fmaf_rn(mu, bb[e], d) : bb[e];                  // d.add(buf, alpha=mu)
        }
        pp[e] = __fmaf_rn(-lr, d, pp[e]);                                    // param.add_(d, alpha=-lr)
    }
    p[i] = pv;
    if (mu != 0.f) buf[i] = bv;
}

__global__ void gather_rows_kernel(const float4 *__restrict__ src, const int64_t *__restrict__ index, float4 *__restrict__ out,
                                   int n_out, int cols4, int64_t n_src) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_out * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (size_t)r * cols4);
    const int64_t s = index[r];
    out[i] = (s >= 0 && s < n_src) ? src[(size_t)s * cols4 + c] : make_float4(0, 0, 0, 0);
}

// channels-last PReLU with per-channel slopes (arcface_model.py:54): y = x > 0 ? x : alpha[c] * x
__global__ void prelu_fwd_kernel(const float4 *__restrict__ x, const float *__restrict__ alpha, float4 *__restrict__ y, size_t n4,
                                 int C4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = x[i], a = *reinterpret_cast<const float4 *>(alpha + (i % C4) * 4);
    y[i] = make_float4(v.x > 0.f ? v.x : a.x * v.x, v.y > 0.f ? v.y : a.y * v.y, v.z > 0.f ? v.z : a.z * v.z,
                       v.w > 0.f ? v.w : a.w * v.w);
}
// torch's prelu backward: dx = x > 0 ? dy : alpha[c]*dy; the slope gradient is the column sum of t = x > 0 ? 0 : x*dy
__global__ void prelu_bwd_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ alpha,
                                 float4 *__restrict__ dx, float4 *__restrict__ t, size_t n4, int C4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 g = dy[i], v = x[i], a = *reinterpret_cast<const float4 *>(alpha + (i % C4) * 4);
    dx[i] = make_float4(v.x > 0.f ? g.x : a.x * g.x, v.y > 0.f ? g.y : a.y * g.y, v.z > 0.f ? g.z : a.z * g.z,
                        v.w > 0.f ? g.w : a.w * g.w);
    t[i] = make_float4(v.x > 0.f ? 0.f : v.x * g.x, v.y > 0.f ? 0.f : v.y * g.y, v.z > 0.f ? 0.f : v.z * g.z,
                       v.w > 0.f ? 0.f : v.w * g.w);
}

// y = x / ||x||  ->  dx = (dy - y * (y . dy)) / ||x||   (one wave per row; reference models/arcface_model.py:17-20)
__global__ void l2norm_rows_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ dx,
                                       int rows, int cols) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * cols, *gr = dy + (size_t)row * cols;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < cols; c += 64) {
        ss += xr[c] * xr[c];
        dot += xr[c] * gr[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o);
        dot += __shfl_xor(dot, o);
    }
    const float inv = 1.f / sqrtf(ss);
    const float k = dot * inv * inv * inv;  // (y . dy) / ||x|| * (1/||x||) with y = x/||x||
    for (int c = lane; c < cols; c += 64) dx[(size_t)row * cols + c] = gr[c] * inv - xr[c] * k;
}

}  // namespace cer

using namespace cer;

extern "C" int cer_l2norm_rows_bwd(const float *dy, const float *x, float *dx, int rows, int cols, void *stream) {
    if (!dy || !x || !dx || rows <= 0 || cols <= 0) return cer_set_error(CER_ERR_INVALID_ARG, "l2norm_rows_bwd: bad argument");
    CER_LAUNCH(l2norm_rows_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, x, dx, rows, cols);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_sgd_nesterov_flat(float *param, const float *grad, float *momentum_buf, size_t n, float lr, float momentum,
                                     float dampening, float weight_decay, int nesterov, int first_step, void *stream) {
    if (!param || !grad || n == 0 || (n & 3) || (momentum != 0.f && !momentum_buf))
        return cer_set_error(CER_ERR_INVALID_ARG, "sgd_nesterov_flat: needs param, grad, (momentum_buf), n a positive multiple of 4");
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15)
        return cer_set_error(CER_ERR_INVALID_ARG, "sgd_nesterov_flat: buffers must be 16-byte aligned");
    if (nesterov && (momentum <= 0.f || dampening != 0.f))
        return cer_set_error(CER_ERR_INVALID_ARG, "sgd_nesterov_flat: Nesterov momentum requires a momentum and zero dampening");
    CER_LAUNCH(sgd_nesterov_flat_kernel, dim3(cer_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (float4 *)param,
               (const float4 *)grad, (float4 *)momentum_buf, n / 4, lr, momentum, dampening, weight_decay, nesterov, first_step);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_gather_rows(const float *src, const int64_t *index, float *out, int n_out, int cols, int64_t n_src,
                               void *stream) {
    if (!src || !index || !out || n_out <= 0 || cols <= 0 || (cols & 3) || n_src <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "gather_rows: needs src, index, out and cols % 4 == 0");
    CER_LAUNCH(gather_rows_kernel, dim3(cer_blocks((size_t)n_out * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream,
               (const float4 *)src, index, (float4 *)out, n_out, cols / 4, n_src);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_prelu_fwd(const float *x, const float *alpha, float *y, size_t rows, int C, void *stream) {
    if (!x || !alpha || !y || rows == 0 || C <= 0 || (C & 3)) return cer_set_error(CER_ERR_INVALID_ARG, "prelu_fwd: C % 4 == 0");
    const size_t n4 = rows * (C / 4);
    CER_LAUNCH(prelu_fwd_kernel, dim3(cer_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, alpha, (float4 *)y,
               n4, C / 4);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_prelu_bwd(const float *dy, const float *x, const float *alpha, float *dx, float *dalpha_terms, size_t rows,
                             int C, void *stream) {
    if (!dy || !x || !alpha || !dx || !dalpha_terms || rows == 0 || C <= 0 || (C & 3))
        return cer_set_error(CER_ERR_INVALID_ARG, "prelu_bwd: C % 4 == 0");
    const size_t n4 = rows * (C / 4);
    CER_LAUNCH(prelu_bwd_kernel, dim3(cer_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)dy, (const float4 *)x,
               alpha, (float4 *)dx, (float4 *)dalpha_terms, n4, C / 4);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
