// cer_common.hip -- error state, weight packing and small bandwidth-bound kernels.
#include <stdarg.h>
#include <stdio.h>

#include "cer_internal.h"

static thread_local char g_err[512] = "";

int cer_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *cer_last_error(void) { return g_err; }
extern "C" int cer_version(void) { return 100; }

namespace cer {

// OIHW -> [rows][Kpad].  Forward layout: rows = Cout, k = (kh*KW + kw)*Cin + c, optional
// per-cout scale (BatchNorm fold).  transpose != 0 builds the DATA-GRADIENT filter of a
// stride-1 conv instead: rows = Cin, k = (kh*KW + kw)*Cout + o, taps flipped when flip != 0.
__global__ void pack_conv_weight_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                        float *__restrict__ out, int Cout, int Cin, int KH, int KW,
                                        int Kpad, int flip, int transpose) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = transpose ? Cin : Cout, inner = transpose ? Cout : Cin;
    if (idx >= (size_t)rows * Kpad) return;
    int row = (int)(idx / Kpad), k = (int)(idx - (size_t)row * Kpad);
    float v = 0.f;
    if (k < KH * KW * inner) {
        int tap = k / inner, c = k - tap * inner;
        int kh = tap / KW, kw = tap - kh * KW;
        if (flip) { kh = KH - 1 - kh; kw = KW - 1 - kw; }
        const int o = transpose ? c : row, i = transpose ? row : c;
        v = w[(((size_t)o * Cin + i) * KH + kh) * KW + kw];
        if (scale) v *= scale[o];
    }
    out[idx] = v;
}

// One wave per row: y = x / ||x||_2 (no epsilon, like the reference).
__global__ void l2norm_rows_kernel(const float *__restrict__ x, float *__restrict__ y, int rows, int cols) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { float v = xr[c]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.f / sqrtf(s);
    float *yr = y + (size_t)row * cols;
    for (int c = lane; c < cols; c += 64) yr[c] = xr[c] * inv;
}

__global__ void maxpool2x2_nhwc_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, int N, int H,
                                       int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)N * Ho * Wo * C4;
    if (idx >= total) return;
    int c = (int)(idx % C4);
    size_t t = idx / C4;
    int wo = (int)(t % Wo); t /= Wo;
    int ho = (int)(t % Ho);
    int n = (int)(t / Ho);
    const float4 *p = x + (((size_t)n * H + 2 * ho) * W + 2 * wo) * C4 + c;
    float4 a = p[0], b = p[C4], d = p[(size_t)W * C4], e = p[(size_t)W * C4 + C4];
    float4 r;
    r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x));
    r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
    r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z));
    r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
    y[idx] = r;
}

}  // namespace cer

using namespace cer;

extern "C" int cer_pack_conv_weight(const float *w_oihw, const float *out_scale, float *w_packed, int Cout,
                                    int Cin, int KH, int KW, int flip, int transpose, void *stream) {
    if (!w_oihw || !w_packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "pack_conv_weight: bad argument");
    const int Kpad = cer_conv_kpad(KH, KW, transpose ? Cout : Cin);
    size_t n = (size_t)(transpose ? Cin : Cout) * Kpad;
    CER_LAUNCH(pack_conv_weight_kernel, dim3(cer_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, out_scale, w_packed, Cout, Cin, KH, KW, Kpad, flip, transpose);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_l2norm_rows(const float *x, float *y, int rows, int cols, void *stream) {
    if (!x || !y || rows <= 0 || cols <= 0) return cer_set_error(CER_ERR_INVALID_ARG, "l2norm_rows: bad argument");
    CER_LAUNCH(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_maxpool2x2_nhwc(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    if (!x || !y || N <= 0 || H < 2 || W < 2 || C <= 0 || (C & 3) || (H & 1) || (W & 1))
        return cer_set_error(CER_ERR_INVALID_ARG, "maxpool2x2_nhwc: need even H, W and C % 4 == 0");
    size_t n = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    CER_LAUNCH(maxpool2x2_nhwc_kernel, dim3(cer_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)x, (float4 *)y, N, H, W, C / 4);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
