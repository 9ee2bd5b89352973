// encoder_bn.hip -- batch-statistics BatchNorm2d for the vision encoder in train mode.
//
// The reference calls model.train() on the whole LFAN (trainer.py:318), which also flips the
// FROZEN IR-50's 53 BatchNorm2d layers to batch statistics and keeps updating their running
// buffers.  To stay faithful without extra read passes, the producing kernels
// emit deterministic per-tile partial sums (sum, sum of squares per channel): the conv epilogue
// for raw conv results, bn_apply_nhwc for unit outputs.  bn_finalize reduces the partials in
// double precision into a per-channel scale/shift and performs torch's running-stat update;
// bn_apply_nhwc is the one bandwidth-bound pass per unit that normalises the conv result, adds
// the (optionally normalised) shortcut and gathers the statistics of the sum for the next unit.
#include "conv_common.h"

namespace cer {

// Stage A: G blocks each fold a contiguous run of partial rows into one double-precision row
// (threads run across channels, so every load is coalesced).  Stage B: one thread per channel
// folds the G rows and produces scale/shift + the running-stat update.  Fixed order -> deterministic.
constexpr int FIN_MAX_GROUPS = 512;

__global__ void bn_partial_reduce_kernel(const float *__restrict__ partials, int tiles, int C, int rows_per_group,
                                         double *__restrict__ out) {
    const int g = blockIdx.x;
    const int t0 = g * rows_per_group, t1 = min(tiles, t0 + rows_per_group);
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {  // 2*C contiguous floats per partial row
        double s = 0.0;
        for (int t = t0; t < t1; ++t) s += (double)partials[(size_t)t * 2 * C + c];
        out[(size_t)g * 2 * C + c] = s;
    }
}

// One wave per channel: lanes fold the group rows, then a double-precision wave reduction.
__global__ void bn_finalize_kernel(const double *__restrict__ groups, int ngroups, int C, double count,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float *__restrict__ running_mean, float *__restrict__ running_var,
                                   float momentum, float eps, float *__restrict__ scale, float *__restrict__ shift) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < ngroups; t += 64) {
        s1 += groups[((size_t)t * 2 + 0) * C + c];
        s2 += groups[((size_t)t * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane != 0) return;
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

static int apply_rows_per_block(int P) {  // ~2048 blocks, 128..2048 rows each (multiple of 128)
    int r = (P / 2048 + 127) / 128 * 128;
    return r < 128 ? 128 : (r > 2048 ? 2048 : r);
}

struct ApplyArgs {
    const float *y, *scale, *shift, *alpha, *res, *res_scale, *res_shift, *mask;
    float *out, *stats;
    int P, Ho, Wo, C, res_stride, Hr, Wr, rows_per_block;
    const uint16_t *res_hi, *res_lo;  // residual as a split tensor (alternative to res)
    uint16_t *out_hi, *out_lo;        // result stored split (besides / instead of out)
};

// Threads are laid out [rows_per_pass][C/4]; each thread keeps its 4 channels for the whole block
// so the channel statistics accumulate in registers and meet in LDS once at the end.
__global__ __launch_bounds__(256) void bn_apply_nhwc_kernel(ApplyArgs p) {
    __shared__ float red[2][256][4];
    const int c4n = p.C >> 2;                    // float4 per row (16..128)
    const int rpp = 256 / c4n;                   // rows per pass
    const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
    const int c = tc * 4;
    const bool active = tr < rpp;
    const float4 sc = *reinterpret_cast<const float4 *>(p.scale + c);
    const float4 sh = *reinterpret_cast<const float4 *>(p.shift + c);
    float4 al = make_float4(1, 1, 1, 1), rs = make_float4(1, 1, 1, 1), rt = make_float4(0, 0, 0, 0);
    if (p.alpha) al = *reinterpret_cast<const float4 *>(p.alpha + c);
    if (p.res_scale) {
        rs = *reinterpret_cast<const float4 *>(p.res_scale + c);
        rt = *reinterpret_cast<const float4 *>(p.res_shift + c);
    }
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const int row0 = blockIdx.x * p.rows_per_block;
    if (active) {
        for (int r = row0 + tr; r < min(p.P, row0 + p.rows_per_block); r += rpp) {
            const size_t off = (size_t)r * p.C + c;
            float4 v = *reinterpret_cast<const float4 *>(p.y + off);
            float o[4] = {v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w};
            if (p.alpha) {
                o[0] = o[0] >= 0.f ? o[0] : o[0] * al.x; o[1] = o[1] >= 0.f ? o[1] : o[1] * al.y;
                o[2] = o[2] >= 0.f ? o[2] : o[2] * al.z; o[3] = o[3] >= 0.f ? o[3] : o[3] * al.w;
            }
            if (p.mask) {
                const float4 m = *reinterpret_cast<const float4 *>(p.mask + off);
                o[0] *= m.x; o[1] *= m.y; o[2] *= m.z; o[3] *= m.w;
            }
            if (p.res || p.res_hi) {
                size_t roff;
                if (p.res_stride == 1 && p.Hr == p.Ho && p.Wr == p.Wo) {
                    roff = off;
                } else {
                    const int hw = p.Ho * p.Wo;
                    const int n = r / hw, q = r - n * hw;
                    const int ho = q / p.Wo, wo = q - ho * p.Wo;
                    roff = ((size_t)(n * p.Hr + ho * p.res_stride) * p.Wr + wo * p.res_stride) * p.C + c;
                }
                float4 x;
                if (p.res) {
                    x = *reinterpret_cast<const float4 *>(p.res + roff);
                } else {
                    const ushort4 h = *reinterpret_cast<const ushort4 *>(p.res_hi + roff);
                    const ushort4 l = *reinterpret_cast<const ushort4 *>(p.res_lo + roff);
                    x = make_float4(bf16_to_f32(h.x) + bf16_to_f32(l.x), bf16_to_f32(h.y) + bf16_to_f32(l.y),
                                    bf16_to_f32(h.z) + bf16_to_f32(l.z), bf16_to_f32(h.w) + bf16_to_f32(l.w));
                }
                o[0] += x.x * rs.x + rt.x; o[1] += x.y * rs.y + rt.y;
                o[2] += x.z * rs.z + rt.z; o[3] += x.w * rs.w + rt.w;
            }
            if (p.out) *reinterpret_cast<float4 *>(p.out + off) = make_float4(o[0], o[1], o[2], o[3]);
            if (p.out_hi) store_split4(p.out_hi + off, p.out_lo + off, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += o[e]; s2[e] += o[e] * o[e]; }
        }
    }
    if (!p.stats) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < c4n) {
        float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
        for (int t = 0; t < rpp; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a1[e] += red[0][t * c4n + threadIdx.x][e];
                a2[e] += red[1][t * c4n + threadIdx.x][e];
            }
        }
        float *dst = p.stats + (size_t)blockIdx.x * 2 * p.C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[threadIdx.x * 4 + e] = a1[e];
            dst[p.C + threadIdx.x * 4 + e] = a2[e];
        }
    }
}

// ---- narrow (single 16-bit plane) variant: 8 channels per thread = 16-byte loads / stores of the narrow tensors ----
struct ApplyArgsN16 {
    const float *y;          // conv result as fp32 ...
    const uint16_t *y16;     // ... or as a narrow plane (exactly one of the two)
    const float *scale, *shift, *alpha, *res, *res_scale, *res_shift, *mask;
    const uint16_t *res16;   // residual as a narrow plane (alternative to res)
    float *out, *stats;
    uint16_t *out16;
    int P, Ho, Wo, C, res_stride, Hr, Wr, rows_per_block, narrow;
};

__device__ __forceinline__ void load8(const float *p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8_n16(const uint16_t *p, float v[8], int narrow) {
    const uint4 q = *reinterpret_cast<const uint4 *>(p);
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = n16_to_f32((uint16_t)(w[i] & 0xffffu), narrow);
        v[2 * i + 1] = n16_to_f32((uint16_t)(w[i] >> 16), narrow);
    }
}
__device__ __forceinline__ void store8_n16(uint16_t *p, const float v[8], int narrow) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_n16(v[2 * i], narrow) | ((unsigned)f32_to_n16(v[2 * i + 1], narrow) << 16);
    *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void bn_apply_n16_kernel(ApplyArgsN16 p) {
    __shared__ float red[2][256][8];
    const int c8n = p.C >> 3;                    // 8-channel groups per row (8..128)
    const int rpp = 256 / c8n;                   // rows per pass
    const int tc = threadIdx.x % c8n, tr = threadIdx.x / c8n;
    const int c = tc * 8;
    float sc[8], sh[8], al[8], rs[8], rt[8];
    load8(p.scale + c, sc);
    load8(p.shift + c, sh);
    if (p.alpha) load8(p.alpha + c, al);
    if (p.res_scale) {
        load8(p.res_scale + c, rs);
        load8(p.res_shift + c, rt);
    }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    const int row0 = blockIdx.x * p.rows_per_block;
    for (int r = row0 + tr; r < min(p.P, row0 + p.rows_per_block); r += rpp) {
        const size_t off = (size_t)r * p.C + c;
        float o[8];
        if (p.y16) load8_n16(p.y16 + off, o, p.narrow);
        else load8(p.y + off, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * sc[e] + sh[e];
        if (p.alpha) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] >= 0.f ? o[e] : o[e] * al[e];
        }
        if (p.mask) {
            float m[8];
            load8(p.mask + off, m);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= m[e];
        }
        if (p.res || p.res16) {
            size_t roff;
            if (p.res_stride == 1 && p.Hr == p.Ho && p.Wr == p.Wo) {
                roff = off;
            } else {
                const int hw = p.Ho * p.Wo;
                const int n = r / hw, q = r - n * hw;
                const int ho = q / p.Wo, wo = q - ho * p.Wo;
                roff = ((size_t)(n * p.Hr + ho * p.res_stride) * p.Wr + wo * p.res_stride) * p.C + c;
            }
            float x[8];
            if (p.res) load8(p.res + roff, x);
            else load8_n16(p.res16 + roff, x, p.narrow);
            if (p.res_scale) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += x[e] * rs[e] + rt[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += x[e];
            }
        }
        if (p.out) {
            *reinterpret_cast<float4 *>(p.out + off) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4 *>(p.out + off + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (p.out16) store8_n16(p.out16 + off, o, p.narrow);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += o[e]; s2[e] += o[e] * o[e]; }
    }
    if (!p.stats) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < c8n) {
        float a1[8], a2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a1[e] = a2[e] = 0.f;
        for (int t = 0; t < rpp; ++t) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a1[e] += red[0][t * c8n + threadIdx.x][e];
                a2[e] += red[1][t * c8n + threadIdx.x][e];
            }
        }
        float *dst = p.stats + (size_t)blockIdx.x * 2 * p.C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dst[threadIdx.x * 8 + e] = a1[e];
            dst[p.C + threadIdx.x * 8 + e] = a2[e];
        }
    }
}

// ---- fold a pre-conv BatchNorm (per-input-channel scale s, shift t) into a packed 3x3 weight, one launch per unit ----
// conv3x3(pad0(s*x + t)) = conv3x3'(pad0(x)) + sum over the taps INSIDE the image of W_tap . t:  w'[o][tap][c] = w * s[c]
// (written as fp32, as a split hi/lo pair or as one narrow plane) and bias9[3*ry + rx][o] = sum of B[o][tap] over the taps
// that exist for an output pixel on the first / an inner / the last row (ry) and column (rx), B[o][tap] = sum_c w * t[c].
// One block per output channel; replaces an einsum + six sums / stacks + pack + split chain of torch launches per unit
// and training step (ops.fold_input_bn_3x3).
__global__ __launch_bounds__(256) void fold_bn_3x3_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int Cout, int Cin, int Kpad,
                                                          float *__restrict__ w_f32, uint16_t *__restrict__ w_hi,
                                                          uint16_t *__restrict__ w_lo, int storage, float *__restrict__ bias9) {
    __shared__ float red[9][256];
    const int o = blockIdx.x, tid = threadIdx.x;
    const float *wr = w + (size_t)o * Kpad;
    float bt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) bt[t] = 0.f;
    for (int c = tid; c < Cin; c += 256) {
        const float s = scale[c], sh = shift[c];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const size_t k = (size_t)t * Cin + c;
            const float v = wr[k];
            bt[t] += v * sh;
            const float ws = v * s;
            const size_t idx = (size_t)o * Kpad + k;
            if (w_f32) w_f32[idx] = ws;
            if (w_hi) {
                if (storage) w_hi[idx] = f32_to_n16(ws, storage);
                else split_bf16(ws, w_hi[idx], w_lo[idx]);
            }
        }
    }
    for (int k = 9 * Cin + tid; k < Kpad; k += 256) {  // K padding stays zero
        const size_t idx = (size_t)o * Kpad + k;
        if (w_f32) w_f32[idx] = 0.f;
        if (w_hi) { w_hi[idx] = 0; if (!storage) w_lo[idx] = 0; }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[t][tid] = bt[t];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int t = 0; t < 9; ++t) red[t][tid] += red[t][tid + s];
        }
        __syncthreads();
    }
    if (tid < 9) {  // border case 3*ry + rx: rows kh in [ry == 0 ? 1 : 0, ry == 2 ? 1 : 2], same for columns
        const int ry = tid / 3, rx = tid % 3;
        float b = 0.f;
        for (int kh = (ry == 0 ? 1 : 0); kh <= (ry == 2 ? 1 : 2); ++kh)
            for (int kw = (rx == 0 ? 1 : 0); kw <= (rx == 2 ? 1 : 2); ++kw) b += red[kh * 3 + kw][0];
        bias9[(size_t)tid * Cout + o] = b;
    }
}

}  // namespace cer

using namespace cer;

extern "C" size_t cer_bn_finalize_workspace_bytes(int tiles, int C) {
    if (tiles <= 0 || C <= 0) return 0;
    const int groups = tiles < FIN_MAX_GROUPS ? tiles : FIN_MAX_GROUPS;
    return (size_t)groups * 2 * C * sizeof(double);
}

extern "C" int cer_bn_finalize(const float *partials, int tiles, int C, double count, const float *gamma,
                               const float *beta, float *running_mean, float *running_var, float momentum, float eps,
                               float *scale, float *shift, void *workspace, size_t workspace_bytes, void *stream) {
    if (!partials || tiles <= 0 || C <= 0 || !(count > 0) || !gamma || !beta || !scale || !shift ||
        ((running_mean == nullptr) != (running_var == nullptr)))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_finalize: bad argument");
    if (!workspace || workspace_bytes < cer_bn_finalize_workspace_bytes(tiles, C) || ((uintptr_t)workspace & 7))
        return cer_set_error(CER_ERR_WORKSPACE, "bn_finalize: workspace too small or not 8-byte aligned");
    const int groups = tiles < FIN_MAX_GROUPS ? tiles : FIN_MAX_GROUPS;
    const int rows_per_group = (tiles + groups - 1) / groups;
    const int used = (tiles + rows_per_group - 1) / rows_per_group;
    CER_LAUNCH(bn_partial_reduce_kernel, dim3(used), dim3(256), 0, (hipStream_t)stream, partials, tiles, C,
               rows_per_group, (double *)workspace);
    CER_LAUNCH(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const double *)workspace,
               used, C, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_bn_apply_stats_tiles(int P) {
    if (P <= 0) return 0;
    const int r = apply_rows_per_block(P);
    return (P + r - 1) / r;
}

static int bn_apply_launch(const float *y, const float *scale, const float *shift, const float *alpha, const float *res,
                           const uint16_t *res_hi, const uint16_t *res_lo, const float *res_scale, const float *res_shift,
                           const float *mask, float *out, uint16_t *out_hi, uint16_t *out_lo, float *stats, int N, int Ho,
                           int Wo, int C, int res_stride, int Hr, int Wr, void *stream) {
    if (!y || !scale || !shift || (!out && !out_hi) || N <= 0 || Ho <= 0 || Wo <= 0 || C < 4 || C > 1024 || (C & 3) ||
        (256 % (C / 4)) != 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc: C must be 16..1024 with C/4 dividing 256");
    if ((long long)N * Ho * Wo >= (1ll << 31)) return cer_set_error(CER_ERR_UNSUPPORTED, "bn_apply_nhwc: too many pixels");
    if ((res_hi == nullptr) != (res_lo == nullptr) || (out_hi == nullptr) != (out_lo == nullptr) || (res && res_hi))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc: split tensors need both planes; residual is fp32 OR split");
    const bool has_res = res || res_hi;
    if ((res_scale == nullptr) != (res_shift == nullptr) || (res_scale && !has_res))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc: residual affine needs res, res_scale and res_shift");
    if (has_res && (res_stride <= 0 || (Ho - 1) * res_stride >= Hr || (Wo - 1) * res_stride >= Wr))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc: residual geometry out of range");
    ApplyArgs a{y, scale, shift, alpha, res, res_scale, res_shift, mask, out, stats,
                N * Ho * Wo, Ho, Wo, C, res_stride, Hr, Wr, apply_rows_per_block(N * Ho * Wo),
                res_hi, res_lo, out_hi, out_lo};
    CER_LAUNCH(bn_apply_nhwc_kernel, dim3(cer_bn_apply_stats_tiles(a.P)), dim3(256), 0, (hipStream_t)stream, a);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_bn_apply_nhwc(const float *y, const float *scale, const float *shift, const float *alpha,
                                 const float *res, const float *res_scale, const float *res_shift, const float *mask,
                                 float *out, float *stats, int N, int Ho, int Wo, int C, int res_stride, int Hr, int Wr,
                                 void *stream) {
    if (!out) return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc: out is NULL");
    return bn_apply_launch(y, scale, shift, alpha, res, nullptr, nullptr, res_scale, res_shift, mask, out, nullptr, nullptr,
                           stats, N, Ho, Wo, C, res_stride, Hr, Wr, stream);
}

extern "C" int cer_bn_apply_nhwc_b3(const float *y, const float *scale, const float *shift, const float *alpha,
                                    const float *res, const uint16_t *res_hi, const uint16_t *res_lo, const float *res_scale,
                                    const float *res_shift, const float *mask, float *out, uint16_t *out_hi, uint16_t *out_lo,
                                    float *stats, int N, int Ho, int Wo, int C, int res_stride, int Hr, int Wr, void *stream) {
    return bn_apply_launch(y, scale, shift, alpha, res, res_hi, res_lo, res_scale, res_shift, mask, out, out_hi, out_lo, stats,
                           N, Ho, Wo, C, res_stride, Hr, Wr, stream);
}

extern "C" int cer_bn_apply_nhwc_n16(const float *y, const uint16_t *y16, const float *scale, const float *shift,
                                     const float *alpha, const float *res, const uint16_t *res16, const float *res_scale,
                                     const float *res_shift, const float *mask, float *out, uint16_t *out16, float *stats,
                                     int N, int Ho, int Wo, int C, int res_stride, int Hr, int Wr, int storage, void *stream) {
    if ((y == nullptr) == (y16 == nullptr) || !scale || !shift || (!out && !out16) || N <= 0 || Ho <= 0 || Wo <= 0 || C < 64 ||
        C > 2048 || (C & 7) || (256 % (C / 8)) != 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc_n16: one of y / y16; C must be 64..2048 with C/8 dividing 256");
    if (storage != CER_STORE_BF16 && storage != CER_STORE_F16)
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc_n16: storage must be bf16 or f16");
    if ((long long)N * Ho * Wo >= (1ll << 31)) return cer_set_error(CER_ERR_UNSUPPORTED, "bn_apply_nhwc_n16: too many pixels");
    if (res && res16) return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc_n16: residual is fp32 OR narrow");
    const bool has_res = res || res16;
    if ((res_scale == nullptr) != (res_shift == nullptr) || (res_scale && !has_res))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc_n16: residual affine needs a residual, res_scale and res_shift");
    if (has_res && (res_stride <= 0 || (Ho - 1) * res_stride >= Hr || (Wo - 1) * res_stride >= Wr))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_apply_nhwc_n16: residual geometry out of range");
    ApplyArgsN16 a{y, y16, scale, shift, alpha, res, res_scale, res_shift, mask, res16, out, stats, out16,
                   N * Ho * Wo, Ho, Wo, C, res_stride, Hr, Wr, apply_rows_per_block(N * Ho * Wo), storage};
    CER_LAUNCH(bn_apply_n16_kernel, dim3(cer_bn_apply_stats_tiles(a.P)), dim3(256), 0, (hipStream_t)stream, a);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_fold_bn_3x3(const float *w_packed, const float *scale, const float *shift, int Cout, int Cin, float *w_f32,
                               uint16_t *w_hi, uint16_t *w_lo, int storage, float *bias9, void *stream) {
    if (!w_packed || !scale || !shift || !bias9 || Cout <= 0 || Cin <= 0 || (!w_f32 && !w_hi))
        return cer_set_error(CER_ERR_INVALID_ARG, "fold_bn_3x3: bad argument");
    if (storage != CER_STORE_NONE && storage != CER_STORE_BF16 && storage != CER_STORE_F16)
        return cer_set_error(CER_ERR_INVALID_ARG, "fold_bn_3x3: unknown storage");
    if (w_hi && ((storage == CER_STORE_NONE) != (w_lo != nullptr)))
        return cer_set_error(CER_ERR_INVALID_ARG, "fold_bn_3x3: split output needs w_hi and w_lo, narrow output w_hi only");
    CER_LAUNCH(fold_bn_3x3_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w_packed, scale, shift, Cout, Cin,
               cer_conv_kpad(3, 3, Cin), w_f32, w_hi, w_lo, storage, bias9);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
