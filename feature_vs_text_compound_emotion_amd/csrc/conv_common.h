// conv_common.h -- argument block and fused epilogue shared by the fp32 and the bf16x3
// implicit-GEMM kernels (conv_igemm.hip, conv_b3.hip) and by the split-K reducer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cer_internal.h"

namespace cer {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;  // K padding granule of packed weights ([Cout][Kpad], Kpad % 32 == 0)

struct ConvArgs {
    const float *x, *w, *in_scale, *in_shift, *bias, *alpha, *res, *mask;
    float *y;        // output, or split-K partial slabs [split][M][Cout]
    float *aux;      // optional [M][Cout]: mask*act1(conv+bias), i.e. the value before the residual add
    float *stats;    // optional [tiles_m][2][Cout]: per-tile sum / sum of squares of the RAW conv result
    // ---- split-bf16 ("bf16x3") operands and outputs: a value v is carried as hi = bf16(v),
    // lo = bf16(v - hi); products use hi*hi + hi*lo + lo*hi on the bf16 matrix cores ----
    const uint16_t *x_hi, *x_lo, *w_hi, *w_lo;  // b3 kernel inputs (x/w above are unused then)
    const uint16_t *res_hi, *res_lo;            // residual given as a split tensor (alternative to res)
    uint16_t *y_hi, *y_lo;                      // optional split copy of the output
    const float *bias9;                         // optional [9][Cout] border-dependent bias (replaces bias)
    const float *s2, *t2;                       // optional second output: out * s2[c] + t2[c] ...
    uint16_t *y2_hi, *y2_lo;                    // ... stored split (the NEXT layer's pre-conv BatchNorm)
    int x_ld, y_ld;  // row pitches (elements) of x pixels / y rows
    int N, H, W, Cin, Ho, Wo, Cout;
    int KH, KW, stride, dil_h, dil_w, pad_t, pad_l;
    int x_nchw, res_stride, Hr, Wr, act1, act2;
    float slope;
    int Kpad, M, tiles_m, tiles_n, steps, steps_per_split, split_k, cin_steps;
    // narrow storage (cer_conv_desc.storage): 0 = fp32 / split tensors as the pointers say; CER_STORE_BF16 / CER_STORE_F16 =
    // every 16-bit tensor of the launch (x_hi, w_hi, res_hi, y_hi) is ONE plane of that type and the *_lo pointers are NULL
    int narrow;
    // cer_conv_desc.y_s2d: the 16-bit output planes are written space-to-depth (row permutation s2d_row() of the stores);
    // x_s2d: the input is such a tensor (x_ld = 4 Cin), conv_b3_s2d.hip
    int y_s2d, x_s2d;
};

// Row of output pixel m = (n, ho, wo) in the space-to-depth view [N * Ho/2 * Wo/2 * 4][C] of an [N, Ho, Wo, C] tensor:
// 4 * ((n * Ho/2 + ho/2) * Wo/2 + wo/2) + blk with blk = 3 - 2 (ho & 1) - (wo & 1) (block order P11 | P10 | P01 | P00, the
// order conv_b3_s2d_kernel walks the phases in).  Ho and Wo are even.
__host__ __device__ __forceinline__ int s2d_row(int m, int ho, int wo, int Wo) {
    return (m - ho * Wo - wo) + 4 * ((ho >> 1) * (Wo >> 1) + (wo >> 1)) + 3 - 2 * (ho & 1) - (wo & 1);
}

// Division of block / pixel indices by launch constants (the kernels' prologues: an emulated 32-bit division is ~25
// instructions, and a 16x16-patch block needs four of them before it can issue its first load).  magic = 2^32 / d + 1 makes
// umulhi(x, magic) the quotient or the quotient + 1 for every 32-bit x; one compare corrects it.
struct FastDiv { unsigned d, magic; };
inline FastDiv make_fastdiv(unsigned d) { return FastDiv{d, d > 1 ? (unsigned)((1ull << 32) / d) + 1u : 0u}; }
__device__ __forceinline__ unsigned fdiv(unsigned x, const FastDiv f) {
    if (f.d == 1) return x;
    const unsigned q = __umulhi(x, f.magic);
    return q - (q * f.d > x ? 1u : 0u);
}

// block -> (pixel tile, cout tile), pixel -> (image row, column): the window kernels' launch constants
struct WinGeo { FastDiv tiles_n, hw, w; };

template <int I> struct IdxC { static constexpr int v = I; };
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(IdxC<N - 1>{});
    }
}

// bf16 <-> f32 bit helpers; the split is round-to-nearest-even (the hardware cvt) on both parts
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ void split_bf16(float v, uint16_t &hi, uint16_t &lo) {
    hi = f32_to_bf16(v);
    lo = f32_to_bf16(v - bf16_to_f32(hi));
}
// one 16-bit plane: bf16 (CER_STORE_BF16) or IEEE half (CER_STORE_F16), round-to-nearest-even
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f32_to_n16(float f, int narrow) { return narrow == CER_STORE_F16 ? f32_to_f16(f) : f32_to_bf16(f); }
__device__ __forceinline__ float n16_to_f32(uint16_t h, int narrow) { return narrow == CER_STORE_F16 ? f16_to_f32(h) : bf16_to_f32(h); }
__device__ __forceinline__ void store_narrow4(uint16_t *dst, const float o[4], int narrow) {
    ushort4 h;
    if (narrow == CER_STORE_F16) {
        h.x = f32_to_f16(o[0]); h.y = f32_to_f16(o[1]); h.z = f32_to_f16(o[2]); h.w = f32_to_f16(o[3]);
    } else {
        h.x = f32_to_bf16(o[0]); h.y = f32_to_bf16(o[1]); h.z = f32_to_bf16(o[2]); h.w = f32_to_bf16(o[3]);
    }
    *reinterpret_cast<ushort4 *>(dst) = h;
}
__device__ __forceinline__ void load_narrow4(const uint16_t *src, float o[4], int narrow) {
    const ushort4 h = *reinterpret_cast<const ushort4 *>(src);
    if (narrow == CER_STORE_F16) {
        o[0] = f16_to_f32(h.x); o[1] = f16_to_f32(h.y); o[2] = f16_to_f32(h.z); o[3] = f16_to_f32(h.w);
    } else {
        o[0] = bf16_to_f32(h.x); o[1] = bf16_to_f32(h.y); o[2] = bf16_to_f32(h.z); o[3] = bf16_to_f32(h.w);
    }
}

__device__ __forceinline__ void store_split4(uint16_t *hi, uint16_t *lo, const float o[4]) {
    ushort4 h, l;
    split_bf16(o[0], h.x, l.x); split_bf16(o[1], h.y, l.y);
    split_bf16(o[2], h.z, l.z); split_bf16(o[3], h.w, l.w);
    *reinterpret_cast<ushort4 *>(hi) = h;
    *reinterpret_cast<ushort4 *>(lo) = l;
}

__device__ __forceinline__ float act_apply(float v, int act, float a, float slope) {
    switch (act) {
        case CER_ACT_PRELU: return v >= 0.f ? v : v * a;
        case CER_ACT_LEAKY: return v >= 0.f ? v : v * slope;
        case CER_ACT_RELU: return v > 0.f ? v : 0.f;
        case CER_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        default: return v;
    }
}

// Shared epilogue for the fused path and the split-K reducer.
// v: 4 consecutive couts starting at c for output row m.
// Row / column state of output pixel m for bias9 (0 first, 2 last, 1 inner): index 3*ry + rx.
__device__ __forceinline__ int bias9_case(const ConvArgs &p, int m) {
    const int hw = p.Ho * p.Wo;
    const int r = m % hw;
    const int ho = r / p.Wo, wo = r - ho * p.Wo;
    const int ry = ho == 0 ? 0 : (ho == p.Ho - 1 ? 2 : 1), rx = wo == 0 ? 0 : (wo == p.Wo - 1 ? 2 : 1);
    return 3 * ry + rx;
}

// bias_row: the bias vector to use for this pixel (p.bias, or the caller's pre-selected bias9 row)
__device__ __forceinline__ void epilogue_store4(const ConvArgs &p, int m, int c, float v[4], const float *bias);

__device__ __forceinline__ void epilogue_store4(const ConvArgs &p, int m, int c, float v[4]) {
    epilogue_store4(p, m, c, v, p.bias9 ? p.bias9 + (size_t)bias9_case(p, m) * p.Cout : p.bias);
}

__device__ __forceinline__ void epilogue_store4(const ConvArgs &p, int m, int c, float v[4], const float *bias) {
    const bool vec = ((p.Cout & 3) == 0);
    size_t roff = 0;
    if (p.res || p.res_hi) {
        if (p.res_stride == 1 && p.Hr == p.Ho && p.Wr == p.Wo) {
            roff = (size_t)m * p.Cout;
        } else {
            int hw = p.Ho * p.Wo;
            int n = m / hw, r = m - n * hw;
            int ho = r / p.Wo, wo = r - ho * p.Wo;
            roff = ((size_t)(n * p.Hr + ho * p.res_stride) * p.Wr + wo * p.res_stride) * p.Cout;
        }
    }
    const size_t yoff = (size_t)m * p.y_ld + c;
    const size_t doff = (size_t)m * p.Cout + c;  // dense offset (mask, aux)
    if (vec && ((p.y_ld & 3) == 0) && c + 3 < p.Cout) {
        float4 b = bias ? *reinterpret_cast<const float4 *>(bias + c) : make_float4(0, 0, 0, 0);
        float4 a = (p.act1 == CER_ACT_PRELU) ? *reinterpret_cast<const float4 *>(p.alpha + c) : make_float4(0, 0, 0, 0);
        float bb[4] = {b.x, b.y, b.z, b.w}, aa[4] = {a.x, a.y, a.z, a.w};
        float rr[4] = {0, 0, 0, 0}, mm[4] = {1, 1, 1, 1};
        if (p.res) {
            float4 r = *reinterpret_cast<const float4 *>(p.res + roff + c);
            rr[0] = r.x; rr[1] = r.y; rr[2] = r.z; rr[3] = r.w;
        } else if (p.res_hi && p.narrow) {
            load_narrow4(p.res_hi + roff + c, rr, p.narrow);
        } else if (p.res_hi) {
            const ushort4 h = *reinterpret_cast<const ushort4 *>(p.res_hi + roff + c);
            const ushort4 l = *reinterpret_cast<const ushort4 *>(p.res_lo + roff + c);
            rr[0] = bf16_to_f32(h.x) + bf16_to_f32(l.x); rr[1] = bf16_to_f32(h.y) + bf16_to_f32(l.y);
            rr[2] = bf16_to_f32(h.z) + bf16_to_f32(l.z); rr[3] = bf16_to_f32(h.w) + bf16_to_f32(l.w);
        }
        if (p.mask) {
            float4 k = *reinterpret_cast<const float4 *>(p.mask + doff);
            mm[0] = k.x; mm[1] = k.y; mm[2] = k.z; mm[3] = k.w;
        }
        float o[4], u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = act_apply(v[e] + bb[e], p.act1, aa[e], p.slope) * mm[e];
            u[e] = t;
            o[e] = act_apply(t + rr[e], p.act2, 0.f, p.slope);
        }
        if (p.aux) *reinterpret_cast<float4 *>(p.aux + doff) = make_float4(u[0], u[1], u[2], u[3]);
        if (p.y) *reinterpret_cast<float4 *>(p.y + yoff) = make_float4(o[0], o[1], o[2], o[3]);
        if (p.y_hi) {
            if (p.narrow) store_narrow4(p.y_hi + yoff, o, p.narrow);
            else store_split4(p.y_hi + yoff, p.y_lo + yoff, o);
        }
        if (p.y2_hi) {
            const float4 s2 = *reinterpret_cast<const float4 *>(p.s2 + c), t2 = *reinterpret_cast<const float4 *>(p.t2 + c);
            float o2[4] = {o[0] * s2.x + t2.x, o[1] * s2.y + t2.y, o[2] * s2.z + t2.z, o[3] * s2.w + t2.w};
            store_split4(p.y2_hi + yoff, p.y2_lo + yoff, o2);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c + e < p.Cout) {
                float t = v[e] + (bias ? bias[c + e] : 0.f);
                t = act_apply(t, p.act1, p.act1 == CER_ACT_PRELU ? p.alpha[c + e] : 0.f, p.slope);
                if (p.mask) t *= p.mask[doff + e];
                if (p.aux) p.aux[doff + e] = t;
                if (p.res) t += p.res[roff + c + e];
                else if (p.res_hi && p.narrow) t += n16_to_f32(p.res_hi[roff + c + e], p.narrow);
                else if (p.res_hi) t += bf16_to_f32(p.res_hi[roff + c + e]) + bf16_to_f32(p.res_lo[roff + c + e]);
                const float o = act_apply(t, p.act2, 0.f, p.slope);
                if (p.y) p.y[yoff + e] = o;
                if (p.y_hi && p.narrow) p.y_hi[yoff + e] = f32_to_n16(o, p.narrow);
                else if (p.y_hi) split_bf16(o, p.y_hi[yoff + e], p.y_lo[yoff + e]);
                if (p.y2_hi) split_bf16(o * p.s2[c + e] + p.t2[c + e], p.y2_hi[yoff + e], p.y2_lo[yoff + e]);
            }
        }
    }
}

// ---- streamlined epilogue for the staged (LDS -> compact row loop) epilogues ----
// epilogue_store4 above is fully general and costs ~2600 instructions of branch skeleton per call (two activation switches with
// an erf expansion per element, every optional tensor tested per element); called once per 4 couts it made the store phase of
// a 256 x 128 tile take 12 us -- more than the tile's MFMA time on the K-short layers.  The encoder's launches use a small
// subset: bias / border bias (bias9), PReLU or ReLU or no activation, an optional residual, fp32 / split / narrow outputs.
// EpiCtx decides ONCE per launch (uniformly) whether that subset applies, keeps the thread's per-channel constants in
// registers (no loads in the row loop besides the residual: a load's vmcnt wait would also wait for the previous rows'
// stores), and epi_store4 is then ~40 instructions; anything else falls back to epilogue_store4.
__host__ __device__ __forceinline__ int epi_mode(const ConvArgs &p);

struct EpiCtx {
    bool fast;
    int mode;         // epi_mode(): EPI_GENERIC or one of the specialised row epilogues
    float aa[4];      // PReLU slopes of the thread's 4 couts
    float b9[9][4];   // bias rows of the thread's 4 couts: [0] = p.bias (or zeros), all nine when p.bias9
};

__device__ __forceinline__ void epi_init(const ConvArgs &p, int c, EpiCtx &e) {
    e.fast = (p.Cout & 3) == 0 && (p.y_ld & 3) == 0 && p.act2 == CER_ACT_NONE &&
             (p.act1 == CER_ACT_NONE || p.act1 == CER_ACT_PRELU || p.act1 == CER_ACT_RELU) && !p.mask && !p.aux && !p.y2_hi;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) e.b9[k][t] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) e.aa[t] = 0.f;
    e.mode = epi_mode(p);
    if (!e.fast || c >= p.Cout) return;
    if (p.bias9) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float4 b = *reinterpret_cast<const float4 *>(p.bias9 + (size_t)k * p.Cout + c);
            e.b9[k][0] = b.x; e.b9[k][1] = b.y; e.b9[k][2] = b.z; e.b9[k][3] = b.w;
        }
    } else if (p.bias) {
        const float4 b = *reinterpret_cast<const float4 *>(p.bias + c);
        e.b9[0][0] = b.x; e.b9[0][1] = b.y; e.b9[0][2] = b.z; e.b9[0][3] = b.w;
    }
    if (p.act1 == CER_ACT_PRELU) {
        const float4 a = *reinterpret_cast<const float4 *>(p.alpha + c);
        e.aa[0] = a.x; e.aa[1] = a.y; e.aa[2] = a.z; e.aa[3] = a.w;
    }
}

// cs: the pixel's bias9 case (3 * ry + rx), ignored without bias9
__device__ __forceinline__ void epi_store4(const ConvArgs &p, const EpiCtx &e, int m, int c, float v[4], int cs) {
    if (!e.fast) {
        epilogue_store4(p, m, c, v, p.bias9 ? p.bias9 + (size_t)cs * p.Cout : p.bias);
        return;
    }
    float o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float b = e.b9[0][t];
        if (p.bias9) {
#pragma unroll
            for (int k = 1; k < 9; ++k) b = cs == k ? e.b9[k][t] : b;
        }
        o[t] = v[t] + b;
    }
    if (p.act1 == CER_ACT_PRELU) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = o[t] >= 0.f ? o[t] : o[t] * e.aa[t];
    } else if (p.act1 == CER_ACT_RELU) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = o[t] > 0.f ? o[t] : 0.f;
    }
    if (p.res || p.res_hi) {
        size_t roff;
        if (p.res_stride == 1 && p.Hr == p.Ho && p.Wr == p.Wo) {
            roff = (size_t)m * p.Cout + c;
        } else {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, r = m - n * hw;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            roff = ((size_t)(n * p.Hr + ho * p.res_stride) * p.Wr + wo * p.res_stride) * p.Cout + c;
        }
        float rr[4];
        if (p.res) {
            const float4 r = *reinterpret_cast<const float4 *>(p.res + roff);
            rr[0] = r.x; rr[1] = r.y; rr[2] = r.z; rr[3] = r.w;
        } else if (p.narrow) {
            load_narrow4(p.res_hi + roff, rr, p.narrow);
        } else {
            const ushort4 h = *reinterpret_cast<const ushort4 *>(p.res_hi + roff);
            const ushort4 l = *reinterpret_cast<const ushort4 *>(p.res_lo + roff);
            rr[0] = bf16_to_f32(h.x) + bf16_to_f32(l.x); rr[1] = bf16_to_f32(h.y) + bf16_to_f32(l.y);
            rr[2] = bf16_to_f32(h.z) + bf16_to_f32(l.z); rr[3] = bf16_to_f32(h.w) + bf16_to_f32(l.w);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] += rr[t];
    }
    const size_t yoff = (size_t)m * p.y_ld + c;
    if (p.y) *reinterpret_cast<float4 *>(p.y + yoff) = make_float4(o[0], o[1], o[2], o[3]);
    if (p.y_hi) {
        if (p.narrow) store_narrow4(p.y_hi + yoff, o, p.narrow);
        else store_split4(p.y_hi + yoff, p.y_lo + yoff, o);
    }
}


// epi_store4's fast path for kernels whose threads call the epilogue with MANY channel groups (the fp32 igemm kernel: 8-16
// groups of 4 couts per thread, so an EpiCtx per group would not fit the registers): the per-channel constants are re-loaded
// per call (L1 hits).  The Cin = 3 stem at 224x224 spent its time in epilogue_store4's branch skeleton (~2 600 instructions
// per call, 8 calls per thread, 400 000 tiles): 10.5 ms for a launch whose HBM floor is 2.8 ms.
__device__ __forceinline__ bool epi_fast(const ConvArgs &p) {
    return (p.Cout & 3) == 0 && (p.y_ld & 3) == 0 && p.act2 == CER_ACT_NONE &&
           (p.act1 == CER_ACT_NONE || p.act1 == CER_ACT_PRELU || p.act1 == CER_ACT_RELU) && !p.mask && !p.aux && !p.y2_hi;
}

__device__ __forceinline__ void epi_store4_direct(const ConvArgs &p, bool fast, int m, int c, float v[4]) {
    if (!fast || c + 3 >= p.Cout) {
        epilogue_store4(p, m, c, v);
        return;
    }
    float o[4] = {v[0], v[1], v[2], v[3]};
    const float *bias = p.bias9 ? p.bias9 + (size_t)bias9_case(p, m) * p.Cout : p.bias;
    if (bias) {
        const float4 b = *reinterpret_cast<const float4 *>(bias + c);
        o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w;
    }
    if (p.act1 == CER_ACT_PRELU) {
        const float4 a = *reinterpret_cast<const float4 *>(p.alpha + c);
        o[0] = o[0] >= 0.f ? o[0] : o[0] * a.x; o[1] = o[1] >= 0.f ? o[1] : o[1] * a.y;
        o[2] = o[2] >= 0.f ? o[2] : o[2] * a.z; o[3] = o[3] >= 0.f ? o[3] : o[3] * a.w;
    } else if (p.act1 == CER_ACT_RELU) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = o[t] > 0.f ? o[t] : 0.f;
    }
    if (p.res || p.res_hi) {
        size_t roff;
        if (p.res_stride == 1 && p.Hr == p.Ho && p.Wr == p.Wo) {
            roff = (size_t)m * p.Cout + c;
        } else {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, r = m - n * hw;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            roff = ((size_t)(n * p.Hr + ho * p.res_stride) * p.Wr + wo * p.res_stride) * p.Cout + c;
        }
        float rr[4];
        if (p.res) {
            const float4 r = *reinterpret_cast<const float4 *>(p.res + roff);
            rr[0] = r.x; rr[1] = r.y; rr[2] = r.z; rr[3] = r.w;
        } else if (p.narrow) {
            load_narrow4(p.res_hi + roff, rr, p.narrow);
        } else {
            const ushort4 h = *reinterpret_cast<const ushort4 *>(p.res_hi + roff);
            const ushort4 l = *reinterpret_cast<const ushort4 *>(p.res_lo + roff);
            rr[0] = bf16_to_f32(h.x) + bf16_to_f32(l.x); rr[1] = bf16_to_f32(h.y) + bf16_to_f32(l.y);
            rr[2] = bf16_to_f32(h.z) + bf16_to_f32(l.z); rr[3] = bf16_to_f32(h.w) + bf16_to_f32(l.w);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] += rr[t];
    }
    const size_t yoff = (size_t)m * p.y_ld + c;
    if (p.y) *reinterpret_cast<float4 *>(p.y + yoff) = make_float4(o[0], o[1], o[2], o[3]);
    if (p.y_hi) {
        if (p.narrow) store_narrow4(p.y_hi + yoff, o, p.narrow);
        else store_split4(p.y_hi + yoff, p.y_lo + yoff, o);
    }
}

// ---- fully specialised row epilogues for the encoder's launches ----
// Even the streamlined epi_store4 spends ~20 uniform branches per 4 couts.  The launches that carry the encoder's time use six
// exact combinations; epi_mode names them once per launch and epi_row<MODE> is straight-line code (~25 instructions):
//   RAW_F32 / RAW_N16         raw conv result -> fp32 / narrow        (batch-statistics conv2 and shortcut convs, + stats)
//   B9_PRELU_SPLIT / _N16     + border bias, PReLU -> split / narrow  (conv1 of every unit: its input BatchNorm is folded)
//   BIAS_RES_SPLIT / _N16     + bias + same-geometry residual         (conv2 of a unit with running statistics)
// Everything else is EPI_GENERIC = epi_store4.  The interior bias row (case 4) is the common one: the other eight are selected
// only in waves that hold a border pixel.
enum { EPI_GENERIC = 0, EPI_RAW_F32, EPI_RAW_N16, EPI_B9_PRELU_SPLIT, EPI_B9_PRELU_N16, EPI_BIAS_RES_SPLIT, EPI_BIAS_RES_N16 };

__host__ __device__ __forceinline__ int epi_mode(const ConvArgs &p) {
    if ((p.Cout & 3) || p.y_ld != p.Cout || p.act2 != CER_ACT_NONE || p.mask || p.aux || p.y2_hi || p.res) return EPI_GENERIC;
    const bool same_res = p.res_hi && p.res_stride == 1 && p.Hr == p.Ho && p.Wr == p.Wo;
    if (!p.bias && !p.bias9 && p.act1 == CER_ACT_NONE && !p.res_hi) {
        if (p.y && !p.y_hi) return EPI_RAW_F32;
        if (!p.y && p.y_hi && p.narrow) return EPI_RAW_N16;
    }
    if (p.bias9 && p.act1 == CER_ACT_PRELU && !p.res_hi && !p.y && p.y_hi) return p.narrow ? EPI_B9_PRELU_N16 : EPI_B9_PRELU_SPLIT;
    if (p.bias && !p.bias9 && p.act1 == CER_ACT_NONE && same_res && !p.y && p.y_hi) return p.narrow ? EPI_BIAS_RES_N16 : EPI_BIAS_RES_SPLIT;
    return EPI_GENERIC;
}

template <class F> __device__ __forceinline__ void epi_dispatch(int mode, F &&f) {
    switch (mode) {
        case EPI_RAW_F32: f(IdxC<EPI_RAW_F32>{}); break;
        case EPI_RAW_N16: f(IdxC<EPI_RAW_N16>{}); break;
        case EPI_B9_PRELU_SPLIT: f(IdxC<EPI_B9_PRELU_SPLIT>{}); break;
        case EPI_B9_PRELU_N16: f(IdxC<EPI_B9_PRELU_N16>{}); break;
        case EPI_BIAS_RES_SPLIT: f(IdxC<EPI_BIAS_RES_SPLIT>{}); break;
        case EPI_BIAS_RES_N16: f(IdxC<EPI_BIAS_RES_N16>{}); break;
        default: f(IdxC<EPI_GENERIC>{}); break;
    }
}

// the specialised modes on an already chosen bias vector b4 (ignored by the RAW modes) and PReLU slopes aa
template <int MODE>
__device__ __forceinline__ void epi_row_core(const ConvArgs &p, const float aa[4], const float b4[4], int m, int c, const float v[4]) {
    static_assert(MODE != EPI_GENERIC, "specialised modes only");
    const size_t off = (size_t)m * p.Cout + c;
    float o[4] = {v[0], v[1], v[2], v[3]};
    if constexpr (MODE == EPI_B9_PRELU_SPLIT || MODE == EPI_B9_PRELU_N16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            o[t] += b4[t];
            o[t] = o[t] >= 0.f ? o[t] : o[t] * aa[t];
        }
    }
    if constexpr (MODE == EPI_BIAS_RES_SPLIT || MODE == EPI_BIAS_RES_N16) {
        float rr[4];
        if constexpr (MODE == EPI_BIAS_RES_N16) {
            load_narrow4(p.res_hi + off, rr, p.narrow);
        } else {
            const ushort4 h = *reinterpret_cast<const ushort4 *>(p.res_hi + off);
            const ushort4 l = *reinterpret_cast<const ushort4 *>(p.res_lo + off);
            rr[0] = bf16_to_f32(h.x) + bf16_to_f32(l.x); rr[1] = bf16_to_f32(h.y) + bf16_to_f32(l.y);
            rr[2] = bf16_to_f32(h.z) + bf16_to_f32(l.z); rr[3] = bf16_to_f32(h.w) + bf16_to_f32(l.w);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = (o[t] + b4[t]) + rr[t];
    }
    if constexpr (MODE == EPI_RAW_F32) {
        *reinterpret_cast<float4 *>(p.y + off) = make_float4(o[0], o[1], o[2], o[3]);
    } else if constexpr (MODE == EPI_RAW_N16 || MODE == EPI_B9_PRELU_N16 || MODE == EPI_BIAS_RES_N16) {
        store_narrow4(p.y_hi + off, o, p.narrow);
    } else {
        store_split4(p.y_hi + off, p.y_lo + off, o);
    }
}

template <int MODE>
__device__ __forceinline__ void epi_row(const ConvArgs &p, const EpiCtx &e, int m, int c, float v[4], int cs) {
    if constexpr (MODE == EPI_GENERIC) {
        epi_store4(p, e, m, c, v, cs);
    } else {
        float b[4] = {e.b9[0][0], e.b9[0][1], e.b9[0][2], e.b9[0][3]};
        if constexpr (MODE == EPI_B9_PRELU_SPLIT || MODE == EPI_B9_PRELU_N16) {
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = e.b9[4][t];
            if (cs != 4) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    b[t] = e.b9[0][t];
#pragma unroll
                    for (int k = 1; k < 9; ++k) b[t] = cs == k ? e.b9[k][t] : b[t];
                }
            }
        }
        epi_row_core<MODE>(p, e.aa, b, m, c, v);
    }
}

// ---- direct epilogue: accumulators -> global memory without the LDS round trip ----
// In-kernel stamps of the 64 -> 64 @224x224 launch (tools/exp_stamp.py, profiles/round3_stamps_*.txt): of a block's 15.3 us,
// 1.5 us went into writing the accumulators to LDS (ds_write_b128 runs at a third of the read rate), 3.2 us into the row
// loop that read them back one 16-byte granule per iteration, and the K loop itself took 5.3 us.  The kernels whose wave tile
// is [16 TC couts] x [16 TP pixels] with the WEIGHTS as the MFMA A operand hold, per lane, pixel l15 of every pixel tile and
// rows 4 kg .. 4 kg + 3 of every cout tile.  Dealing the weight rows to the tile rows as
//     tile a, tile row 4 kg + r   <->   cout  32 (a / 2) + 8 kg + 4 (a % 2) + r   of the wave's 16 TC couts
// makes the accumulators of a tile PAIR 8 consecutive couts of one pixel: one 16-byte store per 16-bit plane (two for fp32),
// and the four kg lanes of a pixel cover 64 contiguous bytes.  Only the permutation of the weight rows in the DMA addresses
// differs from the plain layout; the per-cout batch statistics are reduced over the 16 pixel lanes with DPP row rotations.
__host__ __device__ constexpr int epi_cout_of_row(int rho) {   // rho = 16 a + 4 kg + r, within a wave's span of 16 TC rows
    return ((rho >> 5) << 5) + (((rho >> 2) & 3) << 3) + (((rho >> 4) & 1) << 2) + (rho & 3);
}

__device__ __forceinline__ float row16_sum(float v) {      // total of the 16 lanes of a DPP row, in every lane of the row
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

template <int NARROW> __device__ __forceinline__ uint32_t pack2_n16(float a, float b) {
    if constexpr (NARROW == CER_STORE_F16) return (uint32_t)f32_to_f16(a) | ((uint32_t)f32_to_f16(b) << 16);
    else return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}
template <int NARROW> __device__ __forceinline__ void unpack8_n16(const uint4 q, float o[8]) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (NARROW == CER_STORE_F16) {
            o[2 * i] = f16_to_f32((uint16_t)(w[i] & 0xffffu));
            o[2 * i + 1] = f16_to_f32((uint16_t)(w[i] >> 16));
        } else {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
}

// One pixel x 8 consecutive couts (c .. c + 7 < Cout) of a specialised mode.  row: the output row (the pixel, or its
// space-to-depth position); cs: the pixel's border case for the bias9 modes; aa / bb: PReLU slopes and the interior (or plain)
// bias of the 8 couts.  NARROW: CER_STORE_BF16 / CER_STORE_F16 for the *_N16 modes, ignored otherwise.
// BORDER = false: the caller knows (wave-uniformly) that no lane of the wave holds a border pixel -- the bias9 modes then contain
// no load at all.  It matters: the compiler places the s_waitcnt vmcnt(0) of the border rows' loads AFTER the divergent branch,
// where every wave executes it, loads or not -- and on gfx9 that one counter also covers the stores, so every 16-byte store of the
// epilogue waited for the previous one to be acknowledged (and for the window DMA in flight).
template <int MODE, int NARROW, bool BORDER = true>
__device__ __forceinline__ void epi_direct8(const ConvArgs &p, const float (&aa)[8], const float (&bb)[8], size_t row, int c, int cs,
                                            const float (&v)[8]) {
    static_assert(MODE != EPI_GENERIC, "specialised modes only");
    const size_t off = row * (size_t)p.Cout + c;
    float o[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = v[t];
    if constexpr (MODE == EPI_B9_PRELU_SPLIT || MODE == EPI_B9_PRELU_N16) {
        float b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) b[t] = bb[t];
        if constexpr (BORDER) {
            if (cs != 4) {   // a border pixel: its own bias row (L1 / L2 hits; one lane in eight at most on the 16x16 patches)
                const float4 q0 = *reinterpret_cast<const float4 *>(p.bias9 + (size_t)cs * p.Cout + c);
                const float4 q1 = *reinterpret_cast<const float4 *>(p.bias9 + (size_t)cs * p.Cout + c + 4);
                b[0] = q0.x; b[1] = q0.y; b[2] = q0.z; b[3] = q0.w; b[4] = q1.x; b[5] = q1.y; b[6] = q1.z; b[7] = q1.w;
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            o[t] += b[t];
            o[t] = o[t] >= 0.f ? o[t] : o[t] * aa[t];
        }
    }
    if constexpr (MODE == EPI_BIAS_RES_SPLIT || MODE == EPI_BIAS_RES_N16) {
        float rr[8];
        if constexpr (MODE == EPI_BIAS_RES_N16) {
            unpack8_n16<NARROW>(*reinterpret_cast<const uint4 *>(p.res_hi + off), rr);
        } else {
            float lo[8];
            unpack8_n16<CER_STORE_BF16>(*reinterpret_cast<const uint4 *>(p.res_hi + off), rr);
            unpack8_n16<CER_STORE_BF16>(*reinterpret_cast<const uint4 *>(p.res_lo + off), lo);
#pragma unroll
            for (int t = 0; t < 8; ++t) rr[t] += lo[t];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = (o[t] + bb[t]) + rr[t];
    }
    if constexpr (MODE == EPI_RAW_F32) {
        *reinterpret_cast<float4 *>(p.y + off) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(p.y + off + 4) = make_float4(o[4], o[5], o[6], o[7]);
    } else if constexpr (MODE == EPI_RAW_N16 || MODE == EPI_B9_PRELU_N16 || MODE == EPI_BIAS_RES_N16) {
        uint4 q;
        q.x = pack2_n16<NARROW>(o[0], o[1]); q.y = pack2_n16<NARROW>(o[2], o[3]);
        q.z = pack2_n16<NARROW>(o[4], o[5]); q.w = pack2_n16<NARROW>(o[6], o[7]);
        *reinterpret_cast<uint4 *>(p.y_hi + off) = q;
    } else {
        uint16_t h[8], l[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) split_bf16(o[t], h[t], l[t]);
        uint4 qh, ql;
        qh.x = h[0] | ((uint32_t)h[1] << 16); qh.y = h[2] | ((uint32_t)h[3] << 16); qh.z = h[4] | ((uint32_t)h[5] << 16); qh.w = h[6] | ((uint32_t)h[7] << 16);
        ql.x = l[0] | ((uint32_t)l[1] << 16); ql.y = l[2] | ((uint32_t)l[3] << 16); ql.z = l[4] | ((uint32_t)l[5] << 16); ql.w = l[6] | ((uint32_t)l[7] << 16);
        *reinterpret_cast<uint4 *>(p.y_hi + off) = qh;
        *reinterpret_cast<uint4 *>(p.y_lo + off) = ql;
    }
}

// The slopes / bias of the lane's 8 couts starting at c (zeros where the mode does not use them or past Cout)
template <int MODE>
__device__ __forceinline__ void epi_direct_consts(const ConvArgs &p, int c, float (&aa)[8], float (&bb)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) aa[t] = bb[t] = 0.f;
    if (c + 7 >= p.Cout) return;
    const float *bias = nullptr;
    if constexpr (MODE == EPI_B9_PRELU_SPLIT || MODE == EPI_B9_PRELU_N16) bias = p.bias9 + (size_t)4 * p.Cout;
    if constexpr (MODE == EPI_BIAS_RES_SPLIT || MODE == EPI_BIAS_RES_N16) bias = p.bias;
    if (bias) {
        const float4 q0 = *reinterpret_cast<const float4 *>(bias + c), q1 = *reinterpret_cast<const float4 *>(bias + c + 4);
        bb[0] = q0.x; bb[1] = q0.y; bb[2] = q0.z; bb[3] = q0.w; bb[4] = q1.x; bb[5] = q1.y; bb[6] = q1.z; bb[7] = q1.w;
    }
    if constexpr (MODE == EPI_B9_PRELU_SPLIT || MODE == EPI_B9_PRELU_N16) {
        const float4 q0 = *reinterpret_cast<const float4 *>(p.alpha + c), q1 = *reinterpret_cast<const float4 *>(p.alpha + c + 4);
        aa[0] = q0.x; aa[1] = q0.y; aa[2] = q0.z; aa[3] = q0.w; aa[4] = q1.x; aa[5] = q1.y; aa[6] = q1.z; aa[7] = q1.w;
    }
}


// The lane's pixel of one pixel tile: does it exist, its output row (the pixel, or its space-to-depth position), its border case
struct EpiPix { bool live; size_t row; int cs; };

// All accumulators of a wave tile [16 TC couts] x [16 TP pixels] -> global memory in a specialised MODE, and the lane's partial
// batch statistics (sums over its TP pixels of the RAW conv results of its 8 TC / 2 couts).  cw: the wave's first cout.
template <int MODE, int NARROW, int TC, int TP, class ACC>
__device__ __forceinline__ void epi_direct_stores(const ConvArgs &p, const ACC (&acc)[TC][TP], int cw, int kg, const EpiPix (&px)[TP],
                                                  float (&s1)[TC / 2][8], float (&s2)[TC / 2][8]) {
    static_assert(TC % 2 == 0, "cout tiles come in pairs (epi_cout_of_row)");
    static_for<TC / 2>([&](auto J) {
        constexpr int j = decltype(J)::v;
        const int c = cw + j * 32 + kg * 8;
        float aa[8], bb[8];
        epi_direct_consts<MODE>(p, c, aa, bb);
#pragma unroll
        for (int t = 0; t < 8; ++t) s1[j][t] = s2[j][t] = 0.f;
        if (c + 7 < p.Cout) {
            static_for<TP>([&](auto B) {
                constexpr int b = decltype(B)::v;
                if (px[b].live) {
                    const float v[8] = {acc[2 * j][b][0], acc[2 * j][b][1], acc[2 * j][b][2], acc[2 * j][b][3],
                                        acc[2 * j + 1][b][0], acc[2 * j + 1][b][1], acc[2 * j + 1][b][2], acc[2 * j + 1][b][3]};
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        s1[j][t] += v[t];
                        s2[j][t] += v[t] * v[t];
                    }
                    if constexpr (MODE == EPI_B9_PRELU_SPLIT || MODE == EPI_B9_PRELU_N16) {
                        // (wave-uniform: does any lane of the wave hold a border pixel of this pixel tile?)
                        if (__ballot(px[b].cs != 4) != 0ull) epi_direct8<MODE, NARROW, true>(p, aa, bb, px[b].row, c, px[b].cs, v);
                        else epi_direct8<MODE, NARROW, false>(p, aa, bb, px[b].row, c, 4, v);
                    } else {
                        epi_direct8<MODE, NARROW, false>(p, aa, bb, px[b].row, c, px[b].cs, v);
                    }
                }
            });
        }
    });
}

// The block's batch statistics from the lanes' partial sums: DPP row sums over the 16 pixel lanes, the WP waves of a cout group
// through LDS (red: [WP][2][BN] floats, free once every wave has left the K loop: the first barrier), one row of p.stats.
template <int TC, int WP, int BN>
__device__ __forceinline__ void epi_direct_stats(const ConvArgs &p, float (&s1)[TC / 2][8], float (&s2)[TC / 2][8], float *red, int wp, int wc,
                                                 int kg, int l15, int tid, int c0, size_t stats_row) {
#pragma unroll
    for (int j = 0; j < TC / 2; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            s1[j][t] = row16_sum(s1[j][t]);
            s2[j][t] = row16_sum(s2[j][t]);
        }
    __syncthreads();
    if (l15 == 0) {
#pragma unroll
        for (int j = 0; j < TC / 2; ++j) {
            float *d1 = red + (wp * 2 + 0) * BN + wc * TC * 16 + j * 32 + kg * 8, *d2 = d1 + BN;
            *reinterpret_cast<float4 *>(d1) = make_float4(s1[j][0], s1[j][1], s1[j][2], s1[j][3]);
            *reinterpret_cast<float4 *>(d1 + 4) = make_float4(s1[j][4], s1[j][5], s1[j][6], s1[j][7]);
            *reinterpret_cast<float4 *>(d2) = make_float4(s2[j][0], s2[j][1], s2[j][2], s2[j][3]);
            *reinterpret_cast<float4 *>(d2 + 4) = make_float4(s2[j][4], s2[j][5], s2[j][6], s2[j][7]);
        }
    }
    __syncthreads();
    if (tid < BN && c0 + tid < p.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WP; ++w) {
            t1 += red[(w * 2 + 0) * BN + tid];
            t2 += red[(w * 2 + 1) * BN + tid];
        }
        p.stats[(stats_row * 2 + 0) * p.Cout + c0 + tid] = t1;
        p.stats[(stats_row * 2 + 1) * p.Cout + c0 + tid] = t2;
    }
}

}  // namespace cer
