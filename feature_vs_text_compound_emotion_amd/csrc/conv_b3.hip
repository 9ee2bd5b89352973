// conv_b3.hip -- implicit-GEMM convolution with fp32-class accuracy on the BF16 matrix cores.
//
// Every operand is carried as a SPLIT value: hi = bf16(v), lo = bf16(v - hi) (two bf16 planes, the
// same 4 bytes per element as fp32).  A product a*b is evaluated as
//        a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (the dropped a_lo*b_lo term is ~2^-18 relative)
// with three v_mfma_f32_32x32x16_bf16 into one fp32 accumulator.  That is 96 MFMA cycles per 16
// k-values of a 32x32 tile against 512 cycles on the fp32 MFMA path: a 5.3x higher matrix ceiling
// (2.5 PFLOP/s / 3 = 833 TFLOP/s effective) while the measured end-to-end logit error of the whole
// IR-50 + LFAN stack stays at 1.3e-6 (plain bf16: 8e-4), DESIGN.md section 4.
//
// Structure mirrors conv_igemm.hip: D[i = cout][j = pixel] so the epilogue owns 4 consecutive couts
// per lane, register-staged single LDS buffer, scalar-base + 32-bit-offset addressing with a
// tap-validity bit mask for the zero padding.  LDS planes are [row][k] bf16 with a (BK+8)-element
// pitch (80 / 144 bytes): 16-byte fragment reads (8 consecutive k = one MFMA operand) are conflict
// free for the 16 rows of a ds_read_b128 lane group.
#include "conv_common.h"

namespace cer {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // 16-byte staging unit (first-class vector:
                                                                 // HIP's uint4 struct arrays ended up in scratch)

__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// ABL != 0: timing-only ablation (no staging inside the K loop; results are WRONG by construction)
template <int BM, int BN, int WP, int WC, int BKT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_b3_kernel(ConvArgs p) {
    static_assert(WP * WC == 4, "4 waves per block");
    constexpr int CH = BKT / 8;    // 16-byte chunks (8 bf16) per staged row and plane
    constexpr int RPP = 256 / CH;  // rows staged per pass
    constexpr int XR = BM / RPP, WR = BN / RPP;
    constexpr int PB = BKT + 8;    // LDS row pitch in bf16 elements
    constexpr int TP = BM / (32 * WP), TC = BN / (32 * WC);
    constexpr int NK = BKT / 16;   // MFMA k-substeps per staged step
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_b3[];
    uint16_t *Xh = smem_b3, *Xl = Xh + BM * PB, *Wh = Xl + BM * PB, *Wl = Wh + BN * PB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wc = wave / WP;
    const int half = lane >> 5, l31 = lane & 31;

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, c0 = tile_n * BN;
    const int split = blockIdx.z;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.steps, s_begin + p.steps_per_split);

    // ---- staging assignment ----
    const int chunk = tid % CH, srow = tid / CH;
    unsigned x_rel[XR], x_taps[XR], w_rel[WR];
    bool w_ok[WR];
    long long tile_base;  // bytes from the plane base to the tile's first row, tap (0,0)
    {
        const int hw = p.Ho * p.Wo;
        const int mm = m0 < p.M ? m0 : 0;
        const int n = mm / hw, r = mm - n * hw;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        tile_base = ((long long)(n * p.H + ho * p.stride - p.pad_t) * p.W + (wo * p.stride - p.pad_l)) * p.x_ld * 2;
    }
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + srow + RPP * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int n = mm / hw, r = mm - n * hw;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
        const long long rb = ((long long)(n * p.H + hi0) * p.W + wi0) * p.x_ld * 2;
        x_rel[i] = ok ? (unsigned)(rb - tile_base) + chunk * 16u : 0u;
        unsigned bits = 0;
        if (ok) {
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int kh = t / p.KW, kw = t - kh * p.KW;
                const int hi = hi0 + kh * p.dil_h, wi = wi0 + kw * p.dil_w;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) bits |= 1u << t;
            }
        }
        x_taps[i] = bits;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        w_ok[i] = c0 + srow + RPP * i < p.Cout;
        w_rel[i] = (unsigned)(((size_t)(srow + RPP * i) * p.Kpad + chunk * 8) * 2);
    }

    u32x4 xh[XR], xl[XR], wh[WR], wl[WR];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_step = [&](int s) {
        const int tap = s / p.cin_steps, cc = s - tap * p.cin_steps;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const long long soff = tile_base + ((long long)(kh * p.dil_h * p.W + kw * p.dil_w) * p.x_ld + cc * BKT) * 2;
        const char *bh = reinterpret_cast<const char *>(p.x_hi) + soff;
        const char *bl = reinterpret_cast<const char *>(p.x_lo) + soff;
        const unsigned tapbit = 1u << tap;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            if (x_taps[i] & tapbit) {
                xh[i] = *reinterpret_cast<const u32x4 *>(bh + x_rel[i]);
                xl[i] = *reinterpret_cast<const u32x4 *>(bl + x_rel[i]);
            } else {
                xh[i] = zero4;
                xl[i] = zero4;
            }
        }
        const size_t woff = ((size_t)c0 * p.Kpad + (size_t)s * BKT) * 2;
        const char *wbh = reinterpret_cast<const char *>(p.w_hi) + woff;
        const char *wbl = reinterpret_cast<const char *>(p.w_lo) + woff;
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            if (w_ok[i]) {
                wh[i] = *reinterpret_cast<const u32x4 *>(wbh + w_rel[i]);
                wl[i] = *reinterpret_cast<const u32x4 *>(wbl + w_rel[i]);
            } else {
                wh[i] = zero4;
                wl[i] = zero4;
            }
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int o = (srow + RPP * i) * PB + chunk * 8;
            *reinterpret_cast<u32x4 *>(Xh + o) = xh[i];
            *reinterpret_cast<u32x4 *>(Xl + o) = xl[i];
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int o = (srow + RPP * i) * PB + chunk * 8;
            *reinterpret_cast<u32x4 *>(Wh + o) = wh[i];
            *reinterpret_cast<u32x4 *>(Wl + o) = wl[i];
        }
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (s_begin < s_end) {
        load_step(s_begin);
        store_step();
    }
    __syncthreads();

    const int arow = (wc * TC * 32 + l31) * PB + half * 8;  // A = weights: row = cout
    const int brow = (wp * TP * 32 + l31) * PB + half * 8;  // B = activations: row = pixel
    for (int s = s_begin; s < s_end; ++s) {
        if constexpr (ABL == 0) {
            if (s + 1 < s_end) load_step(s + 1);
        }
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            bf16x8 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
            for (int a = 0; a < TC; ++a) {
                ah[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wh + arow + a * 32 * PB + kk * 16));
                al[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wl + arow + a * 32 * PB + kk * 16));
            }
#pragma unroll
            for (int b = 0; b < TP; ++b) {
                bh[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Xh + brow + b * 32 * PB + kk * 16));
                bl[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Xl + brow + b * 32 * PB + kk * 16));
            }
#pragma unroll
            for (int a = 0; a < TC; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                }
        }
        __syncthreads();  // everyone is done reading the LDS planes
        if constexpr (ABL == 0) {
            if (s + 1 < s_end) store_step();
        }
        __syncthreads();
    }

    // ---- epilogue (same D layout as the fp32 kernel) ----
    static_for<TP>([&](auto B) {
        constexpr int b = decltype(B)::v;
        const int m = m0 + (wp * TP + b) * 32 + l31;
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            static_for<4>([&](auto Q) {
                constexpr int q = decltype(Q)::v;
                const int c = c0 + (wc * TC + a) * 32 + 8 * q + 4 * half;
                float v[4] = {acc[a][b][4 * q + 0], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                if (m < p.M && c < p.Cout) {
                    if (p.split_k > 1) {
                        float *dst = p.y + ((size_t)split * p.M + m) * p.Cout + c;
                        if (((p.Cout & 3) == 0) && c + 3 < p.Cout) {
                            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c + e < p.Cout) dst[e] = v[e];
                        }
                    } else {
                        epilogue_store4(p, m, c, v);
                    }
                }
            });
        });
    });

    if (p.stats) {
        float *red = reinterpret_cast<float *>(smem_b3);  // [WP][2][BN]; the K loop ended on a barrier
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            static_for<16>([&](auto Rg) {
                constexpr int r = decltype(Rg)::v;
                float s1 = 0.f, s2 = 0.f;
                static_for<TP>([&](auto B) {
                    constexpr int b = decltype(B)::v;
                    const int m = m0 + (wp * TP + b) * 32 + l31;
                    const float v = (m < p.M) ? acc[a][b][r] : 0.f;
                    s1 += v;
                    s2 += v * v;
                });
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o);
                    s2 += __shfl_xor(s2, o);
                }
                if (l31 == 0) {
                    const int ci = (wc * TC + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    red[(wp * 2 + 0) * BN + ci] = s1;
                    red[(wp * 2 + 1) * BN + ci] = s2;
                }
            });
        });
        __syncthreads();
        if (tid < BN && c0 + tid < p.Cout) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WP; ++w) {
                s1 += red[(w * 2 + 0) * BN + tid];
                s2 += red[(w * 2 + 1) * BN + tid];
            }
            p.stats[((size_t)tile_m * 2 + 0) * p.Cout + c0 + tid] = s1;
            p.stats[((size_t)tile_m * 2 + 1) * p.Cout + c0 + tid] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Ping-pong variant: 512 threads = two groups of 4 waves that own alternate K steps of the SAME
// 128x128 output tile.  While one group runs its 24 MFMAs per wave on LDS buffer g, the other
// group writes its next step (loaded two phases earlier, so HBM/L2 latency is covered by a whole
// compute phase) into buffer 1-g and issues the loads after that.  Each SIMD then always hosts one
// matrix wave and one staging wave -- complementary work instead of two waves fighting for the
// matrix pipe and stalling on memory together -- with ONE block barrier per K step.  The two
// partial accumulators meet in LDS at the end; group 0 runs the epilogue.
template <int BKT>
__global__ __launch_bounds__(512, 2) void conv_b3_pp_kernel(ConvArgs p) {
    constexpr int BM = 128, BN = 128, WP = 2, WC = 2, TP = 2, TC = 2;
    constexpr int CH = BKT / 8, RPP = 256 / CH, XR = BM / RPP, WR = BN / RPP, PB = BKT + 8, NK = BKT / 16;
    constexpr int PLANE = 128 * PB;  // elements per plane (BM == BN)
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_b3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3, gtid = tid & 255;
    const int wp = wq % WP, wc = wq / WP;
    const int half = lane >> 5, l31 = lane & 31;
    uint16_t *mybuf = smem_b3 + grp * 4 * PLANE;  // this group's buffer: Xh, Xl, Wh, Wl
    uint16_t *Xh = mybuf, *Xl = Xh + PLANE, *Wh = Xl + PLANE, *Wl = Wh + PLANE;

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, c0 = tile_n * BN;
    const int split = blockIdx.z;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.steps, s_begin + p.steps_per_split);

    const int chunk = gtid % CH, srow = gtid / CH;
    unsigned x_rel[XR], x_taps[XR], w_rel[WR];
    bool w_ok[WR];
    long long tile_base;
    {
        const int hw = p.Ho * p.Wo;
        const int mm = m0 < p.M ? m0 : 0;
        const int n = mm / hw, r = mm - n * hw;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        tile_base = ((long long)(n * p.H + ho * p.stride - p.pad_t) * p.W + (wo * p.stride - p.pad_l)) * p.x_ld * 2;
    }
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + srow + RPP * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int n = mm / hw, r = mm - n * hw;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
        const long long rb = ((long long)(n * p.H + hi0) * p.W + wi0) * p.x_ld * 2;
        x_rel[i] = ok ? (unsigned)(rb - tile_base) + chunk * 16u : 0u;
        unsigned bits = 0;
        if (ok) {
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int kh = t / p.KW, kw = t - kh * p.KW;
                const int hi = hi0 + kh * p.dil_h, wi = wi0 + kw * p.dil_w;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) bits |= 1u << t;
            }
        }
        x_taps[i] = bits;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        w_ok[i] = c0 + srow + RPP * i < p.Cout;
        w_rel[i] = (unsigned)(((size_t)(srow + RPP * i) * p.Kpad + chunk * 8) * 2);
    }
    u32x4 xh[XR], xl[XR], wh[WR], wl[WR];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto load_step = [&](int s) {
        const int tap = s / p.cin_steps, cc = s - tap * p.cin_steps;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const long long soff = tile_base + ((long long)(kh * p.dil_h * p.W + kw * p.dil_w) * p.x_ld + cc * BKT) * 2;
        const char *bh = reinterpret_cast<const char *>(p.x_hi) + soff;
        const char *bl = reinterpret_cast<const char *>(p.x_lo) + soff;
        const unsigned tapbit = 1u << tap;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            if (x_taps[i] & tapbit) {
                xh[i] = *reinterpret_cast<const u32x4 *>(bh + x_rel[i]);
                xl[i] = *reinterpret_cast<const u32x4 *>(bl + x_rel[i]);
            } else {
                xh[i] = zero4;
                xl[i] = zero4;
            }
        }
        const size_t woff = ((size_t)c0 * p.Kpad + (size_t)s * BKT) * 2;
        const char *wbh = reinterpret_cast<const char *>(p.w_hi) + woff;
        const char *wbl = reinterpret_cast<const char *>(p.w_lo) + woff;
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            if (w_ok[i]) {
                wh[i] = *reinterpret_cast<const u32x4 *>(wbh + w_rel[i]);
                wl[i] = *reinterpret_cast<const u32x4 *>(wbl + w_rel[i]);
            } else {
                wh[i] = zero4;
                wl[i] = zero4;
            }
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int o = (srow + RPP * i) * PB + chunk * 8;
            *reinterpret_cast<u32x4 *>(Xh + o) = xh[i];
            *reinterpret_cast<u32x4 *>(Xl + o) = xl[i];
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int o = (srow + RPP * i) * PB + chunk * 8;
            *reinterpret_cast<u32x4 *>(Wh + o) = wh[i];
            *reinterpret_cast<u32x4 *>(Wl + o) = wl[i];
        }
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // prologue: group 0 stages its first step and prefetches the next; group 1 only prefetches
    const int first = s_begin + grp;
    if (first < s_end) load_step(first);
    if (grp == 0) {
        if (first < s_end) store_step();
        if (first + 2 < s_end) load_step(first + 2);
    }
    __syncthreads();

    const int arow = (wc * TC * 32 + l31) * PB + half * 8;
    const int brow = (wp * TP * 32 + l31) * PB + half * 8;
    for (int s = s_begin; s < s_end; ++s) {
        if (((s - s_begin) & 1) == grp) {
            // my step: matrix phase on my buffer
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                bf16x8 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
                for (int a = 0; a < TC; ++a) {
                    ah[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wh + arow + a * 32 * PB + kk * 16));
                    al[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wl + arow + a * 32 * PB + kk * 16));
                }
#pragma unroll
                for (int b = 0; b < TP; ++b) {
                    bh[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Xh + brow + b * 32 * PB + kk * 16));
                    bl[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Xl + brow + b * 32 * PB + kk * 16));
                }
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int b = 0; b < TP; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    }
            }
        } else {
            // the other group's step: staging phase for my next step (s + 1), prefetch of s + 3
            if (s + 1 < s_end) store_step();
            if (s + 3 < s_end) load_step(s + 3);
        }
        __syncthreads();
    }

    // ---- merge the two groups' partial accumulators through LDS (64 KB), group 0 finishes ----
    float *red = reinterpret_cast<float *>(smem_b3);  // [4 waves][TC*TP tiles][16 regs][64 lanes]
    if (grp == 1) {
#pragma unroll
        for (int a = 0; a < TC; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wq * 4 + a * TP + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((wq * 4 + a * TP + b) * 16 + r) * 64 + lane];

    static_for<TP>([&](auto B) {
        constexpr int b = decltype(B)::v;
        const int m = m0 + (wp * TP + b) * 32 + l31;
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            static_for<4>([&](auto Q) {
                constexpr int q = decltype(Q)::v;
                const int c = c0 + (wc * TC + a) * 32 + 8 * q + 4 * half;
                float v[4] = {acc[a][b][4 * q + 0], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                if (m < p.M && c < p.Cout) {
                    if (p.split_k > 1) {
                        float *dst = p.y + ((size_t)split * p.M + m) * p.Cout + c;
                        if (((p.Cout & 3) == 0) && c + 3 < p.Cout) {
                            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c + e < p.Cout) dst[e] = v[e];
                        }
                    } else {
                        epilogue_store4(p, m, c, v);
                    }
                }
            });
        });
    });
    if (p.stats) {
        // group 1 has left: synchronise the four remaining waves with an LDS-only protocol is not
        // possible with s_barrier, so the per-tile sums go through wave shuffles + global partials:
        // each (wp) half writes its own partial row, bn_finalize adds them (2 rows per tile).
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            static_for<16>([&](auto Rg) {
                constexpr int r = decltype(Rg)::v;
                float s1 = 0.f, s2 = 0.f;
                static_for<TP>([&](auto B) {
                    constexpr int b = decltype(B)::v;
                    const int m = m0 + (wp * TP + b) * 32 + l31;
                    const float v = (m < p.M) ? acc[a][b][r] : 0.f;
                    s1 += v;
                    s2 += v * v;
                });
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o);
                    s2 += __shfl_xor(s2, o);
                }
                const int ci = c0 + (wc * TC + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (l31 == 0 && ci < p.Cout) {
                    p.stats[((size_t)(tile_m * 2 + wp) * 2 + 0) * p.Cout + ci] = s1;
                    p.stats[((size_t)(tile_m * 2 + wp) * 2 + 1) * p.Cout + ci] = s2;
                }
            });
        });
    }
}

// v -> (bf16(v'), bf16(v' - bf16(v'))) with v' = v*scale[c] + shift[c] (channels-last, optional),
// 4 elements per thread
__global__ void split_bf16_kernel(const float4 *__restrict__ x, const float *__restrict__ scale,
                                  const float *__restrict__ shift, ushort4 *__restrict__ hi, ushort4 *__restrict__ lo,
                                  size_t n4, int C4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = x[i];
    if (scale) {
        const int c = (int)(i % C4) * 4;
        const float4 s = *reinterpret_cast<const float4 *>(scale + c), t = *reinterpret_cast<const float4 *>(shift + c);
        v = make_float4(v.x * s.x + t.x, v.y * s.y + t.y, v.z * s.z + t.z, v.w * s.w + t.w);
    }
    ushort4 h, l;
    split_bf16(v.x, h.x, l.x); split_bf16(v.y, h.y, l.y);
    split_bf16(v.z, h.z, l.z); split_bf16(v.w, h.w, l.w);
    hi[i] = h;
    lo[i] = l;
}

template <int BM, int BN, int WP, int WC, int BKT, int ABL = 0>
static int launch_b3(const ConvArgs &a, hipStream_t st) {
    const size_t lds = (size_t)(2 * BM + 2 * BN) * (BKT + 8) * sizeof(uint16_t);
    auto k = conv_b3_kernel<BM, BN, WP, WC, BKT, ABL>;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CER_LAUNCH(k, dim3(a.tiles_m * a.tiles_n, 1, a.split_k), dim3(256), lds, st, a);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

// tile ids (desc.tile): 0 auto; 1 = 128x128 BK32, 2 = 128x64 BK32, 3 = 128x128 BK64, 4 = 64x128 BK32, 5 = 64x64 BK32,
// 6 = ping-pong 128x128 (measured 5-10 % slower than 1: kept as a tested A/B variant), 9 = timing-only ablation.
// Measured and dropped: 256x128 / 128x256 tiles with 128x64 per wave (2 waves/SIMD, -3..-8 %), 256x64 for the
// Cout = 64 layers (-13 % vs 128x64).  The no-staging
// ablation reaches ~500 TFLOP/s effective and LDS store bandwidth (32 KB per K step at ~80 B/clk against 768
// MFMA cycles) is what the register-staged structure runs into; LDS-DMA staging is the next step.
int conv_b3_tile_dims(int tile, int Cout, long long M, int &bm, int &bn, int &bk) {
    if (tile == 0) tile = Cout <= 64 ? 2 : ((M + 127) / 128 * ((Cout + 127) / 128) >= 512 ? 1 : (Cout >= 128 ? 4 : 5));
    switch (tile) {
        case 1: bm = 128; bn = 128; bk = 32; break;
        case 2: bm = 128; bn = 64; bk = 32; break;
        case 3: bm = 128; bn = 128; bk = 64; break;
        case 4: bm = 64; bn = 128; bk = 32; break;
        case 5: bm = 64; bn = 64; bk = 32; break;
        case 6: bm = 128; bn = 128; bk = 32; break;  // ping-pong, 512 threads
        case 9: bm = 128; bn = 128; bk = 32; break;  // timing-only ablation of tile 1
        default: return 0;
    }
    return tile;
}

int conv_b3_launch(int tile, const ConvArgs &a, hipStream_t st) {
    switch (tile) {
        case 1: return launch_b3<128, 128, 2, 2, 32>(a, st);
        case 2: return launch_b3<128, 64, 2, 2, 32>(a, st);
        case 3: return launch_b3<128, 128, 2, 2, 64>(a, st);
        case 4: return launch_b3<64, 128, 1, 4, 32>(a, st);
        case 5: return launch_b3<64, 64, 2, 2, 32>(a, st);
        case 6: {
            const size_t lds = (size_t)2 * 4 * 128 * (32 + 8) * sizeof(uint16_t);
            auto k = conv_b3_pp_kernel<32>;
            CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CER_LAUNCH(k, dim3(a.tiles_m * a.tiles_n, 1, a.split_k), dim3(512), lds, st, a);
            CER_HIP_CHECK(hipGetLastError());
            return CER_OK;
        }
        case 9: return launch_b3<128, 128, 2, 2, 32, 1>(a, st);
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (bf16x3): unknown tile id");
    }
}

}  // namespace cer

using namespace cer;

extern "C" int cer_split_bf16(const float *x, const float *scale, const float *shift, int C, uint16_t *hi, uint16_t *lo,
                              size_t n, void *stream) {
    if (!x || !hi || !lo || n == 0 || (n & 3)) return cer_set_error(CER_ERR_INVALID_ARG, "split_bf16: n must be a positive multiple of 4");
    if ((scale == nullptr) != (shift == nullptr) || (scale && (C <= 0 || (C & 3) || n % C)))
        return cer_set_error(CER_ERR_INVALID_ARG, "split_bf16: the per-channel affine needs scale, shift and C % 4 == 0 dividing n");
    CER_LAUNCH(split_bf16_kernel, dim3(cer_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, scale,
               shift, (ushort4 *)hi, (ushort4 *)lo, n / 4, scale ? C / 4 : 1);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
