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

// ------------------------------------------------------------------------------------------
// LDS-DMA variant (`buffer_load_dwordx4 ... offen lds`): the four operand planes of a K step go
// from L2/HBM straight into LDS -- no staging VGPRs, no ds_write_b128 (whose VGPR->LDS path moves
// only ~79 B/clk/CU and was the limiter of the register-staged kernel above) -- into one of TWO LDS
// buffers, so the transfer of step s+1 flies during the MFMAs of step s and a step costs ONE barrier.
//
// One wave-instruction writes 1 KiB of LDS linearly (lane l -> base + 16*l), i.e. 16 rows of 64 bytes
// (BK = 32 bf16) of a plane.  Rows are therefore unpadded and the bank conflicts of the fragment
// reads are removed by an XOR swizzle applied on the SOURCE side: LDS 16-byte slot q of row r holds
// k-chunk q ^ ((r >> 2) & 3), so the 16 lanes of a ds_read_b128 group (rows {0-3,12-15,20-27} /
// {4-11,16-19,28-31}, same k-chunk) hit 16 different slots of the 256-byte bank row.
// Zero padding: a tap outside the image (or a row past M / Cout) gets the byte offset 0x80000000,
// beyond the descriptor's 2 GiB record count -- a range-checked buffer load returns zeros, and
// zeros are what the DMA then writes.
typedef __attribute__((address_space(3))) void *lds_ptr_t;

// ABL (timing-only builds, results are wrong by construction): bit 0 / bit 1 give the activation / weight
// descriptors ZERO records, so every DMA of that operand is dropped by the range check while the
// instruction stream, the waits and the barriers stay -- the price of that operand's memory traffic.
template <int BM, int BN, int WP, int WC, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_b3_dma_kernel(ConvArgs p) {
    static_assert(WP * WC == 4 && BM % 64 == 0 && BN % 64 == 0, "4 waves; 1-KiB pieces are dealt round-robin to them");
    constexpr int BKT = 32, ROWB = BKT * 2;        // bytes per row and plane
    constexpr int XP = BM / 64, WQ = BN / 64;      // 1-KiB pieces per wave and plane
    constexpr int PX = BM * ROWB, PW = BN * ROWB;  // plane sizes in bytes
    constexpr int BUF = 2 * PX + 2 * PW;
    constexpr int TP = BM / (32 * WP), TC = BN / (32 * WC);
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_b3[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_b3);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // ---- DMA assignment: wave w moves pieces w, w+4, ... of every plane; in a piece lane l owns
    // row l/4, LDS slot l%4 ----
    const int prow = lane >> 2, slot = lane & 3;
    unsigned x_off[XP], x_taps[XP], w_off[WQ];
    long long tile_base;  // bytes from the plane base to the tile's first row, tap (0,0)
    {
        const int hw = p.Ho * p.Wo;
        const int mm = m0 < p.M ? m0 : 0;
        const int n = mm / hw, r = mm - n * hw;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        tile_base = ((long long)(n * p.H + ho * p.stride - p.pad_t) * p.W + (wo * p.stride - p.pad_l)) * p.x_ld * 2;
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int row = (wave + 4 * i) * 16 + prow;
        const int chunk = slot ^ ((row >> 2) & 3);
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int n = mm / hw, r = mm - n * hw;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
        const long long rb = ((long long)(n * p.H + hi0) * p.W + wi0) * p.x_ld * 2;
        x_off[i] = (unsigned)(rb - tile_base) + chunk * 16u;
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
    for (int i = 0; i < WQ; ++i) {
        const int row = (wave + 4 * i) * 16 + prow;
        const int chunk = slot ^ ((row >> 2) & 3);
        w_off[i] = c0 + row < p.Cout ? (unsigned)(((size_t)row * p.Kpad + chunk * 8) * 2) : OOB;
    }

    auto issue = [&](int s, int buf) {
        const int tap = s / p.cin_steps, cc = s - tap * p.cin_steps;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const long long soff = tile_base + ((long long)(kh * p.dil_h * p.W + kw * p.dil_w) * p.x_ld + cc * BKT) * 2;
        const size_t woff = ((size_t)c0 * p.Kpad + (size_t)s * BKT) * 2;
        char *xhb = const_cast<char *>(reinterpret_cast<const char *>(p.x_hi)) + soff;
        char *xlb = const_cast<char *>(reinterpret_cast<const char *>(p.x_lo)) + soff;
        char *whb = const_cast<char *>(reinterpret_cast<const char *>(p.w_hi)) + woff;
        char *wlb = const_cast<char *>(reinterpret_cast<const char *>(p.w_lo)) + woff;
        constexpr int XREC = (ABL & 1) ? 0 : (int)OOB, WREC = (ABL & 2) ? 0 : (int)OOB;
        const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc(xhb, 0, XREC, 0x00020000);
        const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc(xlb, 0, XREC, 0x00020000);
        const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(whb, 0, WREC, 0x00020000);
        const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(wlb, 0, WREC, 0x00020000);
        const unsigned tapbit = 1u << tap;
        unsigned char *dst = smem + buf * BUF + wave * 1024;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int vo = (int)((x_taps[i] & tapbit) ? x_off[i] : OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_ptr_t)(dst + i * 4096), 16, vo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_ptr_t)(dst + PX + i * 4096), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, (lds_ptr_t)(dst + 2 * PX + i * 4096), 16, (int)w_off[i], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, (lds_ptr_t)(dst + 2 * PX + PW + i * 4096), 16, (int)w_off[i], 0, 0, 0);
        }
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (s_begin < s_end) issue(s_begin, 0);

    // fragment addresses: row * 64 bytes + swizzled slot of k-chunk (2*kk + half)
    const int sw = (l31 >> 2) & 3;
    const int arow = (wc * TC * 32 + l31) * ROWB + ((half ^ sw) << 4);  // A = weights: row = cout
    const int brow = (wp * TP * 32 + l31) * ROWB + ((half ^ sw) << 4);  // B = activations: row = pixel
    for (int s = s_begin; s < s_end; ++s) {
        const int cur = (s - s_begin) & 1;
        // every wave has seen its own pieces of step s land, and (barrier) everyone else's; the barrier also
        // closes the reads of step s-1, whose buffer the next DMA overwrites
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 1 < s_end) issue(s + 1, cur ^ 1);
        const unsigned char *Xh = smem + cur * BUF, *Xl = Xh + PX, *Wh = Xh + 2 * PX, *Wl = Wh + PW;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
            for (int a = 0; a < TC; ++a) {
                ah[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wh + ((arow + a * 32 * ROWB) ^ (kk << 5))));
                al[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wl + ((arow + a * 32 * ROWB) ^ (kk << 5))));
            }
#pragma unroll
            for (int b = 0; b < TP; ++b) {
                bh[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Xh + ((brow + b * 32 * ROWB) ^ (kk << 5))));
                bl[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Xl + ((brow + b * 32 * ROWB) ^ (kk << 5))));
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
    }

    // ---- epilogue (same D layout as the kernels above) ----
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
        __syncthreads();  // the last step's fragment reads are done: the planes can be reused
        float *red = reinterpret_cast<float *>(smem_b3);  // [WP][2][BN]
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

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swz16(int q) { return (0x78 >> (2 * q)) & 3; }  // F = {0, 2, 3, 1}

// The same kernel on v_mfma_f32_16x16x32_bf16: one MFMA eats the whole 32-deep K step of a 16x16 tile.  Same
// FLOPs per cycle as 32x32x16, but the chip holds a higher clock on this shape under the power limit that
// governs bf16 MFMA loops on real data (MI355X_MICROARCH.md, DVFS note 7) -- measured here, not assumed: see
// DESIGN.md section 4.  Fragment = row (lane & 15), 16-byte k-chunk (lane >> 4); a ds_read_b128 lane group then
// covers 16 different rows, 8 with chunk c and 8 with chunk c^1, and the XOR swizzle that keeps that conflict
// free is slot = chunk ^ F[(row >> 2) & 3] with F = {0, 2, 3, 1} (applied on the DMA's source side as before).
// NW = 4 waves (two blocks per CU) or 8 waves: the 256x256 tile of 8 waves x (128 pixels x 64 couts) halves the
// L2 -> LDS bytes per FLOP of the 128x128 tile and the LDS fragment reads per MFMA drop by a quarter; one block per CU.
// STAGES = 2: the DMA of step s+1 flies during step s (vmcnt(0) at every step).  STAGES = 3: two steps ahead with a
// counted vmcnt -- the 128x64 tile's step is only 384 MFMA cycles per wave, less than an L2 round trip under load (PMC:
// 42 % of its wave cycles parked at the wait/barrier), so it wants the deeper ring even at 2 instead of 3 blocks per CU.
template <int BM, int BN, int WP, int WC, int NW = 4, int STAGES = 2>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv_b3_dma16_kernel(ConvArgs p) {
    static_assert(WP * WC == NW && BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "1-KiB pieces are dealt round-robin to the waves");
    constexpr int BKT = 32, ROWB = BKT * 2;        // bytes per row and plane
    constexpr int XP = BM / (16 * NW), WQ = BN / (16 * NW);  // 1-KiB pieces per wave and plane
    constexpr int PX = BM * ROWB, PW = BN * ROWB;  // plane sizes in bytes
    constexpr int BUF = 2 * PX + 2 * PW;
    constexpr int NDMA = 2 * XP + 2 * WQ;  // DMA instructions a wave issues per step
    static_assert(STAGES == 2 || STAGES == 3, "ring depth");
    constexpr int TP = BM / (16 * WP), TC = BN / (16 * WC);  // 16x16 tiles per wave
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_b3[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_b3);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % WP, wc = wave / WP;
    const int kg = lane >> 4, l15 = lane & 15;

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

    // ---- DMA assignment: wave w moves pieces w, w+4, ... of every plane; in a piece lane l owns
    // row l/4, LDS slot l%4 ----
    const int prow = lane >> 2, slot = lane & 3;
    unsigned x_off[XP], x_taps[XP], w_off[WQ];
    long long tile_base;  // bytes from the plane base to the tile's first row, tap (0,0)
    {
        const int hw = p.Ho * p.Wo;
        const int mm = m0 < p.M ? m0 : 0;
        const int n = mm / hw, r = mm - n * hw;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        tile_base = ((long long)(n * p.H + ho * p.stride - p.pad_t) * p.W + (wo * p.stride - p.pad_l)) * p.x_ld * 2;
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int row = (wave + NW * i) * 16 + prow;
        const int chunk = slot ^ swz16((row >> 2) & 3);
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int n = mm / hw, r = mm - n * hw;
        const int ho = r / p.Wo, wo = r - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
        const long long rb = ((long long)(n * p.H + hi0) * p.W + wi0) * p.x_ld * 2;
        x_off[i] = (unsigned)(rb - tile_base) + chunk * 16u;
        unsigned bits = 0;
        if (ok) {  // nested loops: no per-tap integer division in the prologue (it is ~30 % of a K = 576 block otherwise)
            int t = 0;
            for (int kh = 0; kh < p.KH; ++kh) {
                const bool hok = (unsigned)(hi0 + kh * p.dil_h) < (unsigned)p.H;
                for (int kw = 0; kw < p.KW; ++kw, ++t)
                    if (hok && (unsigned)(wi0 + kw * p.dil_w) < (unsigned)p.W) bits |= 1u << t;
            }
        }
        x_taps[i] = bits;
    }
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        const int row = (wave + NW * i) * 16 + prow;
        const int chunk = slot ^ swz16((row >> 2) & 3);
        w_off[i] = c0 + row < p.Cout ? (unsigned)(((size_t)row * p.Kpad + chunk * 8) * 2) : OOB;
    }

    auto issue = [&](int s, int buf) {
        // K order: channel-chunk major (all KH*KW taps of channels [32cc, 32cc+32), then the next chunk): the nine taps
        // re-read the same input rows in consecutive steps, while L2 still holds them (tap-major order, as the weights are
        // packed, spreads those re-reads over the whole K loop: 2.9x the algorithmic bytes reached the fabric)
        const int T = p.KH * p.KW;
        const int cc = s / T, tap = s - cc * T;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const long long soff = tile_base + ((long long)(kh * p.dil_h * p.W + kw * p.dil_w) * p.x_ld + cc * BKT) * 2;
        const size_t woff = ((size_t)c0 * p.Kpad + (size_t)tap * p.Cin + (size_t)cc * BKT) * 2;
        char *xhb = const_cast<char *>(reinterpret_cast<const char *>(p.x_hi)) + soff;
        char *xlb = const_cast<char *>(reinterpret_cast<const char *>(p.x_lo)) + soff;
        char *whb = const_cast<char *>(reinterpret_cast<const char *>(p.w_hi)) + woff;
        char *wlb = const_cast<char *>(reinterpret_cast<const char *>(p.w_lo)) + woff;
        constexpr int XREC = (int)OOB, WREC = (int)OOB;
        const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc(xhb, 0, XREC, 0x00020000);
        const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc(xlb, 0, XREC, 0x00020000);
        const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(whb, 0, WREC, 0x00020000);
        const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(wlb, 0, WREC, 0x00020000);
        const unsigned tapbit = 1u << tap;
        unsigned char *dst = smem + buf * BUF + wave * 1024;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int vo = (int)((x_taps[i] & tapbit) ? x_off[i] : OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_ptr_t)(dst + i * (NW * 1024)), 16, vo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_ptr_t)(dst + PX + i * (NW * 1024)), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, (lds_ptr_t)(dst + 2 * PX + i * (NW * 1024)), 16, (int)w_off[i], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, (lds_ptr_t)(dst + 2 * PX + PW + i * (NW * 1024)), 16, (int)w_off[i], 0, 0, 0);
        }
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (s_begin + st < s_end) issue(s_begin + st, st);

    // fragment addresses: row * 64 bytes + swizzled slot of k-chunk kg
    const int sw = swz16((l15 >> 2) & 3);
    const int arow = (wc * TC * 16 + l15) * ROWB + ((kg ^ sw) << 4);  // A = weights: row = cout
    const int brow = (wp * TP * 16 + l15) * ROWB + ((kg ^ sw) << 4);  // B = activations: row = pixel
    int cur = 0;
    for (int s = s_begin; s < s_end; ++s, cur = (cur + 1 == STAGES ? 0 : cur + 1)) {
        // every wave has seen its own pieces of step s land (the STAGES-2 younger steps may still fly), and (barrier)
        // everyone else's; the barrier also closes the reads of step s-1, whose buffer the next DMA overwrites
        if (STAGES == 3 && s + 1 < s_end) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (s + STAGES - 1 < s_end) issue(s + STAGES - 1, cur == 0 ? STAGES - 1 : cur - 1);
        const unsigned char *Xh = smem + cur * BUF, *Xl = Xh + PX, *Wh = Xh + 2 * PX, *Wl = Wh + PW;
        bf16x8 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
        for (int a = 0; a < TC; ++a) {
            ah[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wh + arow + a * 16 * ROWB));
            al[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wl + arow + a * 16 * ROWB));
        }
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            bh[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Xh + brow + b * 16 * ROWB));
            bl[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Xl + brow + b * 16 * ROWB));
        }
#pragma unroll
        for (int a = 0; a < TC; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            }
    }

    // ---- epilogue ----
    // Accumulator layout: D[i = cout][j = pixel] of a 16x16 tile, lane = couts 4*(lane>>4) + 0..3 of pixel lane&15.  Storing
    // straight from it means 16 scattered 16-byte pieces per instruction and one fully inlined copy of the (large) fused
    // epilogue per 16x16 tile -- 20-40k instructions, several times the instruction cache, and as long as the whole K loop
    // of a K = 576 layer.  So the tile goes through LDS once (the staging buffers are free now): rows of BN floats, 16-byte
    // granules XOR-swizzled by (row & 15) so that the 16 pixels of a ds_write_b128 group hit 16 different granules; then
    // ONE compact loop in which consecutive lanes own consecutive couts of a pixel (1 KiB contiguous per wave store).
    constexpr bool LDSEPI = BM * BN * 4 <= STAGES * BUF;
    if constexpr (LDSEPI) {
        constexpr int G = BN / 4;               // 16-byte granules per row (>= 16)
        constexpr int RPI = NW * 64 / G;        // rows per loop iteration
        float *Ct = reinterpret_cast<float *>(smem_b3);
        __syncthreads();  // every wave is done with the fragment reads of the last step
        static_for<TP>([&](auto B) {
            constexpr int b = decltype(B)::v;
            const int ml = (wp * TP + b) * 16 + l15;
            static_for<TC>([&](auto A) {
                constexpr int a = decltype(A)::v;
                const int g = (wc * TC + a) * 4 + kg;
                *reinterpret_cast<f32x4 *>(Ct + ml * BN + ((g ^ (ml & 15)) << 2)) = acc[a][b];
            });
        });
        __syncthreads();
        const int g = tid % G, r0 = tid / G;
        const int c = c0 + g * 4;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        // row / column of the thread's first pixel once (bias9), then stepped without divisions
        int ho = 0, wo = 0;
        if (p.bias9) {
            const int mm = m0 + r0 < p.M ? m0 + r0 : 0;
            const int r = mm % (p.Ho * p.Wo);
            ho = r / p.Wo;
            wo = r - ho * p.Wo;
        }
        for (int ml = r0; ml < BM; ml += RPI) {
            const int m = m0 + ml;
            if (m >= p.M) break;
            const f32x4 q = *reinterpret_cast<const f32x4 *>(Ct + ml * BN + ((g ^ (ml & 15)) << 2));
            float v[4] = {q[0], q[1], q[2], q[3]};
            if (c < p.Cout) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] += v[e];
                    s2[e] += v[e] * v[e];
                }
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
                    const float *brow = p.bias;
                    if (p.bias9) {
                        const int ry = ho == 0 ? 0 : (ho == p.Ho - 1 ? 2 : 1), rx = wo == 0 ? 0 : (wo == p.Wo - 1 ? 2 : 1);
                        brow = p.bias9 + (size_t)(3 * ry + rx) * p.Cout;
                    }
                    epilogue_store4(p, m, c, v, brow);
                }
            }
            if (p.bias9) {  // advance RPI pixels
                wo += RPI;
                while (wo >= p.Wo) {
                    wo -= p.Wo;
                    if (++ho == p.Ho) ho = 0;
                }
            }
        }
        if (p.stats) {
            __syncthreads();  // Ct has been consumed
            float *red = reinterpret_cast<float *>(smem_b3);  // [RPI][2][BN]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[(r0 * 2 + 0) * BN + g * 4 + e] = s1[e];
                red[(r0 * 2 + 1) * BN + g * 4 + e] = s2[e];
            }
            __syncthreads();
            if (tid < BN && c0 + tid < p.Cout) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < RPI; ++w) {
                    t1 += red[(w * 2 + 0) * BN + tid];
                    t2 += red[(w * 2 + 1) * BN + tid];
                }
                p.stats[((size_t)tile_m * 2 + 0) * p.Cout + c0 + tid] = t1;
                p.stats[((size_t)tile_m * 2 + 1) * p.Cout + c0 + tid] = t2;
            }
        }
        return;
    }
    // ---- direct epilogue (tiles whose accumulators do not fit the staging LDS: the 8-wave 256x256 variant) ----
    static_for<TP>([&](auto B) {
        constexpr int b = decltype(B)::v;
        const int m = m0 + (wp * TP + b) * 16 + l15;
        // the pixel's bias row is chosen ONCE per pixel tile (two integer divisions), not per 4-cout store
        const float *brow = (p.bias9 && m < p.M) ? p.bias9 + (size_t)bias9_case(p, m) * p.Cout : p.bias;
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            const int c = c0 + (wc * TC + a) * 16 + 4 * kg;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
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
                    epilogue_store4(p, m, c, v, brow);
                }
            }
        });
    });

    if (p.stats) {
        __syncthreads();  // the last step's fragment reads are done: the planes can be reused
        float *red = reinterpret_cast<float *>(smem_b3);  // [WP][2][BN]
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            static_for<4>([&](auto Rg) {
                constexpr int r = decltype(Rg)::v;
                float s1 = 0.f, s2 = 0.f;
                static_for<TP>([&](auto B) {
                    constexpr int b = decltype(B)::v;
                    const int m = m0 + (wp * TP + b) * 16 + l15;
                    const float v = (m < p.M) ? acc[a][b][r] : 0.f;
                    s1 += v;
                    s2 += v * v;
                });
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o);
                    s2 += __shfl_xor(s2, o);
                }
                if (l15 == 0) {
                    const int ci = (wc * TC + a) * 16 + 4 * kg + r;
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
// Window variant for "same" convolutions (stride 1, Ho == H, Wo == W): what limits the kernels above
// is the L2 -> LDS traffic of BOTH operands (zero-record ablations: activations dropped +56 %, weights
// dropped +48 %, both 1.85x, the DMA instructions themselves 5 %).  Here
//   * the tile's input pixels are fetched ONCE per 32-channel chunk and serve all KH*KW taps: for a
//     stride-1 conv the input of output pixel m under tap (kh, kw) is the pixel m + (kh-pad)*W + (kw-pad)
//     of the same flat [N*H*W] pixel array, so the LDS image is simply the contiguous window
//     [m0 - pad_t*W - pad_l, m0 + BM + ...) of that array (BM + (KH-1)*dil*W + (KW-1)*dil rows of 64
//     bytes, two planes), double buffered over chunks.  A tap is a scalar row shift of the fragment
//     address; taps that fall off the image (zero padding, the previous / next frame of the batch) read a
//     zeroed 64-byte slot instead (same address in every such lane: an LDS broadcast).
//   * 8 waves share a 256-pixel tile, so a weight slice is fetched once per 256 pixels (ring of 3 slices,
//     two K steps of latency tolerance, counted vmcnt).
// L2 -> LDS bytes per K step drop from 32 KB per 128x128x32 MACs to ~21 KB per 256x128x32 (3.1x less per
// FLOP).  Activation rows keep the (row >> 2) & 3 XOR swizzle of the DMA kernel: the 16 lanes of a
// ds_read_b128 group read 16 consecutive window rows -> conflict free for every tap shift.
// K order is chunk-major (all taps of channels [32cc, 32cc+32), then the next chunk).
template <int BM, int BN, int WP, int WC>
__global__ __launch_bounds__(512) void conv_b3_win_kernel(ConvArgs p, int NP) {
    static_assert(WP * WC == 8, "8 waves per block");
    constexpr int ROWB = 64;                         // bytes per row (32 bf16) and plane
    constexpr int TP = BM / (32 * WP), TC = BN / (32 * WC);
    constexpr int PW = BN * ROWB, WSLOT = 2 * PW;    // one ring slot: hi plane, lo plane
    constexpr int WQ = (2 * BN / 16) / 8;            // weight DMA instructions per wave and step (2 or 1)
    constexpr int NISSUE = 2 + WQ;                   // DMA instructions a wave issues per steady-state step
    constexpr unsigned OOB = 0x80000000u;
    static_assert(WQ == 1 || WQ == 2, "BN is 64 or 128");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_b3[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_b3);
    const int XPL = NP * 1024, XBUF = 2 * XPL;       // activation plane / buffer bytes
    unsigned char *Wr = smem + 2 * XBUF;             // weight ring [3][hi | lo]
    unsigned char *dummy = Wr + 3 * WSLOT;           // 1 KiB sink for the pieces a wave has no use for
    unsigned char *zslot = dummy + 1024;             // 64 zero bytes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % WP, wc = wave / WP;
    const int half = lane >> 5, l31 = lane & 31;
    if (tid < 16) reinterpret_cast<unsigned *>(zslot)[tid] = 0u;

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, c0 = tile_n * BN;
    const int T = p.KH * p.KW, total = p.cin_steps * T;
    const int wstart = m0 - (p.pad_t * p.W + p.pad_l);  // first window pixel (may be negative)

    // ---- per-lane DMA constants ----
    const int prow = lane >> 2, slot = lane & 3;
    const int dchunk = slot ^ ((prow >> 2) & 3);     // piece bases are multiples of 16 rows
    const int xrow0 = wave * 16 + prow;              // window row of this lane in piece `wave` (tap 0)
    const unsigned x_voff0 = (unsigned)xrow0 * p.x_ld * 2 + dchunk * 16;
    const unsigned x_vstep = 128u * p.x_ld * 2;      // 8 pieces further
    unsigned w_off;
    {
        const int piece = WQ == 2 ? wave : (wave & 3);
        const int row = piece * 16 + prow;
        w_off = c0 + row < p.Cout ? (unsigned)(((size_t)row * p.Kpad + dchunk * 8) * 2) : OOB;
    }
    const long long xbase0 = (long long)wstart * p.x_ld * 2;

    auto issue_x = [&](int cc, int t) {  // piece t*8 + wave of chunk cc (2 instructions, always)
        const int px = t * 8 + wave;
        const bool real = cc < p.cin_steps && px < NP && t < 8;
        const long long g = (long long)wstart + xrow0 + t * 128;
        const int vo = (real && g >= 0 && g < p.M) ? (int)(x_voff0 + t * x_vstep) : (int)OOB;
        char *bh = const_cast<char *>(reinterpret_cast<const char *>(p.x_hi)) + xbase0 + cc * 64;
        char *bl = const_cast<char *>(reinterpret_cast<const char *>(p.x_lo)) + xbase0 + cc * 64;
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(bh, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(bl, 0, (int)OOB, 0x00020000);
        unsigned char *dh = real ? smem + (cc & 1) * XBUF + px * 1024 : dummy;
        unsigned char *dl = real ? dh + XPL : dummy;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)dh, 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_ptr_t)dl, 16, vo, 0, 0, 0);
    };
    auto issue_w = [&](int cc, int tap, int ring) {  // WQ instructions
        const size_t woff = ((size_t)c0 * p.Kpad + (size_t)tap * p.Cin + (size_t)cc * 32) * 2;
        char *bh = const_cast<char *>(reinterpret_cast<const char *>(p.w_hi)) + woff;
        char *bl = const_cast<char *>(reinterpret_cast<const char *>(p.w_lo)) + woff;
        unsigned char *dst = Wr + ring * WSLOT;
        if constexpr (WQ == 2) {
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(bh, 0, (int)OOB, 0x00020000);
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(bl, 0, (int)OOB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + wave * 1024), 16, (int)w_off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_ptr_t)(dst + PW + wave * 1024), 16, (int)w_off, 0, 0, 0);
        } else {
            const bool lo = wave >= 4;
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(lo ? bl : bh, 0, (int)OOB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + (lo ? PW : 0) + (wave & 3) * 1024), 16, (int)w_off, 0, 0, 0);
        }
    };

    // ---- per-lane fragment constants ----
    int rbase[TP];
    unsigned x_taps[TP];
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        rbase[b] = (wp * TP + b) * 32 + l31;
        const int m = m0 + rbase[b];
        unsigned bits = 0;
        if (m < p.M) {
            const int hw = p.H * p.W;
            const int r = m % hw;
            const int ho = r / p.W, wo = r - ho * p.W;
            for (int t = 0; t < T; ++t) {
                const int kh = t / p.KW, kw = t - kh * p.KW;
                const int hi = ho - p.pad_t + kh * p.dil_h, wi = wo - p.pad_l + kw * p.dil_w;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) bits |= 1u << t;
            }
        }
        x_taps[b] = bits;
    }
    const int sw = (l31 >> 2) & 3;
    const int arow = (wc * TC * 32 + l31) * ROWB + ((half ^ sw) << 4);  // weights: row = cout

    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- software pipeline ----
    // Step s = (chunk cc, tap).  At the top of step s every wave waits for its own DMAs (weight slice s+1 and the
    // window pieces issued during step s-1) and meets the others at the barrier; it then issues slice s+2 and one
    // window piece of chunk cc+1, runs the 24 MFMAs of step s on fragments that are ALREADY in registers, and -- between
    // them -- reads the fragments of step s+1 (slice s+1 and the resident window are visible since the barrier).
    // So nothing but the barrier itself separates the MFMA streams of consecutive steps.
    struct Frags {
        bf16x8 ah[TC], al[TC], bh[TP], bl[TP];
    };
    auto load_frags = [&](Frags &f, int cc, int tap, int ring, int kk) {
        const unsigned char *Wh = Wr + ring * WSLOT, *Wl = Wh + PW;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const int toff = kh * p.dil_h * p.W + kw * p.dil_w;
        const int xbo = (cc & 1) * XBUF, zo = 4 * XPL + 3 * WSLOT + 1024;  // LDS byte offsets (ints: a select
                                                                             // between pointers decays to flat loads)
#pragma unroll
        for (int a = 0; a < TC; ++a) {
            f.ah[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wh + ((arow + a * 32 * ROWB) ^ (kk << 5))));
            f.al[a] = as_bf16x8(*reinterpret_cast<const u32x4 *>(Wl + ((arow + a * 32 * ROWB) ^ (kk << 5))));
        }
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            const int r = rbase[b] + toff;
            const int a0 = (r << 6) | ((((r >> 2) ^ half) & 3) << 4);
            const bool ok = (x_taps[b] >> tap) & 1u;
            f.bh[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(smem + ((ok ? xbo + a0 : zo) ^ (kk << 5))));
            f.bl[b] = as_bf16x8(*reinterpret_cast<const u32x4 *>(smem + ((ok ? xbo + XPL + a0 : zo) ^ (kk << 5))));
        }
    };
    auto mma = [&](const Frags &f) {
#pragma unroll
        for (int a = 0; a < TC; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[a], f.bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[a], f.bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[a], f.bh[b], acc[a][b], 0, 0, 0);
            }
    };

    // prologue: the whole window of chunk 0, weight slices 0 and 1
    for (int t = 0; t < 8; ++t) issue_x(0, t);
    issue_w(0, 0, 0);
    if (total > 1) issue_w(T > 1 ? 0 : 1, T > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;  // k-substeps 0 and 1 of the CURRENT step
    load_frags(f0, 0, 0, 0, 0);
    load_frags(f1, 0, 0, 0, 1);

    int ring = 0, cc = 0, tap = 0;
    for (int s = 0; s < total; ++s) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_x(cc + 1, tap);
        int t1 = tap + 1, c1 = cc;
        if (t1 >= T) { t1 = 0; c1 += 1; }
        int t2 = t1 + 1, c2 = c1;
        if (t2 >= T) { t2 = 0; c2 += 1; }
        const int r1 = ring == 2 ? 0 : ring + 1, r2 = r1 == 2 ? 0 : r1 + 1;
        if (s + 2 < total) issue_w(c2, t2, r2);
        const bool more = s + 1 < total;
        Frags n0;
        mma(f0);
        if (more) load_frags(n0, c1, t1, r1, 0);
        mma(f1);
        if (more) {
            load_frags(f1, c1, t1, r1, 1);
            f0 = n0;
        }
        ring = r1; cc = c1; tap = t1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sink pieces of the last steps

    // ---- epilogue (same D layout as the kernels above) ----
    static_for<TP>([&](auto B) {
        constexpr int b = decltype(B)::v;
        const int m = m0 + (wp * TP + b) * 32 + l31;
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            static_for<4>([&](auto Q) {
                constexpr int q = decltype(Q)::v;
                const int c = c0 + (wc * TC + a) * 32 + 8 * q + 4 * half;
                float v[4] = {acc[a][b][4 * q + 0], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                if (m < p.M && c < p.Cout) epilogue_store4(p, m, c, v);
            });
        });
    });

    if (p.stats) {
        __syncthreads();  // the last step's fragment reads are done: LDS can be reused
        float *red = reinterpret_cast<float *>(smem_b3);  // [WP][2][BN]
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

template <int BM, int BN, int WP, int WC, int ABL = 0>
static int launch_b3_dma(const ConvArgs &a, hipStream_t st) {
    if ((long long)BN * a.Kpad * 2 >= (1ll << 31))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3, LDS-DMA): weight panel exceeds 31-bit offsets");
    const size_t lds = (size_t)2 * (2 * BM + 2 * BN) * 64;
    auto k = conv_b3_dma_kernel<BM, BN, WP, WC, ABL>;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CER_LAUNCH(k, dim3(a.tiles_m * a.tiles_n, 1, a.split_k), dim3(256), lds, st, a);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

template <int BM, int BN, int WP, int WC, int NW = 4, int STAGES = 2>
static int launch_b3_dma16(const ConvArgs &a, hipStream_t st) {
    if ((long long)BN * a.Kpad * 2 >= (1ll << 31))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3, LDS-DMA): weight panel exceeds 31-bit offsets");
    const size_t lds = (size_t)STAGES * (2 * BM + 2 * BN) * 64;
    auto k = conv_b3_dma16_kernel<BM, BN, WP, WC, NW, STAGES>;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CER_LAUNCH(k, dim3(a.tiles_m * a.tiles_n, 1, a.split_k), dim3(NW * 64), lds, st, a);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

// pieces (1 KiB = 16 window rows) per activation plane, or 0 when the window kernel cannot take the conv
static int win_pieces(const ConvArgs &a, int bm, int bn) {
    if (a.stride != 1 || a.Ho != a.H || a.Wo != a.W || a.split_k != 1) return 0;
    const int T = a.KH * a.KW;
    const long long win = (long long)bm + (long long)(a.KH - 1) * a.dil_h * a.W + (a.KW - 1) * a.dil_w;
    const long long np = (win + 15) / 16;
    if (np > 8ll * (T - 1) || np > 64) return 0;  // the next chunk's window is issued during taps 0..T-2
    const long long lds = 4 * np * 1024 + 3ll * 2 * bn * 64 + 1024 + 64;
    return lds <= 160 * 1024 ? (int)np : 0;
}

template <int BM, int BN, int WP, int WC>
static int launch_b3_win(const ConvArgs &a, hipStream_t st) {
    const int np = win_pieces(a, BM, BN);
    if (!np) return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3, window): needs a stride-1 same conv without split-K whose window fits the LDS");
    if ((long long)BN * a.Kpad * 2 >= (1ll << 31) || ((long long)BM + 64 * 16) * a.x_ld * 2 >= (1ll << 31))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3, window): operand panel exceeds 31-bit offsets");
    const size_t lds = (size_t)4 * np * 1024 + (size_t)3 * 2 * BN * 64 + 1024 + 64;
    auto k = conv_b3_win_kernel<BM, BN, WP, WC>;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CER_LAUNCH(k, dim3(a.tiles_m * a.tiles_n, 1, 1), dim3(512), lds, st, a, np);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

// tile ids (desc.tile): 0 auto; 31/32/33 = window kernel 256x128 / 256x64 / 128x128 (8 waves); 11/12/14/15 = LDS-DMA staged 128x128 / 128x64 / 64x128 / 64x64 (BK32, two LDS buffers); 1 = 128x128 BK32, 2 = 128x64 BK32, 3 = 128x128 BK64, 4 = 64x128 BK32, 5 = 64x64 BK32,
// 6 = ping-pong 128x128 (measured 5-10 % slower than 1: kept as a tested A/B variant), 9 = timing-only ablation.
// Measured and dropped: 256x128 / 128x256 tiles with 128x64 per wave (2 waves/SIMD, -3..-8 %), 256x64 for the
// Cout = 64 layers (-13 % vs 128x64).  The no-staging
// ablation reaches ~500 TFLOP/s effective and LDS store bandwidth (32 KB per K step at ~80 B/clk against 768
// MFMA cycles) is what the register-staged structure runs into; LDS-DMA staging is the next step.
int conv_b3_tile_dims(int tile, int Cout, long long M, int K, int &bm, int &bn, int &bk) {
    if (tile == 0) {
        // measured per layer shape on MI355X (tools/bench_conv.py): the LDS-DMA kernels beat their
        // register-staged twins everywhere and the 16x16x32 MFMA shape beats 32x32x16 by 8-18 % (higher sustained clock);
        // with the compact LDS epilogue the 128x128 tile wins for every Cout >= 128 (also at K = 576 / 1152), 128x64
        // or (large M) 256x64 serve Cout <= 64, small grids take 64-row tiles
        const long long t128 = (M + 127) / 128 * ((Cout + 127) / 128);
        (void)K;
        if (Cout <= 64) tile = (M + 255) / 256 >= 512 ? 48 : 42;  // 256x64: 226 vs 204 TF/s on 64->64 @224x224
        else if (t128 >= 512) tile = 41;
        else tile = Cout >= 128 ? 44 : 45;
    }
    switch (tile) {
        case 1: bm = 128; bn = 128; bk = 32; break;
        case 2: bm = 128; bn = 64; bk = 32; break;
        case 3: bm = 128; bn = 128; bk = 64; break;
        case 4: bm = 64; bn = 128; bk = 32; break;
        case 5: bm = 64; bn = 64; bk = 32; break;
        case 6: bm = 128; bn = 128; bk = 32; break;  // ping-pong, 512 threads
        case 9: bm = 128; bn = 128; bk = 32; break;  // timing-only ablation of tile 1
        case 11: bm = 128; bn = 128; bk = 32; break;
        case 12: bm = 128; bn = 64; bk = 32; break;
        case 14: bm = 64; bn = 128; bk = 32; break;
        case 15: bm = 64; bn = 64; bk = 32; break;
        case 21: case 22: case 23: bm = 128; bn = 128; bk = 32; break;  // timing-only ablations of tile 11
        case 41: bm = 128; bn = 128; bk = 32; break;  // 41/42/44/45: tiles 11/12/14/15 on v_mfma_f32_16x16x32_bf16
        case 42: bm = 128; bn = 64; bk = 32; break;
        case 44: bm = 64; bn = 128; bk = 32; break;
        case 45: bm = 64; bn = 64; bk = 32; break;
        case 48: bm = 256; bn = 64; bk = 32; break;   // 4 waves x (64 pixels x 64 couts)
        case 52: bm = 128; bn = 64; bk = 32; break;   // tile 42 with a 3-deep LDS ring (A/B variant: -2..-10 %, 2 blocks/CU)
        case 46: bm = 256; bn = 256; bk = 32; break;  // 8 waves (A/B variant: ties tile 41; 256x128x8 waves and 256x64x4 waves lost 3-18 %)
        case 31: bm = 256; bn = 128; bk = 32; break;
        case 32: bm = 256; bn = 64; bk = 32; break;
        case 33: bm = 128; bn = 128; bk = 32; break;
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
        case 11: return launch_b3_dma<128, 128, 2, 2>(a, st);
        case 12: return launch_b3_dma<128, 64, 2, 2>(a, st);
        case 14: return launch_b3_dma<64, 128, 1, 4>(a, st);
        case 15: return launch_b3_dma<64, 64, 2, 2>(a, st);
        case 41: return launch_b3_dma16<128, 128, 2, 2>(a, st);
        case 42: return launch_b3_dma16<128, 64, 2, 2>(a, st);
        case 44: return launch_b3_dma16<64, 128, 1, 4>(a, st);
        case 45: return launch_b3_dma16<64, 64, 2, 2>(a, st);
        case 48: return launch_b3_dma16<256, 64, 4, 1>(a, st);
        case 52: return launch_b3_dma16<128, 64, 2, 2, 4, 3>(a, st);
        case 46: return launch_b3_dma16<256, 256, 2, 4, 8>(a, st);
        case 31: return launch_b3_win<256, 128, 4, 2>(a, st);
        case 32: return launch_b3_win<256, 64, 8, 1>(a, st);
        case 33: return launch_b3_win<128, 128, 2, 4>(a, st);
        case 21: return launch_b3_dma<128, 128, 2, 2, 1>(a, st);
        case 22: return launch_b3_dma<128, 128, 2, 2, 2>(a, st);
        case 23: return launch_b3_dma<128, 128, 2, 2, 3>(a, st);
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (bf16x3): unknown tile id");
    }
}

}  // namespace cer

using namespace cer;

extern "C" int cer_conv2d_b3_tile(const cer_conv_desc *d) {
    if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return 0;
    int bm, bn, bk;
    return conv_b3_tile_dims(d->tile, d->Cout, (long long)d->N * d->Ho * d->Wo, cer_conv_kpad(d->KH, d->KW, d->Cin), bm, bn, bk);
}

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
