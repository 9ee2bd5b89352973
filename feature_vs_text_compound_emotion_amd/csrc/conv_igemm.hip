// conv_igemm.hip -- implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
// One kernel serves every dense contraction on the hot path (IR-50 convs, the
// IR-50 head FC, VGGish convs/FCs, TCN causal convs, all Linear layers): the
// output tile is [BN couts] x [BM pixels], the reduction runs over
// (kh, kw, cin) in steps of 32 (16 in the BK16 variants), operands are staged
// global -> registers -> LDS (single buffer by default: the smaller footprint
// admits one more block per CU, measured +15 % over double buffering) and
// consumed by v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//
// Operand roles are chosen so that the accumulator's register index runs over
// output channels: D[i = cout][j = pixel].  Each lane then owns 4 consecutive
// couts of one pixel per register quad, which makes the NHWC epilogue
// (bias / PReLU / residual / mask / store) 16-byte wide.
//
// LDS image: [row][k] with k contiguous and a 36-float row pitch.  A lane
// reads 4 consecutive k of its row with one ds_read_b128 and feeds them to 4
// successive MFMAs; lane-half h takes k = 8g+4h+j, which is just a fixed
// permutation of the reduction order shared by both operands.  The 36-float
// pitch makes ds_read_b128 conflict-free (36*r mod 64 is distinct for the 16
// rows of a lane group) and ds_write_b128 conflict-free (8 lanes = 128 B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cer_internal.h"
#include "conv_common.h"

namespace cer {

// VAR bit 0: s_setprio(1) around the MFMA cluster; bit 1: single LDS buffer (two barriers per
// step, half the LDS -> more resident blocks per CU).
template <int BM, int BN, int WP, int WC, bool VEC, int VAR, int BKT>
__global__ __launch_bounds__(WP * WC * 64, 3) void conv_igemm_kernel(ConvArgs p) {
    constexpr int NT = WP * WC * 64;   // threads per block (1 or 4 waves)
    constexpr int CH = BKT / 4;        // 16-byte chunks per staged row
    constexpr int RPP = NT / CH;       // rows staged per pass
    constexpr int PIT = BKT + 4;       // LDS row pitch (floats): conflict-free ds_read_b128
    constexpr int NG = BKT / 8;        // 8-wide reduction groups per step
    constexpr bool PRIO = (VAR & 1) != 0, SINGLE = (VAR & 2) != 0, BLKPRIO = (VAR & 4) != 0;
    // timing-only ablations (WRONG results by construction, never selected by the picker):
    constexpr bool ABL_NOSTAGE = (VAR & 8) != 0;   // no global loads / LDS stores inside the K loop
    constexpr bool ABL_NOBAR = (VAR & 16) != 0;    // ... and no barriers either
    constexpr int NBUF = SINGLE ? 1 : 2;
    constexpr int TP = BM / (32 * WP);  // 32-pixel MFMA tiles per wave
    constexpr int TC = BN / (32 * WC);  // 32-cout MFMA tiles per wave
    constexpr int XR = BM / RPP;        // activation rows staged per thread
    constexpr int WR = BN / RPP;        // weight rows staged per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;                         // [NBUF][BM][PIT]
    float *Ws = smem + NBUF * BM * PIT;     // [NBUF][BN][PIT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wp = wave % WP, wc = wave / WP;
    const int half = lane >> 5, l31 = lane & 31;

    // XCD-aware tile order: blocks b and b+8 share an L2, so give each XCD a
    // contiguous run of tiles (cout tile fastest: neighbours share activations).
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

    // ---- staging assignment: thread -> (row srow+RPP*i, 16-byte chunk) ----
    const int chunk = tid % CH, srow = tid / CH;
    int x_pix[XR], x_hi0[XR], x_wi0[XR];
    bool x_ok[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        int m = m0 + srow + RPP * i;
        x_ok[i] = m < p.M;
        int mm = x_ok[i] ? m : 0;
        int hw = p.Ho * p.Wo;
        int n = mm / hw, r = mm - n * hw;
        int ho = r / p.Wo, wo = r - ho * p.Wo;
        x_pix[i] = n;  // image index; pixel offset formed per tap
        x_hi0[i] = ho * p.stride - p.pad_t;
        x_wi0[i] = wo * p.stride - p.pad_l;
    }
    // Fast addressing (VEC path): every load is  scalar_base(step) + 32-bit per-thread offset.
    //   x: byte offset of the row's (kh = kw = 0) tap relative to the tile's first row, and a bit
    //      mask of the taps that fall inside the image (zero padding) -> no per-step address math;
    //   w: row offset inside the [Cout][Kpad] panel.
    unsigned x_rel[XR], x_taps[XR], w_rel[WR];
    bool w_ok[WR];
    long long tile_base = 0;  // bytes from p.x to the first row's (0,0) tap (may be negative: padding)
    if constexpr (VEC) {
        {
            const int hw = p.Ho * p.Wo;
            const int mm = m0 < p.M ? m0 : 0;
            const int n = mm / hw, r = mm - n * hw;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            tile_base = ((long long)(n * p.H + ho * p.stride - p.pad_t) * p.W + (wo * p.stride - p.pad_l)) * p.x_ld * 4;
        }
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const long long rb = ((long long)(x_pix[i] * p.H + x_hi0[i]) * p.W + x_wi0[i]) * p.x_ld * 4;
            x_rel[i] = x_ok[i] ? (unsigned)(rb - tile_base) + chunk * 16u : 0u;
            unsigned bits = 0;
            if (x_ok[i]) {
                for (int t = 0; t < p.KH * p.KW; ++t) {
                    const int kh = t / p.KW, kw = t - kh * p.KW;
                    const int hi = x_hi0[i] + kh * p.dil_h, wi = x_wi0[i] + kw * p.dil_w;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) bits |= 1u << t;
                }
            }
            x_taps[i] = bits;
        }
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int crow = c0 + srow + RPP * i;
        w_ok[i] = crow < p.Cout;
        w_rel[i] = (unsigned)(((size_t)(srow + RPP * i) * p.Kpad + chunk * 4) * 4);
    }

    float4 xr[XR], wr[WR];
    unsigned xvalid = 0;
    float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);

    auto load_step = [&](int s) {
        if constexpr (VEC) {
            const int tap = s / p.cin_steps, cc = s - tap * p.cin_steps;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int cbase = cc * BKT + chunk * 4;
            // wave-uniform (scalar) base of this step: tile origin + tap displacement + channel chunk
            const char *sbase = reinterpret_cast<const char *>(p.x) + tile_base +
                                ((long long)(kh * p.dil_h * p.W + kw * p.dil_w) * p.x_ld + cc * BKT) * 4;
            const unsigned tapbit = 1u << tap;
            xvalid = 0;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                if (x_taps[i] & tapbit) {
                    xr[i] = *reinterpret_cast<const float4 *>(sbase + x_rel[i]);
                    xvalid |= 1u << i;
                } else {
                    xr[i] = make_float4(0, 0, 0, 0);
                }
            }
            if (p.in_scale) {
                sc = *reinterpret_cast<const float4 *>(p.in_scale + cbase);
                sh = *reinterpret_cast<const float4 *>(p.in_shift + cbase);
            }
        } else {
            // small-Cin gather: k -> (tap, c), element-wise (stem conv, Cin = 1 or 3)
            const int K = p.KH * p.KW * p.Cin;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int k = s * BKT + chunk * 4 + j;
                    float v = 0.f;
                    if (k < K && x_ok[i]) {
                        int tap = k / p.Cin, c = k - tap * p.Cin;
                        int kh = tap / p.KW, kw = tap - kh * p.KW;
                        int hi = x_hi0[i] + kh * p.dil_h, wi = x_wi0[i] + kw * p.dil_w;
                        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                            size_t off = p.x_nchw
                                ? ((size_t)(x_pix[i] * p.Cin + c) * p.H + hi) * p.W + wi
                                : ((size_t)(x_pix[i] * p.H + hi) * p.W + wi) * p.x_ld + c;
                            v = p.x[off];
                            if (p.in_scale) v = v * p.in_scale[c] + p.in_shift[c];
                        }
                    }
                    e[j] = v;
                }
                xr[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        {
            const char *wbase = reinterpret_cast<const char *>(p.w) + ((size_t)c0 * p.Kpad + (size_t)s * BKT) * 4;
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                if (w_ok[i]) wr[i] = *reinterpret_cast<const float4 *>(wbase + w_rel[i]);
                else wr[i] = make_float4(0, 0, 0, 0);
            }
        }
    };

    auto store_step = [&](int buf) {
        float *xs = Xs + buf * BM * PIT, *ws = Ws + buf * BN * PIT;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            float4 v = xr[i];
            if constexpr (VEC) {
                if (p.in_scale && ((xvalid >> i) & 1u)) {
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y;
                    v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                }
            }
            *reinterpret_cast<float4 *>(xs + (srow + RPP * i) * PIT + chunk * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < WR; ++i)
            *reinterpret_cast<float4 *>(ws + (srow + RPP * i) * PIT + chunk * 4) = wr[i];
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
        store_step(0);
    }
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        const int buf = SINGLE ? 0 : ((s - s_begin) & 1);
        if constexpr (!ABL_NOSTAGE) {
            if (s + 1 < s_end) load_step(s + 1);
        }
        if constexpr (BLKPRIO) {
            // same priority for the 4 waves of a block on their 4 SIMDs -> blocks advance in lockstep
            switch ((blockIdx.x >> 3) & 3) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        } else if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        const float *xs = Xs + buf * BM * PIT + (wp * TP * 32 + l31) * PIT + half * 4;
        const float *ws = Ws + buf * BN * PIT + (wc * TC * 32 + l31) * PIT + half * 4;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 af[TC], bf[TP];
#pragma unroll
            for (int a = 0; a < TC; ++a) af[a] = *reinterpret_cast<const float4 *>(ws + a * 32 * PIT + g * 8);
#pragma unroll
            for (int b = 0; b < TP; ++b) bf[b] = *reinterpret_cast<const float4 *>(xs + b * 32 * PIT + g * 8);
#pragma unroll
            for (int a = 0; a < TC; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
                }
        }
        if constexpr (PRIO && !BLKPRIO) __builtin_amdgcn_s_setprio(0);
        if constexpr (SINGLE && !ABL_NOBAR) __syncthreads();  // everyone is done reading the only buffer
        if constexpr (!ABL_NOSTAGE) {
            if (s + 1 < s_end) store_step(SINGLE ? 0 : (buf ^ 1));
        }
        if constexpr (!ABL_NOBAR) __syncthreads();
    }

    // ---- epilogue: D[i = cout][j = pixel]; reg r -> cout (r&3) + 8*(r>>2) + 4*half ----
    // (compile-time indices: a runtime-indexed accumulator array would live in scratch)
    const bool fast = epi_fast(p);   // one uniform decision per launch; the general epilogue otherwise
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
                        epi_store4_direct(p, fast, m, c, v);
                    }
                }
            });
        });
    });

    // ---- optional per-channel batch statistics of the raw result (train-mode BatchNorm that
    // follows this conv): deterministic per-tile partials, reduced later by bn_finalize ----
    if (p.stats) {
        float *red = smem;  // [WP][2][BN]; the K loop ended on a barrier, LDS is free
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

// Sum split-K slabs and apply the fused epilogue.  One thread per 4 couts.
__global__ void splitk_reduce_kernel(ConvArgs p, const float *partial) {
    const int c4 = (p.Cout + 3) >> 2;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)p.M * c4) return;
    int m = (int)(idx / c4), c = (int)(idx - (size_t)m * c4) * 4;
    float v[4] = {0, 0, 0, 0};
    const bool vec = ((p.Cout & 3) == 0);
    for (int s = 0; s < p.split_k; ++s) {
        const float *src = partial + ((size_t)s * p.M + m) * p.Cout + c;
        if (vec) {
            float4 t = *reinterpret_cast<const float4 *>(src);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < p.Cout) v[e] += src[e];
        }
    }
    epi_store4_direct(p, epi_fast(p), m, c, v);
}

// ---------------------------------------------------------------- host side
template <int BM, int BN, int WP, int WC, int VAR, int BKT = 32>
static int launch_cfg(const ConvArgs &a, bool vec, hipStream_t st) {
    const size_t lds = (size_t)((VAR & 2) ? 1 : 2) * (BM + BN) * (BKT + 4) * sizeof(float);
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.split_k), block(WP * WC * 64);
    if (vec) {
        auto k = conv_igemm_kernel<BM, BN, WP, WC, true, VAR, BKT>;
        if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a);
    } else {
        auto k = conv_igemm_kernel<BM, BN, WP, WC, false, VAR, BKT>;
        if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

// Tile shapes: 1 = 128x128 (3 blocks/CU), 2 = 128x64 (4/CU), 4 = 64x128 (4/CU), 5 = 64x64 (5/CU);
// all single-LDS-buffer (VAR 2, measured +15 % over double buffering because the smaller LDS
// footprint admits one more block per CU; s_setprio made no difference in this structure).  Relative per-FLOP rates are measured
// (tools/bench_conv.py); the picker minimises rounds x tile work, which matters when the grid is
// only a few rounds deep (40x40 frames: 1600 tiles of 128x128 on 768 slots).
struct TileInfo { int id, bm, bn, per_cu; float rate; };
static const TileInfo kTiles[] = {{1, 128, 128, 3, 1.00f}, {2, 128, 64, 4, 0.93f}, {4, 64, 128, 4, 0.90f}, {5, 64, 64, 5, 0.75f}};

static int pick_tile(const cer_conv_desc *d, int M) {
    if (d->tile) return d->tile;
    int best = 5;
    float best_t = 3.4e38f;
    for (const TileInfo &t : kTiles) {
        if (t.bn > 64 && d->Cout <= 64) continue;
        const long long blocks = (long long)((M + t.bm - 1) / t.bm) * ((d->Cout + t.bn - 1) / t.bn) * d->split_k;
        const long long slots = 256ll * t.per_cu;
        const long long rounds = (blocks + slots - 1) / slots;
        // a partially filled round still runs at the per-block latency, but with fewer co-resident
        // blocks each block runs a little faster: count it as at least half a round
        const float frac = (float)(blocks - (rounds - 1) * slots) / (float)slots;
        const float eff_rounds = (float)(rounds - 1) + (frac < 0.5f ? 0.5f + frac * 0.5f : (frac < 1.f ? 0.75f + frac * 0.25f : 1.f));
        const float time = eff_rounds * (float)(t.bm * t.bn) * (float)t.per_cu / t.rate;
        if (time < best_t) { best_t = time; best = t.id; }
    }
    return best;
}

static void tile_dims(int tile, int &bm, int &bn) {
    switch (tile) {
        case 1: bm = 128; bn = 128; break;
        case 2: bm = 128; bn = 64; break;
        case 4: bm = 64; bn = 128; break;
        default: bm = 64; bn = 64; break;
    }
}

template <int VAR>
static int launch_shape(int shape, const ConvArgs &a, bool vec, hipStream_t st) {
    switch (shape) {
        case 1: return launch_cfg<128, 128, 2, 2, VAR>(a, vec, st);
        case 2: return launch_cfg<128, 64, 2, 2, VAR>(a, vec, st);
        case 4: return launch_cfg<64, 128, 1, 4, VAR>(a, vec, st);
        case 5: return launch_cfg<64, 64, 2, 2, VAR>(a, vec, st);
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d_fwd: unknown tile id");
    }
}

}  // namespace cer

using namespace cer;

extern "C" int cer_conv_kpad(int KH, int KW, int Cin) {
    int K = KH * KW * Cin;
    return (K + BK - 1) / BK * BK;
}

static int validate_desc(const cer_conv_desc *d) {
    if (!d) return cer_set_error(CER_ERR_INVALID_ARG, "conv desc is NULL");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 ||
        d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->dil_h <= 0 || d->dil_w <= 0 || d->split_k < 1)
        return cer_set_error(CER_ERR_INVALID_ARG, "conv desc: non-positive dimension");
    if ((long long)d->N * d->Ho * d->Wo >= (1ll << 31) || (long long)d->N * d->H * d->W >= (1ll << 31))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv desc: more than 2^31 pixels");
    if (d->x_nchw && (d->Cin % 32) == 0)
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv desc: NCHW input only on the small-Cin path");
    return CER_OK;
}

namespace cer {
int conv_b3_tile_dims(const cer_conv_desc *d, int &bm, int &bn, int &bk);
int conv_n16_tile_dims(const cer_conv_desc *d, int &bm, int &bn, int &bk);
}

extern "C" int cer_conv2d_stats_tiles(const cer_conv_desc *d, int kernel_family) {
    if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
    const int M = d->N * d->Ho * d->Wo;
    int bm, bn, bk;
    if (kernel_family == 2) {
        if (!conv_n16_tile_dims(d, bm, bn, bk)) return 0;
    } else if (kernel_family == 1) {
        const int t = conv_b3_tile_dims(d, bm, bn, bk);
        if (!t) return 0;
        if (t == 6) return 2 * ((M + bm - 1) / bm);  // the ping-pong kernel writes one row per pixel half-tile
    } else {
        tile_dims(pick_tile(d, M), bm, bn);
    }
    return (M + bm - 1) / bm;
}

extern "C" size_t cer_conv2d_workspace_bytes(const cer_conv_desc *d) {
    if (!d || d->split_k <= 1) return 0;
    return (size_t)d->split_k * d->N * d->Ho * d->Wo * d->Cout * sizeof(float);
}

namespace cer {
int conv_b3_launch(int tile, const ConvArgs &a, hipStream_t st);
int conv_n16_launch(int tile, const ConvArgs &a, hipStream_t st);
bool conv_n16_p64_ok(const ConvArgs &a);
}  // namespace cer

extern "C" int cer_conv2d_run(const cer_conv_desc *d, const cer_conv_io *io, void *workspace, size_t workspace_bytes,
                              void *stream) {
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!io) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: io block is NULL");
    if (d->storage != CER_STORE_NONE && d->storage != CER_STORE_BF16 && d->storage != CER_STORE_F16)
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: unknown storage type");
    const bool narrow = d->storage != CER_STORE_NONE;
    const bool n16 = narrow && io->x_hi != nullptr;   // narrow operands -> conv_n16 kernels
    const bool b3 = !narrow && io->x_hi != nullptr;   // split operands -> bf16x3 kernels
    if (narrow && (io->x_lo || io->w_lo || io->y_lo || io->res_lo || io->y2_hi || io->y2_lo))
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (narrow): 16-bit tensors are single planes (every *_lo and y2 pointer must be NULL)");
    if (n16) {
        if (!io->w_hi) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (narrow): x_hi and w_hi must both be given");
        if (io->in_scale || io->mask || io->aux || d->x_nchw)
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow): no input affine / mask / aux / NCHW input");
    } else if (b3) {
        if (!io->x_lo || !io->w_hi || !io->w_lo)
            return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (bf16x3): x_hi, x_lo, w_hi, w_lo must all be given");
        if (io->in_scale || io->mask || io->aux || d->x_nchw)
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3): no input affine / mask / aux / NCHW input");
    } else if (!io->x || !io->w) {
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: x and w must be non-NULL");
    }
    if (!io->y && !io->y_hi && !io->y2_hi) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: no output tensor");
    if ((io->in_scale == nullptr) != (io->in_shift == nullptr))
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: in_scale and in_shift go together");
    if (!narrow && ((io->y_hi == nullptr) != (io->y_lo == nullptr) || (io->res_hi == nullptr) != (io->res_lo == nullptr) ||
                    (io->y2_hi == nullptr) != (io->y2_lo == nullptr) || (io->y2_hi && (!io->s2 || !io->t2))))
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: split tensors need both planes (and s2/t2 for the second output)");
    if (io->residual && io->res_hi) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: give the residual as fp32 OR split");
    if (d->act1 == CER_ACT_PRELU && !io->alpha) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: PReLU needs alpha");
    const bool has_res = io->residual || io->res_hi;
    if (has_res && (d->res_stride <= 0 || d->Hr <= 0 || d->Wr <= 0 || (d->Ho - 1) * d->res_stride >= d->Hr ||
                    (d->Wo - 1) * d->res_stride >= d->Wr))
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: residual geometry out of range");
    if (io->stats && d->split_k > 1)
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d: batch statistics are not available with split-K");
    if ((io->y_hi || io->y2_hi || io->res_hi) && (d->Cout & 3))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d: split outputs / residual need Cout % 4 == 0");
    ConvArgs a{};
    a.x = io->x; a.w = io->w; a.in_scale = io->in_scale; a.in_shift = io->in_shift; a.bias = io->bias; a.alpha = io->alpha;
    a.res = io->residual; a.mask = io->mask; a.y = io->y; a.aux = io->aux; a.stats = io->stats;
    a.x_hi = io->x_hi; a.x_lo = io->x_lo; a.w_hi = io->w_hi; a.w_lo = io->w_lo;
    a.res_hi = io->res_hi; a.res_lo = io->res_lo; a.y_hi = io->y_hi; a.y_lo = io->y_lo;
    a.s2 = io->s2; a.t2 = io->t2; a.y2_hi = io->y2_hi; a.y2_lo = io->y2_lo;
    a.bias9 = io->bias9;
    a.narrow = d->storage;
    if (io->bias9 && (io->bias || d->stride != 1 || d->Ho != d->H || d->Wo != d->W || d->H < 2 || d->W < 2 || d->split_k > 1))
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: bias9 replaces bias and needs a stride-1 same conv on >= 2x2 images without split-K");
    if (d->x_s2d || d->y_s2d) {
        if (!b3 && !n16) return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d: space-to-depth tensors exist in the bf16x3 and narrow modes only");
        if (d->x_s2d && d->x_ld > 0) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: x_s2d fixes the input pitch (x_ld must be 0)");
        if (d->y_s2d && ((d->Ho | d->Wo) & 1 || io->y || io->y2_hi || has_res || io->mask || io->aux || d->y_ld > 0 || !io->y_hi))
            return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: y_s2d needs even Ho / Wo and a plain 16-bit output (no fp32 / second "
                                                      "output, residual, mask, aux or y_ld)");
    }
    a.x_s2d = d->x_s2d != 0; a.y_s2d = d->y_s2d != 0;
    a.x_ld = d->x_s2d ? 4 * d->Cin : (d->x_ld > 0 ? d->x_ld : d->Cin);
    a.y_ld = d->y_ld > 0 ? d->y_ld : d->Cout;
    if (a.x_ld < d->Cin || a.y_ld < d->Cout || ((d->Cin % 32) == 0 && (a.x_ld & 3) != 0))
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: x_ld/y_ld smaller than the channel count or x_ld % 4 != 0");
    if (d->x_nchw && d->x_ld > 0) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: x_ld is meaningless for NCHW input");
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.dil_h = d->dil_h; a.dil_w = d->dil_w;
    a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.x_nchw = d->x_nchw;
    a.res_stride = d->res_stride; a.Hr = d->Hr; a.Wr = d->Wr; a.act1 = d->act1; a.act2 = d->act2;
    a.slope = d->slope;
    a.Kpad = cer_conv_kpad(d->KH, d->KW, d->Cin);
    a.M = d->N * d->Ho * d->Wo;
    const bool vec = (d->Cin % 32) == 0;
    int tile, bm, bn, bk, esz;
    if (n16) {
        tile = conv_n16_tile_dims(d, bm, bn, bk);
        if (!tile) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (narrow): unknown tile id (space-to-depth input: 81 / 82 only)");
        if (d->y_s2d && tile != 71 && tile != 72 && tile != 78 && tile != 73 && tile != 76 && tile != 77 && tile != 79)
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow): y_s2d is written by the window / patch kernels only "
                                                      "(cer_conv2d_n16_tile(desc) in {71, 72, 73, 76, 77, 78, 79})");
        if (d->Cin % bk != 0 || (a.x_ld & 7))
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow): Cin must be a multiple of 64 and x_ld of 8");
        esz = 2;
    } else if (b3) {
        tile = conv_b3_tile_dims(d, bm, bn, bk);
        if (!tile) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (bf16x3): unknown tile id (space-to-depth input: 51 / 52 only)");
        if (d->y_s2d && tile != 53 && tile != 56 && tile != 58 && tile != 59)
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3): y_s2d is written by the window / patch kernels only "
                                                      "(cer_conv2d_b3_tile(desc) in {53, 56, 58, 59})");
        if (d->Cin % bk != 0 || (a.x_ld & 7))
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3): Cin must be a multiple of the K step and x_ld of 8");
        esz = 2;
    } else {
        tile = pick_tile(d, a.M);
        if (tile != 1 && tile != 2 && tile != 4 && tile != 5) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d: unknown tile id");
        bk = 32;
        tile_dims(tile, bm, bn);
        esz = 4;
    }
    a.cin_steps = (vec || b3 || n16) ? d->Cin / bk : 1;
    a.steps = a.Kpad / bk;
    a.split_k = d->split_k > a.steps ? a.steps : d->split_k;
    a.steps_per_split = (a.steps + a.split_k - 1) / a.split_k;
    a.split_k = (a.steps + a.steps_per_split - 1) / a.steps_per_split;
    a.tiles_m = (a.M + bm - 1) / bm;
    a.tiles_n = (a.Cout + bn - 1) / bn;
    if (vec || b3 || n16) {
        // the staging path addresses x and w as scalar base + 32-bit per-thread byte offset
        if (d->KH * d->KW > 32) return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d: more than 32 filter taps");
        // the per-row offsets are UNSIGNED distances from the tile's first row: output pixels must map to non-decreasing
        // input addresses, also across row and image boundaries (an out_hw larger than the natural one breaks that:
        // a wrapped offset would read 4 GiB away)
        if (d->Wo > d->W / d->stride + 1 ||
            (long long)d->H * d->W < (long long)(d->Ho - 1) * d->stride * d->W + (long long)(d->Wo - 1) * d->stride)
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d: the output grid outruns the input (Ho/Wo too large for H/W and stride)");
        const long long span_px = (long long)bm * d->stride * d->stride + 2ll * d->H * d->W + (long long)d->W * d->stride + 2;
        if (span_px * a.x_ld * esz >= (1ll << 31) || (long long)bn * a.Kpad * esz >= (1ll << 32))
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d: tile footprint exceeds 32-bit staging offsets");
    }
    hipStream_t st = (hipStream_t)stream;
    ConvArgs fin = a;
    if (a.split_k > 1) {
        size_t need = (size_t)a.split_k * a.M * a.Cout * sizeof(float);
        if (!workspace || workspace_bytes < need) return cer_set_error(CER_ERR_WORKSPACE, "conv2d: split-K workspace too small");
        a.y = (float *)workspace;
    }
    if (n16) {
        // the picker chose the persistent Cin == 64 patch kernel from the geometry; a launch whose epilogue it does not take
        // (conv_n16_p64_ok) runs on the one-patch-per-block kernel of the same tile shape
        if (tile == 79 && d->tile == 0 && !conv_n16_p64_ok(a)) tile = 71;
        rc = conv_n16_launch(tile, a, st);
    } else if (b3) {
        rc = conv_b3_launch(tile, a, st);
    } else {
        // the shipped variant: single LDS buffer (VAR 2).  The double-buffered / setprio / BK = 16 / ablation builds that
        // were measured against it in round 1 are in the history (they were 96 instantiations and most of the build time)
        rc = launch_shape<2>(tile, a, vec, st);
    }
    if (rc) return rc;
    if (a.split_k > 1) {
        fin.split_k = a.split_k;
        size_t n = (size_t)a.M * ((a.Cout + 3) / 4);
        CER_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, fin, (const float *)workspace);
        CER_HIP_CHECK(hipGetLastError());
    }
    return CER_OK;
}

extern "C" int cer_conv2d_fwd(const cer_conv_desc *d, const float *x, const float *w, const float *in_scale,
                              const float *in_shift, const float *bias, const float *alpha, const float *residual,
                              const float *mask, float *y, float *aux, float *stats, void *workspace,
                              size_t workspace_bytes, void *stream) {
    if (!x || !w || !y) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d_fwd: x, w, y must be non-NULL");
    cer_conv_io io{};
    io.x = x; io.w = w; io.in_scale = in_scale; io.in_shift = in_shift; io.bias = bias; io.alpha = alpha;
    io.residual = residual; io.mask = mask; io.y = y; io.aux = aux; io.stats = stats;
    return cer_conv2d_run(d, &io, workspace, workspace_bytes, stream);
}
