// conv_n16.hip -- implicit-GEMM convolution on NARROW (single 16-bit plane) operands: bf16 or IEEE half storage,
// one v_mfma_f32_16x16x32_{bf16,f16} per 32-deep product, fp32 accumulation, fp32 fused epilogue.
//
// This is the arithmetic of the reference's own GPU recipe (fp16 autocast + GradScaler, trainer.py:14-15,341,367;
// README example has --amp on) and of BASELINE cfg5 ("bf16 storage / fp32 accumulate").  Against the bf16x3 kernel
// (conv_b3.hip) it spends one MFMA instead of three per product and moves half the bytes, so the matrix roofline is the
// un-divided 2.5 PFLOP/s dense bf16/f16 peak.
//
// Structure (same family as conv_b3_dma16_kernel, re-sized for 1/3 of the MFMA time per byte):
//   * D[i = cout][j = pixel]: a lane owns 4 consecutive couts of one pixel;
//   * K step = 64 channels of one filter tap = one full 128-byte line per pixel / per weight row.  Operands go
//     L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB = 8 rows per wave instruction, no staging VGPRs), two
//     LDS stages, ONE barrier per K step; zero padding / ragged edges = a byte offset past the descriptor's record
//     count (range-checked loads return zeros);
//   * rows are unpadded 128-byte lines; the ds_read_b128 fragment reads (row = lane & 15, 16-byte k-chunk = lane >> 4,
//     + 4 for the second 32-deep half) are made conflict free by slot = chunk ^ ((row >> 1) & 7), applied on the DMA's
//     per-lane SOURCE address (the LDS image of a DMA is lane-linear) and on the read: the 16 lanes of every
//     ds_read_b128 lane group then hit 16 different 16-byte slots of the 256-byte bank row (tools/check_swizzle.py);
//   * K order is channel-chunk major (all KH*KW taps of channels [64cc, 64cc+64), then the next chunk): the taps
//     re-read the same input rows in consecutive steps while L2 still holds them;
//   * tiles up to 256 x 256 with 8 waves (128 pixels x 64 couts per wave: 0.375 LDS fragment reads per MFMA, 64 KiB
//     of operands per 2 x 32 MFMAs per wave);
//   * epilogue through LDS in passes (the accumulator tile is larger than the LDS for the big tiles): 16-byte
//     granules XOR-swizzled by (row & 15), then ONE compact loop in which consecutive lanes own consecutive couts of
//     a pixel row -> coalesced residual reads and stores; fused bias / border-dependent bias9 / PReLU / residual /
//     fp32, narrow outputs / per-tile BatchNorm statistics / split-K slabs.
#include "conv_n16.h"

namespace cer {

constexpr int n16_epilogue_rows(int BM, int BN, int WP) {
    // rows of the fp32 accumulator tile that fit the two staging buffers (2 * (BM + BN) * 128 bytes) at once
    int r = BM;
    while (r * BN * 4 > 2 * (BM + BN) * 128 && r > 16 * WP) r >>= 1;
    return r;
}

template <int BM, int BN, int WP, int WC, bool F16, int ILV = 1>
__global__ __launch_bounds__(WP * WC * 64, 2) void conv_n16_kernel(ConvArgs p) {
    constexpr int NW = WP * WC, NT = NW * 64;
    static_assert(BM / (16 * WP) >= 1 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "1-KiB DMA pieces (8 rows) are dealt round-robin to the waves");
    constexpr int BKT = 64, ROWB = BKT * 2;                // bytes per row
    constexpr int XP = BM / (8 * NW), WQ = BN / (8 * NW);  // DMA pieces per wave and step
    constexpr int PX = BM * ROWB, PW = BN * ROWB, BUF = PX + PW;
    constexpr int TP = BM / (16 * WP), TC = BN / (16 * WC);  // 16x16 tiles per wave
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_n16[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_n16);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % WP, wc = wave / WP;
    const int kg = lane >> 4, l15 = lane & 15;

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware remap (bijective): blocks that share an XCD's L2 take neighbouring tiles
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, c0 = tile_n * BN;
    const int split = blockIdx.z;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(p.steps, s_begin + p.steps_per_split);

    // ---- DMA assignment: wave w moves pieces w, w + NW, ... of both operands; in a piece lane l owns row l / 8,
    // LDS slot l % 8 and fetches the source chunk slot ^ ((row >> 1) & 7) ----
    const int prow = lane >> 3, slot = lane & 7;
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
        const int row = (wave + NW * i) * 8 + prow;
        const int chunk = slot ^ ((row >> 1) & 7);
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
        const int row = (wave + NW * i) * 8 + prow;
        const int chunk = slot ^ ((row >> 1) & 7);
        w_off[i] = c0 + row < p.Cout ? (unsigned)(((size_t)row * p.Kpad + chunk * 8) * 2) : OOB;
    }

    // One K step's DMA: prep() builds the two buffer descriptors (scalar work), piece(j) issues the j-th of the wave's
    // XP + WQ 1-KiB pieces.  In the main loop the pieces of step s+1 are spread over the MFMA groups of step s: a DMA
    // instruction costs the issuing wave 60-185 cycles of issue (MI355X_MICROARCH.md, cycle constants), which hides
    // behind the MFMAs of the same wave when they alternate, and is exposed (8 x ~100 cycles against 1024 MFMA cycles
    // per wave and step on the 256x256 tile) when all pieces are issued in one burst right after the barrier.
    __amdgpu_buffer_rsrc_t rx, rw;
    unsigned tapbit = 0;
    unsigned char *dma_dst = smem;
    auto prep = [&](int s, int buf) {
        const int T = p.KH * p.KW;
        const int cc = s / T, tap = s - cc * T;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const long long soff = tile_base + ((long long)(kh * p.dil_h * p.W + kw * p.dil_w) * p.x_ld + cc * BKT) * 2;
        const size_t woff = ((size_t)c0 * p.Kpad + (size_t)tap * p.Cin + (size_t)cc * BKT) * 2;
        char *xb = const_cast<char *>(reinterpret_cast<const char *>(p.x_hi)) + soff;
        char *wb = const_cast<char *>(reinterpret_cast<const char *>(p.w_hi)) + woff;
        rx = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)OOB, 0x00020000);
        rw = __builtin_amdgcn_make_buffer_rsrc(wb, 0, (int)OOB, 0x00020000);
        tapbit = 1u << tap;
        dma_dst = smem + buf * BUF + wave * 1024;
    };
    auto piece = [&](auto J) {
        constexpr int j = decltype(J)::v;
        if constexpr (j < XP) {
            const int vo = (int)((x_taps[j] & tapbit) ? x_off[j] : OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (n_lds_ptr_t)(dma_dst + j * (NW * 1024)), 16, vo, 0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (n_lds_ptr_t)(dma_dst + PX + (j - XP) * (NW * 1024)), 16,
                                                     (int)w_off[j - XP], 0, 0, 0);
        }
    };
    constexpr int NDMA = XP + WQ;

    n_f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    if (s_begin < s_end) {
        prep(s_begin, 0);
        static_for<NDMA>([&](auto J) { piece(J); });
    }

    // fragment addresses: row * 128 bytes + swizzled slot of k-chunk kg (second 32-deep half: ^ 64 bytes).
    // Pixel tiles are dealt round-robin to the WP pixel waves (tile t = b * WP + wp) so that an epilogue pass covers a
    // contiguous range of output rows.
    const int sw = (l15 >> 1) & 7;
    const int arow = PX + (wc * TC * 16 + l15) * ROWB + ((kg ^ sw) << 4);  // A = weights: row = cout
    const int brow = (wp * 16 + l15) * ROWB + ((kg ^ sw) << 4);            // B = activations: row = pixel
    // MFMA groups of one step: (kk, b) -> TC MFMAs each; DMA piece j goes in front of group j * NGRP / NDMA
    constexpr int NGRP = 2 * TP;
    constexpr int DGRP = ILV == 2 ? NGRP / 2 : NGRP;  // groups that carry DMA pieces
    // one K step on stage `cur`; DMA = the pieces of the next step are issued between its MFMA groups (the last step of
    // the loop is peeled so that the body has no branches and stays one scheduling region)
    auto step = [&](int cur, auto WITH_DMA) {
        constexpr bool dma = decltype(WITH_DMA)::v != 0;
        const unsigned char *S = smem + cur * BUF;
        auto lda = [&](int a, int kk) { return *reinterpret_cast<const n_u32x4 *>(S + ((arow + a * 16 * ROWB) ^ (kk << 6))); };
        auto ldb = [&](int b, int kk) { return *reinterpret_cast<const n_u32x4 *>(S + ((brow + b * WP * 16 * ROWB) ^ (kk << 6))); };
        // software pipeline over the 2 * TP MFMA groups (kk, b): the pixel fragment of group g+2 (and, once, the weight
        // fragments of the second 32-deep half) are read behind group g's TC MFMAs: two groups (2 * TC * 16 MFMA cycles) of
        // cover for the LDS latency
        n_u32x4 af[2][TC], bf[NGRP];
#pragma unroll
        for (int a = 0; a < TC; ++a) af[0][a] = lda(a, 0);
        bf[0] = ldb(0, 0);
        bf[1] = ldb(1 % TP, 1 / TP);
        static_for<NGRP>([&](auto G) {
            constexpr int g = decltype(G)::v, kk = g / TP, b = g % TP;
            if constexpr (g + 2 < NGRP) bf[g + 2] = ldb((g + 2) % TP, (g + 2) / TP);
            if constexpr (g == 0) {
#pragma unroll
                for (int a = 0; a < TC; ++a) af[1][a] = lda(a, 1);
            }
            if constexpr (dma && ILV) {
                static_for<NDMA>([&](auto J) {  // the DMA pieces whose slot is this group
                    if constexpr (decltype(J)::v * DGRP / NDMA == g) piece(J);
                });
            } else if constexpr (dma && g == 0) {
                static_for<NDMA>([&](auto J) { piece(J); });
            }
#pragma unroll
            for (int a = 0; a < TC; ++a) acc[a][b] = mfma_n16<F16>(af[kk][a], bf[g], acc[a][b]);
        });
        // Pin the issue order (hipcc otherwise sinks every fragment read to just before its first use, behind an
        // lgkmcnt(0), and the matrix pipe idles for the LDS latency plus the DMA issue of every group):
        // TC MFMAs of group g, THEN the read(s) for group g+1 and the group's DMA piece(s) -- they issue while the pipe
        // is still busy with the MFMAs just queued.
        __builtin_amdgcn_sched_group_barrier(0x100, TC + 2, 0);
        static_for<NGRP>([&](auto G) {
            constexpr int g = decltype(G)::v;
            __builtin_amdgcn_sched_group_barrier(0x008, TC, 0);
            constexpr int nread = (g + 2 < NGRP ? 1 : 0) + (g == 0 ? TC : 0);
            if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);
            if constexpr (dma) {
                constexpr int npiece = ILV ? (g < DGRP ? ((g + 1) * NDMA + DGRP - 1) / DGRP - (g * NDMA + DGRP - 1) / DGRP : 0)
                                           : (g == 0 ? NDMA : 0);
                if constexpr (npiece > 0) __builtin_amdgcn_sched_group_barrier(0x010, npiece, 0);
            }
        });
    };
    int cur = 0;
    for (int s = s_begin; s + 1 < s_end; ++s, cur ^= 1) {
        // every wave has seen its own pieces of step s land and (barrier) everyone else's; the barrier also closes the
        // fragment reads of step s-1, whose stage the next DMA overwrites
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        prep(s + 1, cur ^ 1);
        step(cur, IdxC<1>{});
    }
    if (s_begin < s_end) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        step(cur, IdxC<0>{});
    }

    // ---- epilogue: accumulators -> LDS (fp32, swizzled granules) -> compact coalesced loop, in NPASS passes ----
    constexpr int EPR = n16_epilogue_rows(BM, BN, WP);   // output rows per pass
    constexpr int NPASS = BM / EPR, TPP = TP / NPASS;    // pixel tiles per wave and pass
    static_assert(EPR % (16 * WP) == 0 && TPP * NPASS == TP && EPR * BN * 4 <= 2 * BUF, "epilogue pass geometry");
    constexpr int G = BN / 4;       // 16-byte granules per row
    constexpr int RPI = NT / G;     // rows per loop iteration
    static_assert(NT % G == 0 && G >= 16, "one thread per granule");
    float *Ct = reinterpret_cast<float *>(smem_n16);
    const int g = tid % G, r0 = tid / G;
    const int c = c0 + g * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    EpiCtx ec;
    epi_init(p, c, ec);
    static_for<NPASS>([&](auto E) {
        constexpr int e = decltype(E)::v;
        __syncthreads();  // the fragment reads of the last step / the previous pass's loop are done
        static_for<TPP>([&](auto BB) {
            constexpr int bb = decltype(BB)::v;
            const int ml = (bb * WP + wp) * 16 + l15;
            static_for<TC>([&](auto A) {
                constexpr int a = decltype(A)::v;
                const int gg = (wc * TC + a) * 4 + kg;
                *reinterpret_cast<n_f32x4 *>(Ct + ml * BN + ((gg ^ (ml & 15)) << 2)) = acc[a][e * TPP + bb];
            });
        });
        __syncthreads();
        int ho = 0, wo = 0;
        if (p.bias9) {
            const int mm = m0 + e * EPR + r0 < p.M ? m0 + e * EPR + r0 : 0;
            const int r = mm % (p.Ho * p.Wo);
            ho = r / p.Wo;
            wo = r - ho * p.Wo;
        }
        epi_dispatch(ec.mode, [&](auto MODE_) {
            for (int ml = r0; ml < EPR; ml += RPI) {
                const int m = m0 + e * EPR + ml;
                if (m >= p.M) break;
                const n_f32x4 q = *reinterpret_cast<const n_f32x4 *>(Ct + ml * BN + ((g ^ (ml & 15)) << 2));
                float v[4] = {q[0], q[1], q[2], q[3]};
                if (c < p.Cout) {
    #pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        s1[t] += v[t];
                        s2[t] += v[t] * v[t];
                    }
                    if (p.split_k > 1) {
                        float *dst = p.y + ((size_t)split * p.M + m) * p.Cout + c;
                        if (((p.Cout & 3) == 0) && c + 3 < p.Cout) {
                            *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
    #pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (c + t < p.Cout) dst[t] = v[t];
                        }
                    } else {
                        const int ry = ho == 0 ? 0 : (ho == p.Ho - 1 ? 2 : 1), rx = wo == 0 ? 0 : (wo == p.Wo - 1 ? 2 : 1);
                        epi_row<decltype(MODE_)::v>(p, ec, m, c, v, 3 * ry + rx);
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
        });
    });
    if (p.stats) {
        __syncthreads();  // Ct has been consumed
        float *red = reinterpret_cast<float *>(smem_n16);  // [RPI][2][BN]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            red[(r0 * 2 + 0) * BN + g * 4 + t] = s1[t];
            red[(r0 * 2 + 1) * BN + g * 4 + t] = s2[t];
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
}

// fp32 -> one 16-bit plane (optional per-channel affine first), 4 elements per thread; and back
__global__ void to_n16_kernel(const float4 *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
                              ushort4 *__restrict__ out, size_t n4, int C4, int narrow) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = x[i];
    if (scale) {
        const int c = (int)(i % C4) * 4;
        const float4 s = *reinterpret_cast<const float4 *>(scale + c), t = *reinterpret_cast<const float4 *>(shift + c);
        v = make_float4(v.x * s.x + t.x, v.y * s.y + t.y, v.z * s.z + t.z, v.w * s.w + t.w);
    }
    const float o[4] = {v.x, v.y, v.z, v.w};
    store_narrow4(reinterpret_cast<uint16_t *>(out + i), o, narrow);
}

__global__ void from_n16_kernel(const ushort4 *__restrict__ x, float4 *__restrict__ out, size_t n4, int narrow) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float o[4];
    load_narrow4(reinterpret_cast<const uint16_t *>(x + i), o, narrow);
    out[i] = make_float4(o[0], o[1], o[2], o[3]);
}

template <int BM, int BN, int WP, int WC, int ILV = 1>
static int launch_n16(const ConvArgs &a, hipStream_t st) {
    if ((long long)BN * a.Kpad * 2 >= (1ll << 31))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow): weight panel exceeds 31-bit offsets");
    const size_t lds = (size_t)2 * (BM + BN) * 128;
    const dim3 grid(a.tiles_m * a.tiles_n, 1, a.split_k), block(WP * WC * 64);
    if (a.narrow == CER_STORE_F16) {
        auto k = conv_n16_kernel<BM, BN, WP, WC, true, ILV>;
        if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a);
    } else {
        auto k = conv_n16_kernel<BM, BN, WP, WC, false, ILV>;
        if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

// tile ids (desc.tile): 0 = auto; flat kernels (K step 64): 63 = 256x64, 64 = 128x128, 65 = 128x64, 66 = 64x64, 67 = 64x128
// (4 waves each), 91 = 256x256 (8 waves) and 94 = 128x128 with the step's DMA pieces spread over the first half of the MFMA
// groups; window-resident kernels (conv_n16_patch.hip, 3x3 / stride 1 / pad 1): patches of 16x16 pixels (H, W % 16 == 0):
// 71 = 64 couts (Cin == 64, 4 waves, two blocks per CU), 72 = 128 couts, 78 = 128 couts with ping-pong phases; 1-D windows of
// 256 consecutive pixels (any image size with W <= 86): 73 = 64 couts, 76 = 128 couts with ping-pong phases, 77 = 64 couts,
// Cin == 64, ONE window buffer, 4 waves (two blocks per CU); 79 = persistent 16x16-patch blocks for Cin == 64 (conv_n16_p64.hip).
static bool patch_geometry(const cer_conv_desc *d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil_h == 1 && d->dil_w == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->Ho == d->H && d->Wo == d->W && (d->H & 15) == 0 && (d->W & 15) == 0 && (d->Cin & 63) == 0 && d->split_k <= 1;
}

static bool win_geometry(const cer_conv_desc *d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil_h == 1 && d->dil_w == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->Ho == d->H && d->Wo == d->W && d->H >= 2 && d->W >= 2 && d->W <= 86 && (d->Cin & 63) == 0 && d->split_k <= 1;
}

int conv_n16_tile_dims(const cer_conv_desc *d, int &bm, int &bn, int &bk) {
    int tile = d->tile;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    const int Cout = d->Cout;
    if (d->x_s2d) {   // space-to-depth input: only the window-resident stride-2 kernel reads it (conv_n16_s2d.hip)
        if (tile == 0) tile = Cout > 64 ? 82 : 81;
        else if (tile != 81 && tile != 82) return 0;
    } else if (tile == 81 || tile == 82) {
        return 0;
    }
    if (tile == 0) {
        const long long t256 = (M + 255) / 256;
        if (patch_geometry(d) && d->Cin == 64 && t256 * ((Cout + 63) / 64) >= 1024)
            tile = ((Cout & 63) == 0 && Cout <= 256 && Cout != 192 && t256 >= 2048) ? 79 : 71;   // 79: persistent blocks (conv_n16_p64.hip)
        else if (patch_geometry(d) && Cout >= 128 && t256 * ((Cout + 127) / 128) >= 512) tile = d->Cin >= 128 ? 78 : 72;  // 78: ping-pong
        else if (win_geometry(d) && t256 >= 64) {
            // one block per CU (the windows fill the LDS): whole rounds of 256 blocks; a 128-cout block does twice the work of
            // a 64-cout block in 1.68x the time (fp16, 1024 frames: 56x56 875 -> 957, 28x28 1066 -> 1158, 5x5 720 -> 939 TF/s)
            // measured per shape at 40x40, 80x80 and 224x224 input (tools/bench_conv.py): Cin == 64 -> the single-window tile (two
            // blocks per CU: 64 -> 64 @40x40 448 -> 565 TF/s, 64 -> 128 645 -> 677); else 128 couts with ping-pong phases, or 64
            tile = d->Cin == 64 ? 77 : (Cout > 64 ? 76 : 73);
        }
        else if (Cout <= 64) tile = t256 >= 512 ? 63 : ((M + 127) / 128 >= 256 ? 65 : 66);
        else if (Cout <= 128) tile = (M + 127) / 128 >= 256 ? 64 : 67;
        else if (t256 * ((Cout + 255) / 256) >= 512) tile = 91;
        else if ((M + 127) / 128 * ((Cout + 127) / 128) >= 256) tile = 94;
        else tile = 67;
    }
    bk = 64;
    switch (tile) {
        case 91: bm = 256; bn = 256; break;
        case 63: bm = 256; bn = 64; break;
        case 64: case 94: bm = 128; bn = 128; break;
        case 65: bm = 128; bn = 64; break;
        case 66: bm = 64; bn = 64; break;
        case 67: bm = 64; bn = 128; break;
        case 71: case 79: bm = 256; bn = 64; break;    // a 16x16 patch is 256 output pixels (79: persistent blocks)
        case 72: case 78: bm = 256; bn = 128; break;   // (78: ping-pong phases)
        case 73: case 77: bm = 256; bn = 64; break;    // 1-D window kernels (any image size): 256 consecutive pixels (77: Cin == 64, one window)
        case 76: bm = 256; bn = 128; break;   // (ping-pong phases)
        case 81: bm = 256; bn = 64; break;    // 3x3 / stride 2 on a space-to-depth input (conv_n16_s2d.hip)
        case 82: bm = 256; bn = 128; break;
        default: return 0;
    }
    return tile;
}

int conv_n16_launch(int tile, const ConvArgs &a, hipStream_t st) {
    switch (tile) {
        case 63: return launch_n16<256, 64, 4, 1>(a, st);
        case 64: return launch_n16<128, 128, 2, 2>(a, st);
        case 65: return launch_n16<128, 64, 2, 2>(a, st);
        case 66: return launch_n16<64, 64, 2, 2>(a, st);
        case 67: return launch_n16<64, 128, 1, 4>(a, st);
        case 71: case 72: case 73: case 76: case 77: case 78: return conv_n16_patch_launch(tile, a, st);
        case 79: return conv_n16_p64_launch(a, st);
        case 81: case 82: return conv_n16_s2d_launch(tile, a, st);
        case 91: return launch_n16<256, 256, 2, 4, 2>(a, st);
        case 94: return launch_n16<128, 128, 2, 2, 2>(a, st);
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (narrow): unknown tile id");
    }
}

}  // namespace cer

using namespace cer;

extern "C" int cer_conv2d_n16_tile(const cer_conv_desc *d) {
    if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return 0;
    int bm, bn, bk;
    return conv_n16_tile_dims(d, bm, bn, bk);
}

extern "C" int cer_to_n16(const float *x, const float *scale, const float *shift, int C, uint16_t *out, size_t n, int storage,
                          void *stream) {
    if (!x || !out || n == 0 || (n & 3)) return cer_set_error(CER_ERR_INVALID_ARG, "to_n16: n must be a positive multiple of 4");
    if (storage != CER_STORE_BF16 && storage != CER_STORE_F16) return cer_set_error(CER_ERR_INVALID_ARG, "to_n16: storage must be bf16 or f16");
    if ((scale == nullptr) != (shift == nullptr) || (scale && (C <= 0 || (C & 3) || n % C)))
        return cer_set_error(CER_ERR_INVALID_ARG, "to_n16: the per-channel affine needs scale, shift and C % 4 == 0 dividing n");
    CER_LAUNCH(to_n16_kernel, dim3(cer_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, scale, shift,
               (ushort4 *)out, n / 4, scale ? C / 4 : 1, storage);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_from_n16(const uint16_t *x, float *out, size_t n, int storage, void *stream) {
    if (!x || !out || n == 0 || (n & 3)) return cer_set_error(CER_ERR_INVALID_ARG, "from_n16: n must be a positive multiple of 4");
    if (storage != CER_STORE_BF16 && storage != CER_STORE_F16) return cer_set_error(CER_ERR_INVALID_ARG, "from_n16: storage must be bf16 or f16");
    CER_LAUNCH(from_n16_kernel, dim3(cer_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const ushort4 *)x, (float4 *)out,
               n / 4, storage);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
