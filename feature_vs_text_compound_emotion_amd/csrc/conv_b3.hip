// conv_b3.hip -- implicit-GEMM convolution with ~2^-16-relative-per-product accuracy on the BF16 matrix cores.
//
// Every operand is carried as a SPLIT value: hi = bf16(v), lo = bf16(v - hi) (two bf16 planes, the
// same 4 bytes per element as fp32).  A product a*b is evaluated as
//        a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (the dropped a_lo*b_lo term is ~2^-18 relative)
// with three bf16 MFMAs into one fp32 accumulator: a 5.3x higher matrix ceiling than the fp32 MFMA path
// (2.5 PFLOP/s / 3 = 833 TFLOP/s effective) while the measured end-to-end logit error of the whole
// IR-50 + LFAN stack stays at 1.3e-6 (plain bf16: 8e-4), DESIGN.md section 4.
//
// One kernel family is shipped: conv_b3_dma16_kernel (LDS-DMA staging, v_mfma_f32_16x16x32_bf16, compact LDS
// epilogue).  The register-staged, ping-pong, 32x32x16-MFMA, window-resident and 3-stage variants that were measured
// against it in round 1 (all slower or level) were removed in round 2; they are in the history.
#include "conv_b3.h"

namespace cer {

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
        EpiCtx ec;
        epi_init(p, c, ec);
        // row / column of the thread's first pixel once (bias9), then stepped without divisions
        int ho = 0, wo = 0;
        if (p.bias9) {
            const int mm = m0 + r0 < p.M ? m0 + r0 : 0;
            const int r = mm % (p.Ho * p.Wo);
            ho = r / p.Wo;
            wo = r - ho * p.Wo;
        }
        epi_dispatch(ec.mode, [&](auto MODE_) {
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

// tile ids (desc.tile): 0 auto; 41/42/44/45 = 128x128 / 128x64 / 64x128 / 64x64 (4 waves, two LDS stages);
// 48 = 256x64 (4 waves x (64 pixels x 64 couts)) for Cout <= 64 at large M; window-resident kernels (conv_b3_patch.hip, 3x3 /
// stride 1 / pad 1): 16x16 patches (H, W % 16 == 0): 58 = 128 couts (8 waves, ping-pong phases), 59 = 64 couts on 4 waves with ONE
// window buffer re-filled per chunk (two blocks per CU); 1-D windows of 256 consecutive
// pixels x 64 / 128 couts (any image size with W <= 86).
static bool b3_patch_geometry(const cer_conv_desc *d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil_h == 1 && d->dil_w == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->Ho == d->H && d->Wo == d->W && (d->H & 15) == 0 && (d->W & 15) == 0 && (d->Cin & 31) == 0 && d->split_k <= 1;
}

static bool b3_win_geometry(const cer_conv_desc *d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil_h == 1 && d->dil_w == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->Ho == d->H && d->Wo == d->W && d->H >= 2 && d->W >= 2 && d->W <= 86 && (d->Cin & 31) == 0 && d->split_k <= 1;
}

int conv_b3_tile_dims(const cer_conv_desc *d, int &bm, int &bn, int &bk) {
    int tile = d->tile;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    const int Cout = d->Cout;
    if (d->x_s2d) {   // space-to-depth input: only the window-resident stride-2 kernel reads it (conv_b3_s2d.hip)
        if (tile == 0) tile = Cout > 64 ? 52 : 51;
        else if (tile != 51 && tile != 52) return 0;
    } else if (tile == 51 || tile == 52) {
        return 0;
    }
    if (tile == 0) {
        // measured per layer shape on MI355X (tools/bench_conv.py)
        const long long t128 = (M + 127) / 128 * ((Cout + 127) / 128), t256 = (M + 255) / 256;
        if (b3_patch_geometry(d) && Cout <= 64 && t256 >= 512) tile = 59;   // 4 waves, one window, two blocks per CU: 64 -> 64 @224x224 300 -> 341 TF/s
        else if (b3_patch_geometry(d) && Cout >= 128 && t256 * ((Cout + 127) / 128) >= 512) tile = 58;
        else if (b3_win_geometry(d) && t256 >= 64) tile = Cout > 64 ? 56 : 53;   // ping-pong window kernels: measured per shape at
                                                                                 // 40x40, 80x80 and 224x224 input (tools/bench_conv.py)
        else if (Cout <= 64) tile = t256 >= 512 ? 48 : 42;  // 256x64: 226 vs 204 TF/s on 64->64 @224x224
        else if (t128 >= 512) tile = 41;
        else tile = Cout >= 128 ? 44 : 45;
    }
    bk = 32;
    switch (tile) {
        case 41: bm = 128; bn = 128; break;
        case 42: bm = 128; bn = 64; break;
        case 44: bm = 64; bn = 128; break;
        case 45: bm = 64; bn = 64; break;
        case 48: bm = 256; bn = 64; break;
        case 59: bm = 256; bn = 64; break;    // a 16x16 patch is 256 output pixels (4 waves, one window, two blocks per CU)
        case 58: bm = 256; bn = 128; break;
        case 53: bm = 256; bn = 64; break;    // 1-D window kernels (any image size): 256 consecutive pixels (53: 4 waves, one window)
        case 56: bm = 256; bn = 128; break;
        case 51: bm = 256; bn = 64; break;    // 3x3 / stride 2 on a space-to-depth input (conv_b3_s2d.hip)
        case 52: bm = 256; bn = 128; break;
        default: return 0;
    }
    return tile;
}

int conv_b3_launch(int tile, const ConvArgs &a, hipStream_t st) {
    switch (tile) {
        case 41: return launch_b3_dma16<128, 128, 2, 2>(a, st);
        case 42: return launch_b3_dma16<128, 64, 2, 2>(a, st);
        case 44: return launch_b3_dma16<64, 128, 1, 4>(a, st);
        case 45: return launch_b3_dma16<64, 64, 2, 2>(a, st);
        case 48: return launch_b3_dma16<256, 64, 4, 1>(a, st);
        case 53: case 56: case 58: case 59: return conv_b3_patch_launch(tile, a, st);
        case 51: case 52: return conv_b3_s2d_launch(tile, a, st);
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (bf16x3): unknown tile id");
    }
}

}  // namespace cer

using namespace cer;

extern "C" int cer_conv2d_b3_tile(const cer_conv_desc *d) {
    if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return 0;
    int bm, bn, bk;
    return conv_b3_tile_dims(d, bm, bn, bk);
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
