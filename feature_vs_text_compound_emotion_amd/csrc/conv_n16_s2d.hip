// conv_n16_s2d.hip -- the narrow-storage (one bf16 / fp16 plane) twin of conv_b3_s2d.hip: the 3x3 / STRIDE 2 / pad 1 convolution
// window-resident over a SPACE-TO-DEPTH input.  Read conv_b3_s2d.hip's header for the phase decomposition (P11 | P10 | P01 | P00
// channel blocks, 4 / 2 / 2 / 1 shifts per phase = 9 C / 64 steps), the window schedule (which wave may issue a window when, and
// which wait makes it visible to the other ping-pong group) and the three small buffers of the halo-free P00 phase.
// Differences: one 16-bit plane with 128-byte rows (a chunk is 64 channels, two 32-deep MFMAs per 16x16 tile and step,
// v_mfma_f32_16x16x32_{bf16,f16}), 8-row one-KiB DMA pieces, the slot = chunk ^ (row & 6) swizzle of conv_n16_win_kernel.
// LDS: 98 KiB window region | 3 weight slices of BN x 128 B | 1 KiB sink.  Needs C % 128 == 0 (an even number of chunks per
// phase) and Wo <= 126.
#include "conv_n16.h"

namespace cer {

template <int BN, bool F16>
__global__ __launch_bounds__(512, 2) void conv_n16_s2d_kernel(ConvArgs p, int NPF) {
    constexpr int WP = 4, WC = 2, NW = 8, NT = 512, BM = 256;
    constexpr int WSLICE = BN * 128;
    static_assert(BN % (8 * NW) == 0 && (BN / 16) % 4 == 0, "each group DMAs its own cout half, dealt to its four waves");
    constexpr int WQ = BN / (8 * NW);
    constexpr int TP = BM / (16 * WP), TC = BN / (16 * WC), NGRP = 2 * TP;
    constexpr int XREG = 98 * 1024;                               // window region
    constexpr int WOFF = XREG, SINK = WOFF + 3 * WSLICE;
    constexpr int NP01 = 33, NP00 = 32;                           // 8-row pieces of a P01 / P00 window
    constexpr int XF = 6, X1 = 6, X0 = 8;                         // pieces per wave: full (<= 48), P01 (33: the sixth is a pad), P00 (group 0)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_n16s[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_n16s);
    const int FULLB = NPF * 1024;                                 // bytes of a full (P11 / P10) window buffer

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % WP, wc = wave / WP;
    const int kg = lane >> 4, l15 = lane & 15;

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware remap (bijective)
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, c0 = tile_n * BN;
    const int cpp = p.cin_steps;                                  // 64-channel chunks per phase
    const int S = 9 * cpp;                                        // steps
    const int W = p.Wo;

    // ---- DMA assignment: per-lane byte offsets from the first pixel of each kind of window ----
    const int prow = lane >> 3, slot = lane & 7;
    auto xoff = [&](int q, int np, int rows, long long wstart) -> unsigned {
        const int row = q * 8 + prow;
        const long long pix = wstart + row;
        const bool inb = q < np && row < rows && pix >= 0 && pix < (long long)p.M;
        return inb ? (unsigned)(((size_t)row * p.x_ld + ((slot ^ (row & 6)) << 3)) * 2) : OOB;
    };
    unsigned xoF[XF], xo1[X1], xo0[X0];
#pragma unroll
    for (int i = 0; i < XF; ++i) xoF[i] = xoff(wave + NW * i, NPF, BM + W + 1, (long long)m0 - W - 1);
#pragma unroll
    for (int i = 0; i < X1; ++i) xo1[i] = xoff(wave + NW * i, NP01, BM + 1, (long long)m0 - 1);
#pragma unroll
    for (int i = 0; i < X0; ++i) xo0[i] = wc == 0 ? xoff(wave + 4 * i, NP00, BM, (long long)m0) : OOB;
    unsigned w_off[WQ];
    int w_piece[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        w_piece[i] = (wave >> 2) * (BN / 16) + (wave & 3) + 4 * i;   // the group's own cout half (BN / 16 pieces of 8 rows)
        const int row = w_piece[i] * 8 + prow;
        w_off[i] = c0 + row < p.Cout ? (unsigned)(((size_t)row * p.Kpad + ((slot ^ ((row >> 1) & 7)) << 3)) * 2) : OOB;
    }
    // 64-bit bases of the three window kinds (never dereferenced outside the tensor: those lanes are OOB)
    const long long rowb = (long long)p.x_ld * 2;
    const char *xh = reinterpret_cast<const char *>(p.x_hi);
    const long long bF = ((long long)m0 - W - 1) * rowb, b1 = ((long long)m0 - 1) * rowb, b0 = (long long)m0 * rowb;
    const char *wpanel = reinterpret_cast<const char *>(p.w_hi) + (size_t)c0 * p.Kpad * 2;

    // one window piece of linear chunk `lin` (its channels start at lin * 64 of the 4C); out-of-range lanes write zeros
    auto dma1 = [&](long long base, int lin, unsigned char *dst, unsigned vo) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xh) + base + (size_t)lin * 128, 0, (int)OOB, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (n_lds_ptr_t)dst, 16, (int)vo, 0, 0, 0);
    };
    auto issue_full = [&](int i, int g) {        // piece i of this wave of the full window of linear chunk g (< 2 cpp)
        const int q = wave + NW * i;
        dma1(bF, g, q < NPF ? smem + (g & 1) * FULLB + q * 1024 : smem + SINK, q < NPF ? xoF[i] : OOB);
    };
    auto issue_p01 = [&](int i, int qc) {        // P01 chunk qc (local): low / high slot of the region by parity
        const int q = wave + NW * i;
        dma1(b1, 2 * cpp + qc, q < NP01 ? smem + (qc & 1) * 65536 + q * 1024 : smem + SINK, q < NP01 ? xo1[i] : OOB);
    };
    auto issue_p00 = [&](int i, int k, int sl) { // P00 chunk k (local) into small buffer sl: group 0 fetches, group 1 pads the count
        const bool real = wc == 0 && k < cpp;
        dma1(b0, 3 * cpp + (k < cpp ? k : 0), real ? smem + sl * 32768 + ((wave & 3) + 4 * i) * 1024 : smem + SINK, real ? xo0[i] : OOB);
    };
    auto issue_w = [&](int step, int ring) {
        const bool real = step < S;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wpanel) + (size_t)(real ? step : 0) * 128, 0, (int)OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            unsigned char *dst = real ? smem + WOFF + ring * WSLICE + w_piece[i] * 1024 : smem + SINK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (n_lds_ptr_t)dst, 16, (int)(real ? w_off[i] : OOB), 0, 0, 0);
        }
    };

    n_f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    // per pixel tile: the lane's row within the tile and which of the four shifts (kh, kw) in {0, 1}^2 stay inside its image
    int prow0[TP];
    unsigned vmask[TP];
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        const int pl = (b * WP + wp) * 16 + l15;
        prow0[b] = pl;
        const int m = m0 + pl;
        unsigned bits = 0;
        if (m < p.M) {
            const int r = m % (p.Ho * p.Wo);
            const int y = r / p.Wo, x = r - y * p.Wo;
            for (int kh = 0; kh < 2; ++kh)
                for (int kw = 0; kw < 2; ++kw)
                    if ((kh || y >= 1) && (kw || x >= 1)) bits |= 1u << (kh * 2 + kw);
        }
        vmask[b] = bits;
    }
    const int arow = (wc * TC * 16 + l15) * 128 + ((kg ^ ((l15 >> 1) & 7)) << 4);

#pragma unroll
    for (int i = 0; i < XF; ++i) issue_full(i, 0);
    issue_w(0, 0);
    issue_w(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WQ) : "memory");   // the window and slice 0 (slice 1 may still be in flight)
    __builtin_amdgcn_s_barrier();
    if (wc == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one phase behind group 0

    int s = 0, ring = 0, slot00 = 0;
    static_for<4>([&](auto PHc) {
        constexpr int PH = decltype(PHc)::v;                      // 0..3 = P11, P10, P01, P00
        constexpr int NTAP = PH == 0 ? 4 : (PH == 3 ? 1 : 2);
        for (int ccp = 0; ccp < cpp; ++ccp) {
            const bool last = ccp == cpp - 1;
            const int g = PH * cpp + ccp;
            int xcur, zrow;
            if constexpr (PH < 2) {
                xcur = (g & 1) * FULLB; zrow = NPF * 8 - 1;
            } else if constexpr (PH == 2) {
                xcur = (ccp & 1) * 65536; zrow = NP01 * 8 - 1;
            } else {
                xcur = slot00 * 32768; zrow = NP00 * 8 - 1;       // never selected for a pixel that is stored
            }
            static_for<NTAP>([&](auto Jc) {
                constexpr int J = decltype(Jc)::v;
                constexpr int kh = PH == 0 ? J / 2 : (PH == 1 ? J : 1);
                constexpr int kw = PH == 0 ? J % 2 : (PH == 2 ? J : 1);
                const int toff = PH < 2 ? kh * W + kw : (PH == 2 ? kw : 0);
                const unsigned char *Wr = smem + WOFF + ring * WSLICE;
                const unsigned char *Xb = smem + xcur;
                int baddr[TP];
#pragma unroll
                for (int b = 0; b < TP; ++b) {
                    const int row = ((vmask[b] >> (kh * 2 + kw)) & 1u) ? prow0[b] + toff : zrow;
                    baddr[b] = row * 128 + ((kg ^ (row & 6)) << 4);
                }
                auto lda = [&](int a, int kk) { return *reinterpret_cast<const n_u32x4 *>(Wr + ((arow + a * 16 * 128) ^ (kk << 6))); };
                auto ldb = [&](int b, int kk) { return *reinterpret_cast<const n_u32x4 *>(Xb + (baddr[b] ^ (kk << 6))); };
                // ---- READ phase ----
                n_u32x4 af[2][TC], bf[NGRP];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int a = 0; a < TC; ++a) af[kk][a] = lda(a, kk);
#pragma unroll
                for (int gq = 0; gq < NGRP; ++gq) bf[gq] = ldb(gq % TP, gq / TP);
                issue_w(s + 2, ring == 0 ? 2 : ring - 1);
                if constexpr (PH == 0) {
                    if constexpr (J == 0) { issue_full(0, g + 1); issue_full(1, g + 1); issue_full(2, g + 1); }
                    if constexpr (J == 1) { issue_full(3, g + 1); issue_full(4, g + 1); issue_full(5, g + 1); }
                } else if constexpr (PH == 1) {
                    if constexpr (J == 0) {
                        if (!last) {
#pragma unroll
                            for (int i = 0; i < XF; ++i) issue_full(i, g + 1);
                        } else {
#pragma unroll
                            for (int i = 0; i < X1; ++i) issue_p01(i, 0);
                        }
                    }
                } else if constexpr (PH == 2) {
                    if (!last) {
                        if constexpr (J == 0) {
#pragma unroll
                            for (int i = 0; i < X1; ++i) issue_p01(i, ccp + 1);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < X0; ++i) issue_p00(i, J, J);
                    }
                } else {
                    const int sl = slot00 == 0 ? 2 : slot00 - 1;   // (k + 2) % 3
#pragma unroll
                    for (int i = 0; i < X0; ++i) issue_p00(i, ccp + 2, sl);
                }
                // DMA instructions this wave issued in this READ phase (beyond the WQ of the weight slice)
                constexpr int NXA = PH == 0 ? (J < 2 ? 3 : 0) : (PH == 1 ? (J == 0 ? XF : 0) : (PH == 2 ? (J == 0 ? X1 : 0) : X0));
                constexpr int NXL = PH == 2 ? X0 : NXA;            // ... in the last chunk of the phase
                static_assert(XF == X1, "the last P10 chunk issues a P01 window in place of a full one: same instruction count");
                if constexpr ((PH == 1 || PH == 2) && J == 1) {
                    // the window issued in step 0 is read in the NEXT step by both groups: everything issued before this
                    // READ phase has to have landed at its end
                    if (PH == 2 && last) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WQ + NXL) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WQ + NXA) : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA phase ----
#pragma unroll
                for (int gq = 0; gq < NGRP; ++gq)
#pragma unroll
                    for (int a = 0; a < TC; ++a) acc[a][gq % TP] = mfma_n16<F16>(af[gq / TP][a], bf[gq], acc[a][gq % TP]);
                // everything this wave issued before this step's READ phase has landed
                __builtin_amdgcn_sched_barrier(0);
                if (PH == 2 && last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WQ + NXL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WQ + NXA) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                ++s;
                ring = ring == 2 ? 0 : ring + 1;
            });
            if constexpr (PH == 3) slot00 = slot00 == 2 ? 0 : slot00 + 1;
        }
    });
    if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue (conv_n16_win_kernel's): accumulators -> LDS (fp32, swizzled granules) -> coalesced loop over output rows ----
    constexpr int G = BN / 4, RPI = NT / G;
    static_assert(NT % G == 0, "one thread per granule");
    float *Ct = reinterpret_cast<float *>(smem_n16s);   // the launcher sizes the LDS for 256 * BN floats at least
    const int g = tid % G, r0 = tid / G;
    const int c = c0 + g * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    EpiCtx ec;
    epi_init(p, c, ec);
    __syncthreads();
    static_for<TP>([&](auto B) {
        constexpr int b = decltype(B)::v;
        const int ml = (b * WP + wp) * 16 + l15;
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            const int gg = (wc * TC + a) * 4 + kg;
            *reinterpret_cast<n_f32x4 *>(Ct + ml * BN + ((gg ^ (ml & 15)) << 2)) = acc[a][b];
        });
    });
    __syncthreads();
    epi_dispatch(ec.mode, [&](auto MODE_) {
        for (int ml = r0; ml < BM; ml += RPI) {
            const int m = m0 + ml;
            if (m >= p.M) break;
            const n_f32x4 q = *reinterpret_cast<const n_f32x4 *>(Ct + ml * BN + ((g ^ (ml & 15)) << 2));
            float v[4] = {q[0], q[1], q[2], q[3]};
            if (c < p.Cout) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s1[t] += v[t];
                    s2[t] += v[t] * v[t];
                }
                epi_row<decltype(MODE_)::v>(p, ec, m, c, v, 0);   // no bias9 on a strided conv
            }
        }
    });
    if (p.stats) {
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem_n16s);  // [RPI][2][BN]
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

// full-window pieces (8 rows), or 0 when two full buffers do not fit the 98 KiB window region
static int n16_s2d_pieces(const ConvArgs &a) {
    const int np = (256 + a.Wo + 1 + 1 + 7) / 8;     // + 1: the last row stays zero (masked shifts read it)
    return np <= 48 ? np : 0;
}

bool conv_n16_s2d_ok(const ConvArgs &a) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 2 || a.dil_h != 1 || a.dil_w != 1 || a.pad_t != 1 || a.pad_l != 1 || (a.H & 1) ||
        (a.W & 1) || a.Ho * 2 != a.H || a.Wo * 2 != a.W || (a.Cin & 127) || a.split_k != 1 || a.x_ld != 4 * a.Cin ||
        n16_s2d_pieces(a) == 0 || a.bias9)
        return false;
    return (long long)(512 + a.Wo) * a.x_ld * 2 < (1ll << 31) && (long long)128 * a.Kpad * 2 < (1ll << 31);
}

template <int BN>
static int launch_n16_s2d(const ConvArgs &a, hipStream_t st) {
    const int np = n16_s2d_pieces(a);
    size_t lds = (size_t)98 * 1024 + 3 * (size_t)BN * 128 + 1024;
    if (lds < (size_t)256 * BN * 4) lds = (size_t)256 * BN * 4;   // the epilogue's accumulator tile
    const dim3 grid(a.tiles_m * a.tiles_n, 1, 1), block(512);
    if (a.narrow == CER_STORE_F16) {
        auto k = conv_n16_s2d_kernel<BN, true>;
        CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a, np);
    } else {
        auto k = conv_n16_s2d_kernel<BN, false>;
        CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a, np);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

int conv_n16_s2d_launch(int tile, const ConvArgs &a, hipStream_t st) {
    if (!conv_n16_s2d_ok(a))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow, space-to-depth input): needs a 3x3 / stride 2 / pad 1 conv on even "
                                                   "H and W, Cin % 128 == 0, Wo <= 126, no split-K, no bias9");
    switch (tile) {
        case 81: return launch_n16_s2d<64>(a, st);
        case 82: return launch_n16_s2d<128>(a, st);
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (narrow, space-to-depth input): unknown tile id");
    }
}

}  // namespace cer
