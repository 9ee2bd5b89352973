// conv_b3_s2d.hip -- the bf16x3 3x3 / STRIDE 2 / pad 1 convolution (the second conv of the first unit of every IR-50 stage,
// reference models/arcface_model.py:38-41) as a WINDOW-RESIDENT kernel over a SPACE-TO-DEPTH input.
//
// The flat kernel gathers nine 128-pixel tap tiles per 32-channel chunk for a stride-2 conv (3.1x its algorithmic HBM traffic,
// 0.32 of the matrix ceiling: DESIGN.md section 4).  With the input stored space-to-depth,
//     xs[n][i][j][blk * C + c] = x[n][2i + py][2j + px][c],   blk = 3 - 2 py - px   (block order P11 | P10 | P01 | P00),
// every phase image has the OUTPUT's geometry [N, Ho, Wo], and the conv becomes a stride-1 2x2 "conv" over the 4C channels
// in which a phase only meets the taps that can reach it:
//     P11 (odd row, odd col):   shifts (-1,-1) (-1,0) (0,-1) (0,0)  <- filter taps (0,0) (0,2) (2,0) (2,2)
//     P10 (odd row, even col):  shifts (-1,0) (0,0)                 <- (0,1) (2,1)
//     P01 (even row, odd col):  shifts (0,-1) (0,0)                 <- (1,0) (1,2)
//     P00:                      shift  (0,0)                        <- (1,1)
// = 9 C / 32 steps, exactly the stride-2 conv's K.  The producer (the unit's first conv) writes the space-to-depth layout
// from its epilogue (cer_conv_desc.y_s2d: a row permutation of its stores), so no extra pass exists.
//
// A block owns 256 consecutive flattened output pixels x BN couts (as conv_b3_win_kernel).  Per 32-channel chunk of a phase
// the window [m0 - Wo - 1, m0 + 255] of the phase image is fetched once by LDS-DMA (P01 only needs [m0 - 1, ..], P00 only
// [m0, ..]) and serves the phase's 4 / 2 / 2 / 1 steps; the weights stream through the 3-slot ring in STEP ORDER (the host
// permutes the K columns once: ops.pack_s2d_weight).  Ping-pong phases, counted vmcnt, masks and the epilogue are the
// window kernel's.  What is new is the window schedule, because a chunk now lasts 1-4 steps instead of 9:
//   * a DMA issued by group g in its READ phase of step j may be read by group h in step j + D only if the wave waited
//     d <= 2 D + h - g - 1 phases later (the end-of-step wait is d = 3, a wait at the end of the NEXT step's READ phase d = 2);
//   * P11 chunks (4 steps) issue the next window in steps 0 and 1 (D >= 3); P10 / P01 chunks (2 steps) issue it in step 0 and
//     wait at the end of step 1's READ phase (D = 2, d = 2);
//   * P00 chunks last ONE step, so two alternating buffers cannot work (D = 1 admits no d for group 1's pieces).  P00 windows
//     have no halo (256 rows = 16 pieces), so the window region is re-cut into THREE 32 KiB buffers for that phase and a
//     window is issued two chunks ahead by group 0 (D = 2, g = 0: d = 3, the ordinary end-of-step wait); the last P01 chunk
//     sits at the top of the region and issues the first two P00 windows below itself.
// LDS: 98 KiB window region | 3 weight slices | 1 KiB sink = 147 KiB (BN = 128).  Needs C % 64 == 0 (an even number of chunks
// per phase keeps the buffer parities fixed) and Wo <= 126.
#include "conv_b3.h"

namespace cer {

template <int BN>
__global__ __launch_bounds__(512, 2) void conv_b3_s2d_kernel(ConvArgs p, int NPF) {
    constexpr int WP = 4, WC = 2, NW = 8, NT = 512, BM = 256;
    constexpr int WPL = BN * 64, WSLICE = 2 * WPL, WPIECES = 2 * BN / 16;
    static_assert(WPIECES % NW == 0 && (BN / 16) % 4 == 0, "each group DMAs its own cout half, dealt to its four waves");
    constexpr int WQ = WPIECES / NW;
    constexpr int TP = BM / (16 * WP), TC = BN / (16 * WC);
    constexpr int XREG = 98 * 1024;                               // window region
    constexpr int WOFF = XREG, SINK = WOFF + 3 * WSLICE;
    constexpr int NP01 = 17, NP00 = 16;                           // pieces per plane of a P01 / P00 window
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_b3s[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_b3s);
    const int FULLB = 2 * NPF * 1024;                             // bytes of a full (P11 / P10) window buffer: two planes

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
    const int cpp = p.cin_steps;                                  // 32-channel chunks per phase
    const int S = 9 * cpp;                                        // steps
    const int W = p.Wo;

    // ---- DMA assignment: per-lane byte offsets from the first pixel of each kind of window ----
    const int prow = lane >> 2, slot = lane & 3;
    auto xoff = [&](int q, int np, int rows, long long wstart) -> unsigned {
        const int row = q * 16 + prow;
        const long long pix = wstart + row;
        const bool inb = q < np && row < rows && pix >= 0 && pix < (long long)p.M;
        return inb ? (unsigned)(((size_t)row * p.x_ld + ((slot ^ ((row & 4) >> 1)) << 3)) * 2) : OOB;
    };
    unsigned xoF[3], xo1[3], xo0[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xoF[i] = xoff(wave + NW * i, NPF, BM + W + 1, (long long)m0 - W - 1);
        xo1[i] = xoff(wave + NW * i, NP01, BM + 1, (long long)m0 - 1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) xo0[i] = wc == 0 ? xoff(wave + 4 * i, NP00, BM, (long long)m0) : OOB;
    unsigned w_off[WQ];
    int w_plane[WQ], w_dst[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        const int j = (wave & 3) + 4 * i;
        constexpr int PPL = BN / 32;                              // pieces per plane in this group's pool
        w_plane[i] = j / PPL;
        const int pc = (wave >> 2) * PPL + j % PPL;
        const int row = pc * 16 + prow;
        w_dst[i] = w_plane[i] * WPL + pc * 1024;
        w_off[i] = c0 + row < p.Cout ? (unsigned)(((size_t)row * p.Kpad + ((slot ^ swz16((row >> 2) & 3)) << 3)) * 2) : OOB;
    }
    // 64-bit bases of the three window kinds (never dereferenced outside the tensor: those lanes are OOB)
    const long long rowb = (long long)p.x_ld * 2;
    const char *xh = reinterpret_cast<const char *>(p.x_hi), *xl = reinterpret_cast<const char *>(p.x_lo);
    const long long bF = ((long long)m0 - W - 1) * rowb, b1 = ((long long)m0 - 1) * rowb, b0 = (long long)m0 * rowb;
    const size_t wpan = (size_t)c0 * p.Kpad * 2;
    const char *wh = reinterpret_cast<const char *>(p.w_hi) + wpan, *wl = reinterpret_cast<const char *>(p.w_lo) + wpan;

    // one window piece (both planes) of linear chunk `lin` (its channels start at lin * 32 of the 4C)
    auto dma2 = [&](long long base, int lin, unsigned char *dst, int xpl, unsigned vo) {
        const bool real = vo != OOB;
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xh) + base + (size_t)lin * 64, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xl) + base + (size_t)lin * 64, 0, (int)OOB, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)dst, 16, (int)vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_ptr_t)(dst + xpl), 16, (int)vo, 0, 0, 0);
        (void)real;
    };
    // NOTE on out-of-range lanes: the DMA writes zeros for them (that is what fills the halo outside the tensor and the zero
    // row); a piece that does not exist (q >= np) goes to the sink.
    auto issue_full = [&](int i, int g) {        // piece i of this wave of the full window of linear chunk g (< 2 cpp)
        const int q = wave + NW * i;
        unsigned char *dst = q < NPF ? smem + (g & 1) * FULLB + q * 1024 : smem + SINK;
        dma2(bF, g, dst, q < NPF ? NPF * 1024 : 0, q < NPF ? xoF[i] : OOB);
    };
    auto issue_p01 = [&](int i, int qc) {        // P01 chunk qc (local): low / high slot of the region by parity
        const int q = wave + NW * i;
        unsigned char *dst = q < NP01 ? smem + (qc & 1) * 65536 + q * 1024 : smem + SINK;
        dma2(b1, 2 * cpp + qc, dst, q < NP01 ? NP01 * 1024 : 0, q < NP01 ? xo1[i] : OOB);
    };
    auto issue_p00 = [&](int i, int k, int sl) { // P00 chunk k (local) into small buffer sl: group 0 fetches, group 1 pads the count
        const bool real = wc == 0 && k < cpp;
        unsigned char *dst = real ? smem + sl * 32768 + ((wave & 3) + 4 * i) * 1024 : smem + SINK;
        dma2(b0, 3 * cpp + (k < cpp ? k : 0), dst, real ? NP00 * 1024 : 0, real ? xo0[i] : OOB);
    };
    auto issue_w = [&](int step, int ring) {
        const bool real = step < S;
        const size_t koff = (size_t)(real ? step : 0) * 64;
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            const char *base = (w_plane[i] ? wl : wh) + koff;
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (int)OOB, 0x00020000);
            unsigned char *dst = real ? smem + WOFF + ring * WSLICE + w_dst[i] : smem + SINK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, (int)(real ? w_off[i] : OOB), 0, 0, 0);
        }
    };

    f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    // per pixel tile: the lane's row within the tile and which of the four shifts (kh, kw) in {0, 1}^2 stay inside its image
    // (kh = 0 reads the phase row above: needs ho >= 1; kw = 0 the phase column to the left: needs wo >= 1)
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
    const int arow = (wc * TC * 16 + l15) * 64 + ((kg ^ swz16((l15 >> 2) & 3)) << 4);

    issue_full(0, 0);
    issue_full(1, 0);
    issue_full(2, 0);
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
            int xcur, xpl, zrow;
            if constexpr (PH < 2) {
                xcur = (g & 1) * FULLB; xpl = NPF * 1024; zrow = NPF * 16 - 1;
            } else if constexpr (PH == 2) {
                xcur = (ccp & 1) * 65536; xpl = NP01 * 1024; zrow = NP01 * 16 - 1;
            } else {
                xcur = slot00 * 32768; xpl = NP00 * 1024; zrow = NP00 * 16 - 1;   // never selected for a pixel that is stored
            }
            static_for<NTAP>([&](auto Jc) {
                constexpr int J = decltype(Jc)::v;
                constexpr int kh = PH == 0 ? J / 2 : (PH == 1 ? J : 1);
                constexpr int kw = PH == 0 ? J % 2 : (PH == 2 ? J : 1);
                const int toff = PH < 2 ? kh * W + kw : (PH == 2 ? kw : 0);
                const unsigned char *Wr = smem + WOFF + ring * WSLICE;
                const unsigned char *Xb = smem + xcur;
                auto lda = [&](int a, int pl) { return *reinterpret_cast<const u32x4 *>(Wr + pl * WPL + arow + a * 16 * 64); };
                int baddr[TP];
#pragma unroll
                for (int b = 0; b < TP; ++b) {
                    const int row = ((vmask[b] >> (kh * 2 + kw)) & 1u) ? prow0[b] + toff : zrow;
                    baddr[b] = row * 64 + ((kg ^ ((row & 4) >> 1)) << 4);
                }
                auto ldb = [&](int b, int pl) { return *reinterpret_cast<const u32x4 *>(Xb + pl * xpl + baddr[b]); };
                // ---- READ phase ----
                u32x4 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
                for (int a = 0; a < TC; ++a) { ah[a] = lda(a, 0); al[a] = lda(a, 1); }
#pragma unroll
                for (int b = 0; b < TP; ++b) { bh[b] = ldb(b, 0); bl[b] = ldb(b, 1); }
                issue_w(s + 2, ring == 0 ? 2 : ring - 1);
                if constexpr (PH == 0) {
                    if constexpr (J == 0) { issue_full(0, g + 1); issue_full(1, g + 1); }
                    if constexpr (J == 1) issue_full(2, g + 1);
                } else if constexpr (PH == 1) {
                    if constexpr (J == 0) {
                        if (!last) { issue_full(0, g + 1); issue_full(1, g + 1); issue_full(2, g + 1); }
                        else { issue_p01(0, 0); issue_p01(1, 0); issue_p01(2, 0); }
                    }
                } else if constexpr (PH == 2) {
                    if (!last) {
                        if constexpr (J == 0) { issue_p01(0, ccp + 1); issue_p01(1, ccp + 1); issue_p01(2, ccp + 1); }
                    } else {
                        issue_p00(0, J, J); issue_p00(1, J, J); issue_p00(2, J, J); issue_p00(3, J, J);
                    }
                } else {
                    const int sl = slot00 == 0 ? 2 : slot00 - 1;   // (k + 2) % 3
                    issue_p00(0, ccp + 2, sl); issue_p00(1, ccp + 2, sl); issue_p00(2, ccp + 2, sl); issue_p00(3, ccp + 2, sl);
                }
                // DMA instructions this wave issued in this READ phase (beyond the WQ of the weight slice)
                constexpr int NXA = PH == 0 ? (J == 0 ? 4 : (J == 1 ? 2 : 0)) : (PH == 1 ? (J == 0 ? 6 : 0) : (PH == 2 ? (J == 0 ? 6 : 0) : 8));
                constexpr int NXL = PH == 2 ? 8 : NXA;             // ... in the last chunk of the phase
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
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int gq = 0; gq < TP; ++gq)
                        acc[a][gq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al[a]), as_bf16x8(bh[gq]), acc[a][gq], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int gq = 0; gq < TP; ++gq)
                        acc[a][gq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bl[gq]), acc[a][gq], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int gq = 0; gq < TP; ++gq)
                        acc[a][gq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bh[gq]), acc[a][gq], 0, 0, 0);
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

    // ---- epilogue (conv_b3_win_kernel's): accumulators -> LDS (fp32, swizzled granules) -> coalesced loop over output rows ----
    constexpr int G = BN / 4, RPI = NT / G;
    static_assert(NT % G == 0, "one thread per granule");
    float *Ct = reinterpret_cast<float *>(smem_b3s);   // the launcher sizes the LDS for 256 * BN floats at least
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
            *reinterpret_cast<f32x4 *>(Ct + ml * BN + ((gg ^ (ml & 15)) << 2)) = acc[a][b];
        });
    });
    __syncthreads();
    epi_dispatch(ec.mode, [&](auto MODE_) {
        for (int ml = r0; ml < BM; ml += RPI) {
            const int m = m0 + ml;
            if (m >= p.M) break;
            const f32x4 q = *reinterpret_cast<const f32x4 *>(Ct + ml * BN + ((g ^ (ml & 15)) << 2));
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
        float *red = reinterpret_cast<float *>(smem_b3s);  // [RPI][2][BN]
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

// full-window pieces per plane, or 0 when two full buffers do not fit the 98 KiB window region
static int b3_s2d_pieces(const ConvArgs &a) {
    const int np = (256 + a.Wo + 1 + 1 + 15) / 16;   // + 1: the last row stays zero (masked shifts read it)
    return np <= 24 ? np : 0;
}

bool conv_b3_s2d_ok(const ConvArgs &a) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 2 || a.dil_h != 1 || a.dil_w != 1 || a.pad_t != 1 || a.pad_l != 1 || (a.H & 1) ||
        (a.W & 1) || a.Ho * 2 != a.H || a.Wo * 2 != a.W || (a.Cin & 63) || a.split_k != 1 || a.x_ld != 4 * a.Cin ||
        b3_s2d_pieces(a) == 0 || a.bias9)
        return false;
    return (long long)(512 + a.Wo) * a.x_ld * 2 < (1ll << 31) && (long long)128 * a.Kpad * 2 < (1ll << 31);
}

template <int BN>
static int launch_b3_s2d(const ConvArgs &a, hipStream_t st) {
    const int np = b3_s2d_pieces(a);
    size_t lds = (size_t)98 * 1024 + 3 * (size_t)2 * BN * 64 + 1024;
    if (lds < (size_t)256 * BN * 4) lds = (size_t)256 * BN * 4;   // the epilogue's accumulator tile
    auto k = conv_b3_s2d_kernel<BN>;
    CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CER_LAUNCH(k, dim3(a.tiles_m * a.tiles_n, 1, 1), dim3(512), lds, st, a, np);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

int conv_b3_s2d_launch(int tile, const ConvArgs &a, hipStream_t st) {
    if (!conv_b3_s2d_ok(a))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3, space-to-depth input): needs a 3x3 / stride 2 / pad 1 conv on even "
                                                   "H and W, Cin % 64 == 0, Wo <= 126, no split-K, no bias9");
    switch (tile) {
        case 51: return launch_b3_s2d<64>(a, st);
        case 52: return launch_b3_s2d<128>(a, st);
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (bf16x3, space-to-depth input): unknown tile id");
    }
}

}  // namespace cer

// The K order the space-to-depth kernels consume their weight slices in: phase-major (P11, P10, P01, P00), then the channel
// chunk (32 channels: bf16x3, 64: narrow), then the phase's shifts; each step is `chunk` consecutive channels of one filter tap.
extern "C" int cer_conv_s2d_k_order(int Cin, int chunk, int32_t *order) {
    if (!order || Cin <= 0 || (chunk != 32 && chunk != 64) || Cin % (2 * chunk))
        return cer_set_error(CER_ERR_INVALID_ARG, "conv_s2d_k_order: chunk is 32 or 64 and Cin a positive multiple of 2 * chunk");
    static const int taps[4][4] = {{0, 2, 6, 8}, {1, 7, -1, -1}, {3, 5, -1, -1}, {4, -1, -1, -1}};
    static const int ntap[4] = {4, 2, 2, 1};
    int j = 0;
    for (int ph = 0; ph < 4; ++ph)
        for (int cc = 0; cc < Cin / chunk; ++cc)
            for (int t = 0; t < ntap[ph]; ++t)
                for (int c = 0; c < chunk; ++c) order[j++] = taps[ph][t] * Cin + cc * chunk + c;
    return CER_OK;
}
