// conv_n16_patch.hip -- 3x3 / stride 1 / pad 1 convolution on narrow operands with the INPUT WINDOW RESIDENT IN LDS.
//
// Why.  PMC of the flat implicit-GEMM kernel (conv_n16.hip; profiles/round2_pmc_conv_n16.txt): the matrix pipes are busy a
// third of the time and 41 % of the wave cycles are parked at the vmcnt/barrier.  A flat M tile re-fetches its activation
// rows once per filter tap -- nine L2 -> LDS transfers of every input line, a fifth of which miss the XCD's 4 MiB L2
// because the kh-shifted re-reads come three steps (several MiB of other traffic) later; a 1-KiB DMA piece spans 8
// lines, so most pieces wait for a miss.  The kernel below fetches each input line ONCE per block:
//
//   * a block owns a 16 x 16 patch of output pixels of one image and BN output channels;
//   * per 64-channel chunk the 18 x 18 input window of the patch (halo included; out-of-image pixels come back as zeros
//     from the range-checked DMA) is brought into LDS by 41 one-KiB LDS-DMA pieces; all nine taps then read their pixel
//     fragments from it: the fragment of output row r under tap (kh, kw) is the 16 consecutive window pixels
//     (r + kh) * 18 + kw + (lane & 15).  Those rows start at any alignment, so the bank-conflict swizzle is a function of
//     the window COLUMN found by exhaustive search (tools/check_swizzle.py: conflict free for kw = 0, 1, 2);
//   * only the weights stream per step: one BN x 64 slice per (chunk, tap) through a 3-slot ring, two steps ahead with
//     a counted vmcnt; every block of a launch reads the same slices, so they are L2 hits with short latency, while the
//     long-latency window of the NEXT chunk is fetched a whole chunk (nine steps) ahead into the second window buffer;
//   * the nine taps are unrolled (tap, ring slot and the next slice's tap are compile-time), fragment reads run two
//     MFMA groups ahead and the DMA pieces are placed between MFMA groups, as in conv_n16_kernel.
//
// L2 -> LDS bytes per 256 pixels x 64 channels x 9 taps: 41 KiB of window + 9 weight slices, against 9 x 32 KiB of
// activations + the same 9 slices for the flat 256-row tile.
#include "conv_n16.h"

namespace cer {


template <int BN, int WP, int WC, int XBUFS, bool F16, bool PP>
__global__ __launch_bounds__(WP * WC * 64, 2) void conv_n16_patch_kernel(ConvArgs p, PatchGeo geo) {
    constexpr int NW = WP * WC, NT = NW * 64;
    constexpr int PH = 16, PWD = 16, WW = 18, WROWS = 18 * 18;    // patch and window geometry
    constexpr int XPIECES = (WROWS + 7) / 8;                       // 41 one-KiB pieces (328 rows, the last 4 unused)
    constexpr int XPW = (XPIECES + NW - 1) / NW;                   // window pieces per wave and chunk
    constexpr int XBYTES = XPIECES * 1024;
    constexpr int WSLICE = BN * 128, RING = 3;
    static_assert(BN % (8 * NW) == 0, "weight-slice pieces are dealt round-robin to the waves");
    constexpr int WQ = BN / (8 * NW);                               // weight pieces per wave and step
    constexpr int TP = PH / WP, TC = BN / (16 * WC);                // 16x16 MFMA tiles per wave: output rows x cout tiles
    static_assert(PH % WP == 0 && XBUFS >= 1 && XBUFS <= 2 && (XBUFS == 1 || XPW <= 9) && TC % 2 == 0, "geometry");
    static_assert(!PP || (WP == 4 && WC == 2 && XBUFS == 2 && (BN / 16) % 4 == 0), "ping-pong: waves w and w + 4 share a SIMD and split the couts");
    constexpr int WOFF = XBUFS * XBYTES, SINK = WOFF + RING * WSLICE;  // LDS map: windows | weight ring | 1 KiB sink
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NGRP = 2 * TP;                                    // MFMA groups (kk, b) per step, TC MFMAs each
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_n16p[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_n16p);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % WP, wc = wave / WP;
    const int kg = lane >> 4, l15 = lane & 15;

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware remap (bijective): blocks that share an XCD's L2 take neighbouring patches
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int patch = (int)fdiv((unsigned)bid, geo.tiles_n), tile_n = bid - patch * p.tiles_n;
    const int prw = (int)fdiv((unsigned)patch, geo.pxn), px = patch - prw * (int)geo.pxn.d;
    const int n = (int)fdiv((unsigned)prw, geo.pyn), py = prw - n * (int)geo.pyn.d;
    const int c0 = tile_n * BN;
    const int cin_steps = p.cin_steps;

    // ---- DMA assignment ----  (stamps of the 64 -> 64 @224x224 launch, tools/exp_stamp.py: the block used to spend 2.9 us of its
    // 13.7 on ~570 instructions of 64-bit address arithmetic and emulated divisions before its first load; everything below is
    // 32-bit -- conv_n16_patch_ok bounds an image and a weight panel by 2^31 bytes -- and the weight slices, whose addresses
    // need nothing of the patch, are issued before the window's addresses are computed)
    const int prow = lane >> 3, slot = lane & 7;
    unsigned w_off[WQ];
    int w_piece[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        // PP: the group's own cout half (BN / 16 pieces of 8 rows) dealt to its four waves; else all pieces over all waves
        w_piece[i] = PP ? (wave >> 2) * (BN / 16) + (wave & 3) + 4 * i : wave + NW * i;
        const int row = w_piece[i] * 8 + prow;                                          // LDS row of the slice
        const int grow = row / (TC * 16) * (TC * 16) + epi_cout_of_row(row % (TC * 16));  // the cout it holds (conv_common.h)
        w_off[i] = c0 + grow < p.Cout ? (unsigned)(grow * p.Kpad * 2) + (unsigned)((slot ^ ((row >> 1) & 7)) << 4) : OOB;
    }
    const char *wpanel = reinterpret_cast<const char *>(p.w_hi) + (size_t)c0 * p.Kpad * 2;
    // weight slice (cc, tap) into ring slot `ring` (zeros into the sink past the end of the K loop)
    auto issue_w = [&](int cc, int tap, int ring) {
        const bool real = cc < cin_steps;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(wpanel) + ((size_t)tap * p.Cin + (size_t)cc * 64) * 2, 0, (int)OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            unsigned char *dst = real ? smem + WOFF + ring * WSLICE + w_piece[i] * 1024 : smem + SINK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (n_lds_ptr_t)dst, 16, (int)(real ? w_off[i] : OOB), 0, 0, 0);
        }
    };
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);

    // window: wave w moves pieces w, w + NW, ...; in a piece lane l owns window row 8 * piece + l / 8 = (wy, wx), LDS slot
    // l % 8, and fetches source chunk slot ^ PATCH_F[wx]; offsets are relative to the image's first pixel
    unsigned x_off[XPW];
    bool x_real[XPW];
    {
        const int iy0 = py * PH - 1, ix0 = px * PWD - 1, pitch = p.x_ld * 2;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int q = wave + NW * i;
            const int row = q * 8 + prow;
            const int wy = row / WW, wx = row - wy * WW;
            const int iy = iy0 + wy, ix = ix0 + wx;
            const bool inb = q < XPIECES && row < WROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = (unsigned)((iy * p.W + ix) * pitch) + (unsigned)((slot ^ patch_f(wx)) << 4);
            x_real[i] = q < XPIECES;
            x_off[i] = inb ? off : OOB;
        }
    }
    const char *ximg = reinterpret_cast<const char *>(p.x_hi) + (size_t)n * p.H * p.W * p.x_ld * 2;
    // window piece i of chunk cc into window buffer cc % XBUFS (or a zero piece into the sink: keeps the per-step DMA
    // count of a wave constant, which the counted vmcnt relies on)
    auto issue_x = [&](int i, int cc) {
        const bool real = x_real[i] && cc < cin_steps;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(ximg) + (size_t)cc * 128, 0, (int)OOB, 0x00020000);
        unsigned char *dst = real ? smem + (XBUFS == 2 ? (cc & 1) * XBYTES : 0) + (wave + NW * i) * 1024 : smem + SINK;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (n_lds_ptr_t)dst, 16, (int)(real ? x_off[i] : OOB), 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < XPW; ++i) issue_x(i, 0);

    n_f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    // ---- fragment address bases ----
    // weights: row = cout, slot = chunk ^ ((row >> 1) & 7); pixels: window row (r + kh) * 18 + kw + l15 with output row
    // r = b * WP + wp, slot = chunk ^ PATCH_F[kw + l15]  (second 32-deep half: ^ 64 bytes)
    const int arow = WOFF + (wc * TC * 16 + l15) * 128 + ((kg ^ ((l15 >> 1) & 7)) << 4);
    int bcol[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) bcol[kw] = (wp * WW + kw + l15) * 128 + ((kg ^ patch_f(kw + l15)) << 4);

    // ---- prologue: the weight slices of steps 0 and 1 and the whole window of chunk 0 are in flight (issued above) ----
    if constexpr (PP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // slices 0 / 1 and the window
        __builtin_amdgcn_s_barrier();
        if (wc == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one phase behind group 0
    }

    // One step = one filter tap of one 64-channel chunk.  DMA issued per wave and step: WQ weight pieces (slice of step
    // s + 2) and, with two window buffers, one window piece of the next chunk during taps 0 .. XPW-1: CNT(tap) pieces.
    // At the top of step s every piece issued before step s-1 must have landed (slice s was issued in step s-2, the
    // window of this chunk during the previous chunk / the prologue): vmcnt(CNT(previous tap)).
    for (int cc = 0; cc < cin_steps; ++cc) {
        const int xcur = XBUFS == 2 ? (cc & 1) * XBYTES : 0;
        static_for<9>([&](auto T) {
            constexpr int tap = decltype(T)::v, kh = tap / 3, kw = tap % 3;
            constexpr bool XWIN2 = XBUFS == 2;
            if constexpr (!PP) {
                constexpr int ptap = (tap + 8) % 9;                                  // the previous step's tap
                constexpr int pcnt = WQ + ((XBUFS == 2 && ptap < XPW) ? 1 : 0);      // pieces the previous step issued
                // (lgkmcnt(0): this wave's fragment reads of the previous step have returned before anyone's DMA may
                // overwrite the slot / window they came from)
                if (cc == 0 && tap == 0) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            // prologue: the slices and the window
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(pcnt) : "memory");
                }
                __builtin_amdgcn_s_barrier();
            }
            constexpr int ntap = (tap + 2) % 9, nring = (tap + 2) % 3;
            const int ncc = cc + (tap + 2 >= 9 ? 1 : 0);
            const unsigned char *Wr = smem + (tap % 3) * WSLICE;
            const unsigned char *Xb = smem + xcur + kh * (WW * 128);
            auto lda = [&](int a, int kk) { return *reinterpret_cast<const n_u32x4 *>(Wr + ((arow + a * 16 * 128) ^ (kk << 6))); };
            auto ldb = [&](int b, int kk) { return *reinterpret_cast<const n_u32x4 *>(Xb + ((bcol[kw] + b * WP * WW * 128) ^ (kk << 6))); };
            if constexpr (PP) {
                // ---- READ phase: every fragment of the step, then the step's DMA (slice of step + 2, a window piece) ----
                n_u32x4 af[2][TC], bf[NGRP];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int a = 0; a < TC; ++a) af[kk][a] = lda(a, kk);
#pragma unroll
                for (int g = 0; g < NGRP; ++g) bf[g] = ldb(g % TP, g / TP);
                issue_w(ncc, ntap, nring);
                if constexpr (XWIN2 && tap < XPW) issue_x(tap, cc + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA phase ----
#pragma unroll
                for (int g = 0; g < NGRP; ++g)
#pragma unroll
                    for (int a = 0; a < TC; ++a) acc[a][g % TP] = mfma_n16<F16>(af[g / TP][a], bf[g], acc[a][g % TP]);
                // everything this wave issued before this step's READ phase has landed (slice of step + 1, older window pieces)
                constexpr int cnt = WQ + ((XWIN2 && tap < XPW) ? 1 : 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cnt) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            } else {
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
                if constexpr (g == 0) issue_w(ncc, ntap, nring);
                if constexpr (g == 2 % NGRP && XBUFS == 2 && tap < XPW) issue_x(tap, cc + 1);
#pragma unroll
                for (int a = 0; a < TC; ++a) acc[a][b] = mfma_n16<F16>(af[kk][a], bf[g], acc[a][b]);
            });
            // issue order: TC MFMAs of group g, then the reads for group g+2 and the group's DMA pieces (see conv_n16.hip)
            __builtin_amdgcn_sched_group_barrier(0x100, TC + 2, 0);
            static_for<NGRP>([&](auto G) {
                constexpr int g = decltype(G)::v;
                __builtin_amdgcn_sched_group_barrier(0x008, TC, 0);
                constexpr int nread = (g + 2 < NGRP ? 1 : 0) + (g == 0 ? TC : 0);
                if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);
                constexpr int npiece = (g == 0 ? WQ : 0) + ((g == 2 % NGRP && XBUFS == 2 && tap < XPW) ? 1 : 0);
                if constexpr (npiece > 0) __builtin_amdgcn_sched_group_barrier(0x010, npiece, 0);
            });
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sink pieces of the last steps (zero fills: they return at once)

    // ---- direct epilogue (the encoder's specialised modes): accumulators -> global memory, no LDS round trip (conv_common.h) ----
    const int emode = epi_mode(p);
    if (emode != EPI_GENERIC && (p.Cout & 7) == 0) {
        constexpr int NARROW = F16 ? CER_STORE_F16 : CER_STORE_BF16;
        float s1[TC / 2][8], s2[TC / 2][8];
        EpiPix epx[TP];
        {
            const int gx = px * PWD + l15;
            const int rx = gx == 0 ? 0 : (gx == p.W - 1 ? 2 : 1);
#pragma unroll
            for (int b = 0; b < TP; ++b) {
                const int gy = py * PH + b * WP + wp;
                const int ry = gy == 0 ? 0 : (gy == p.H - 1 ? 2 : 1);
                const int m = (n * p.H + gy) * p.W + gx;
                epx[b] = EpiPix{true, (size_t)(p.y_s2d ? s2d_row(m, gy, gx, p.W) : m), 3 * ry + rx};
            }
        }
        epi_dispatch(emode, [&](auto MODE_) {
            constexpr int MODE = decltype(MODE_)::v;
            if constexpr (MODE != EPI_GENERIC) epi_direct_stores<MODE, NARROW, TC, TP>(p, acc, c0 + wc * TC * 16, kg, epx, s1, s2);
        });
        if constexpr (PP) {
            if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
        }
        if (p.stats) epi_direct_stats<TC, WP, BN>(p, s1, s2, reinterpret_cast<float *>(smem_n16p), wp, wc, kg, l15, tid, c0, (size_t)patch);
        return;
    }
    if constexpr (PP) {
        if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
    }

    // ---- staged epilogue (every other launch): accumulators -> LDS (fp32, 16-byte granules XOR-swizzled by the row) -> compact
    // coalesced loop ----
    constexpr int G = BN / 4, RPI = NT / G;
    static_assert(NT % G == 0 && RPI % 16 == 0 && 256 * BN * 4 <= SINK, "whole patch rows of 16 pixels per loop iteration; one pass");
    float *Ct = reinterpret_cast<float *>(smem_n16p);
    const int g = tid % G, r0 = tid / G, ox = r0 & 15;   // this thread's cout granule, first pixel and patch column
    const int c = c0 + g * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    EpiCtx ec;
    epi_init(p, c, ec);
    __syncthreads();  // the fragment reads of the last step are done
    static_for<TP>([&](auto B) {
        constexpr int b = decltype(B)::v;
        const int ml = (b * WP + wp) * 16 + l15;
        static_for<TC>([&](auto A) {
            constexpr int a = decltype(A)::v;
            const int gg = (wc * TC * 16 + epi_cout_of_row(a * 16 + kg * 4)) >> 2;   // the granule of the lane's 4 couts
            *reinterpret_cast<n_f32x4 *>(Ct + ml * BN + ((gg ^ (ml & 15)) << 2)) = acc[a][b];
        });
    });
    __syncthreads();
    const int gx = px * PWD + ox;
    const int rx = gx == 0 ? 0 : (gx == p.W - 1 ? 2 : 1);
    const size_t pix0 = ((size_t)n * p.H + (size_t)py * PH) * p.W + gx;
    epi_dispatch(ec.mode, [&](auto MODE_) {
        for (int oy = r0 >> 4; oy < PH; oy += RPI / 16) {
            const int ml = oy * 16 + ox;
            const int m = (int)(pix0 + (size_t)oy * p.W);
            const n_f32x4 q = *reinterpret_cast<const n_f32x4 *>(Ct + ml * BN + ((g ^ (ml & 15)) << 2));
            float v[4] = {q[0], q[1], q[2], q[3]};
            if (c < p.Cout) {
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    s1[t] += v[t];
                    s2[t] += v[t] * v[t];
                }
                const int gy = py * PH + oy;
                const int ry = gy == 0 ? 0 : (gy == p.H - 1 ? 2 : 1);
                epi_row<decltype(MODE_)::v>(p, ec, p.y_s2d ? s2d_row(m, gy, gx, p.W) : m, c, v, 3 * ry + rx);
            }
        }
    });
    if (p.stats) {
        __syncthreads();  // Ct has been consumed
        float *red = reinterpret_cast<float *>(smem_n16p);  // [RPI][2][BN]
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
            p.stats[((size_t)patch * 2 + 0) * p.Cout + c0 + tid] = t1;
            p.stats[((size_t)patch * 2 + 1) * p.Cout + c0 + tid] = t2;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv_n16_win_kernel: the window-resident structure for ANY image size (3x3 / stride 1 / pad 1) -- the 56x56 / 28x28 / 14x14 /
// 7x7 layers of the 224x224 pyramid and everything at the reference's 40x40 crop, which the 16x16-patch kernel cannot take.
// A block owns 256 CONSECUTIVE flattened output pixels m0 .. m0+255 (they may span image rows and images) and BN couts; the
// input of pixel m under tap (kh, kw) is pixel m + (kh-1) W + (kw-1) of the same flat array, so the window is the contiguous
// range [m0 - W - 1, m0 + 256 + W]: 258 + 2W rows of 128 bytes, fetched once per 64-channel chunk into one of two buffers; a
// tap is a row shift kh W + kw of the fragment address.  Taps outside the image (zero padding; in the flat array they are the
// neighbouring image row / frame) are masked per lane: the lane reads the window's last row, which the DMA zero-fills.
// Fragment rows start at any alignment: slot = chunk ^ (row & 6) is conflict free for all of them (tools/check_swizzle.py).
// PP (ping-pong): the two waves that share a SIMD (w and w + 4: the cout halves wc = 0 / 1) run half a step apart: a step is a
// READ phase (all 16 fragment reads + the step's DMA issue) and an MFMA phase (32 MFMAs, nothing else) with a block barrier
// after each, and group 1 starts one phase late, so one wave of a SIMD streams MFMAs while the other fetches.  Each group DMAs
// the weight rows of its own cout half (the other group never reads them): both keep the ring's two-step latency budget.
// XBUFS = 1: ONE window buffer (Cin == 64: a single chunk, nothing to prefetch) -- with 4 waves and 64 couts a block needs
// NP + 25 KiB, so two blocks share a CU and overlap each other's window fetch and epilogue (the 64 -> 64 layers at 40 x 40).
template <int BN, int WP, int WC, bool F16, bool PP, int XBUFS>
__global__ __launch_bounds__(WP * WC * 64, 2) void conv_n16_win_kernel(ConvArgs p, int NP, WinGeo geo) {
    constexpr int NW = WP * WC, NT = NW * 64, BM = 256;
    constexpr int NPMAX = 54;                                       // 8-row window pieces per buffer the LDS can hold twice (W <= 86)
    constexpr int XPW = (NPMAX + NW - 1) / NW;
    constexpr int WSLICE = BN * 128, RING = 3;
    static_assert(BN % (8 * NW) == 0, "weight-slice pieces are dealt round-robin to the waves");
    constexpr int WQ = BN / (8 * NW);
    constexpr int TP = BM / (16 * WP), TC = BN / (16 * WC);
    static_assert((XBUFS == 1 || XPW <= 9) && TP >= 2 && (XBUFS == 2 || !PP) && TC % 2 == 0, "geometry");
    static_assert(!PP || (WP == 4 && WC == 2 && (BN / 16) % 4 == 0), "ping-pong: waves w and w + 4 share a SIMD and split the couts");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NGRP = 2 * TP;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_n16p[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_n16p);
    const int XBYTES = NP * 1024, WOFF = XBUFS * XBYTES, SINK = WOFF + RING * WSLICE;
    const int ZROW = NP * 8 - 1;                                    // past the rows the taps address: always zero-filled

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
    const int tile_m = (int)fdiv((unsigned)bid, geo.tiles_n), tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, c0 = tile_n * BN;
    const int cin_steps = p.cin_steps;
    const int rows_needed = BM + 2 * p.W + 2;
    const int wstart = m0 - p.W - 1;                                // first window pixel (negative in the first tile)

    // (32-bit address arithmetic: conv_n16_win_ok bounds a window by 2^31 bytes and M = N H W is an int)
    const int prow = lane >> 3, slot = lane & 7;
    unsigned x_off[XPW];
    bool x_real[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int q = wave + NW * i;
        const int row = q * 8 + prow;
        const int pix = wstart + row;
        const bool inb = q < NP && row < rows_needed && pix >= 0 && pix < p.M;
        x_real[i] = q < NP;
        x_off[i] = inb ? (unsigned)(row * p.x_ld * 2) + (unsigned)((slot ^ (row & 6)) << 4) : OOB;
    }
    unsigned w_off[WQ];
    int w_piece[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        // PP: the group's own cout half (BN / 16 pieces of 8 rows) dealt to its four waves; else all pieces over all waves
        w_piece[i] = PP ? (wave >> 2) * (BN / 16) + (wave & 3) + 4 * i : wave + NW * i;
        const int row = w_piece[i] * 8 + prow;                                          // LDS row of the slice
        const int grow = row / (TC * 16) * (TC * 16) + epi_cout_of_row(row % (TC * 16));  // the cout it holds (conv_common.h)
        w_off[i] = c0 + grow < p.Cout ? (unsigned)(((size_t)grow * p.Kpad + ((slot ^ ((row >> 1) & 7)) << 3)) * 2) : OOB;
    }
    // base of the window's first pixel; lanes whose pixel lies outside the tensor carry the out-of-range offset instead
    const char *xwin = reinterpret_cast<const char *>(p.x_hi) + (long long)wstart * p.x_ld * 2;
    const char *wpanel = reinterpret_cast<const char *>(p.w_hi) + (size_t)c0 * p.Kpad * 2;

    auto issue_x = [&](int i, int cc) {
        const bool real = x_real[i] && cc < cin_steps;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xwin) + (size_t)cc * 128, 0, (int)OOB, 0x00020000);
        unsigned char *dst = real ? smem + (XBUFS == 2 ? (cc & 1) * XBYTES : 0) + (wave + NW * i) * 1024 : smem + SINK;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (n_lds_ptr_t)dst, 16, (int)(real ? x_off[i] : OOB), 0, 0, 0);
    };
    auto issue_w = [&](int cc, int tap, int ring) {
        const bool real = cc < cin_steps;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(wpanel) + ((size_t)tap * p.Cin + (size_t)cc * 64) * 2, 0, (int)OOB, 0x00020000);
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

    // per pixel tile b: the lane's window row under tap (0, 0) and the 9-bit mask of the taps that stay inside its image
    int prow0[TP];
    unsigned taps[TP];
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        const int pl = (b * WP + wp) * 16 + l15;
        prow0[b] = pl;
        const int m = m0 + pl;
        unsigned bits = 0;
        if (m < p.M) {
            const int r = m - (int)fdiv((unsigned)m, geo.hw) * (int)geo.hw.d;
            const int y = (int)fdiv((unsigned)r, geo.w), x = r - y * p.W;
            const unsigned rowm = (y > 0 ? 1u : 0u) | 2u | (y < p.H - 1 ? 4u : 0u), colm = (x > 0 ? 1u : 0u) | 2u | (x < p.W - 1 ? 4u : 0u);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                if ((rowm >> kh) & 1u) bits |= colm << (3 * kh);
        }
        taps[b] = bits;
    }
    const int arow = (wc * TC * 16 + l15) * 128 + ((kg ^ ((l15 >> 1) & 7)) << 4);

#pragma unroll
    for (int i = 0; i < XPW; ++i) issue_x(i, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    if constexpr (PP) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WQ) : "memory");   // the window and slice 0 (slice 1 may still be in flight)
        __builtin_amdgcn_s_barrier();
        if (wc == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one phase behind group 0
    }

    for (int cc = 0; cc < cin_steps; ++cc) {
        const int xcur = XBUFS == 2 ? (cc & 1) * XBYTES : 0;
        static_for<9>([&](auto T) {
            constexpr int tap = decltype(T)::v, kh = tap / 3, kw = tap % 3;
            constexpr bool XWIN2 = XBUFS == 2;
            if constexpr (!PP) {
                constexpr int ptap = (tap + 8) % 9;
                constexpr int pcnt = WQ + ((XWIN2 && ptap < XPW) ? 1 : 0);
                if (cc == 0 && tap == 0) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WQ) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(pcnt) : "memory");
                }
                __builtin_amdgcn_s_barrier();
            }
            constexpr int ntap = (tap + 2) % 9, nring = (tap + 2) % 3;
            const int ncc = cc + (tap + 2 >= 9 ? 1 : 0);
            const unsigned char *Wr = smem + WOFF + (tap % 3) * WSLICE;
            const unsigned char *Xb = smem + xcur;
            const int toff = kh * p.W + kw;
            int baddr[TP];
#pragma unroll
            for (int b = 0; b < TP; ++b) {
                const int row = ((taps[b] >> tap) & 1u) ? prow0[b] + toff : ZROW;
                baddr[b] = row * 128 + ((kg ^ (row & 6)) << 4);
            }
            auto lda = [&](int a, int kk) { return *reinterpret_cast<const n_u32x4 *>(Wr + ((arow + a * 16 * 128) ^ (kk << 6))); };
            auto ldb = [&](int b, int kk) { return *reinterpret_cast<const n_u32x4 *>(Xb + (baddr[b] ^ (kk << 6))); };
            if constexpr (PP) {
                // ---- READ phase: every fragment of the step, then the step's DMA (slice of step + 2, a window piece) ----
                n_u32x4 af[2][TC], bf[NGRP];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int a = 0; a < TC; ++a) af[kk][a] = lda(a, kk);
#pragma unroll
                for (int g = 0; g < NGRP; ++g) bf[g] = ldb(g % TP, g / TP);
                issue_w(ncc, ntap, nring);
                if constexpr (XWIN2 && tap < XPW) issue_x(tap, cc + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA phase ----
#pragma unroll
                for (int g = 0; g < NGRP; ++g)
#pragma unroll
                    for (int a = 0; a < TC; ++a) acc[a][g % TP] = mfma_n16<F16>(af[g / TP][a], bf[g], acc[a][g % TP]);
                // everything this wave issued before this step's READ phase has landed (slice of step + 1, older window pieces)
                constexpr int cnt = WQ + ((XWIN2 && tap < XPW) ? 1 : 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cnt) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            } else {
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
                if constexpr (g == 0) issue_w(ncc, ntap, nring);
                if constexpr (g == 2 % NGRP && XWIN2 && tap < XPW) issue_x(tap, cc + 1);
#pragma unroll
                for (int a = 0; a < TC; ++a) acc[a][b] = mfma_n16<F16>(af[kk][a], bf[g], acc[a][b]);
            });
            __builtin_amdgcn_sched_group_barrier(0x100, TC + 2, 0);
            static_for<NGRP>([&](auto G) {
                constexpr int g = decltype(G)::v;
                __builtin_amdgcn_sched_group_barrier(0x008, TC, 0);
                constexpr int nread = (g + 2 < NGRP ? 1 : 0) + (g == 0 ? TC : 0);
                if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);
                constexpr int npiece = (g == 0 ? WQ : 0) + ((g == 2 % NGRP && XWIN2 && tap < XPW) ? 1 : 0);
                if constexpr (npiece > 0) __builtin_amdgcn_sched_group_barrier(0x010, npiece, 0);
            });
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- direct epilogue (the encoder's specialised modes): accumulators -> global memory, no LDS round trip (conv_common.h) ----
    const int emode = epi_mode(p);
    if (emode != EPI_GENERIC && (p.Cout & 7) == 0) {
        constexpr int NARROW = F16 ? CER_STORE_F16 : CER_STORE_BF16;
        float s1[TC / 2][8], s2[TC / 2][8];
        // the lane's pixel of every pixel tile: its output row and border case (from the tap mask: a missing (0, 1) / (2, 1) /
        // (1, 0) / (1, 2) tap is the first / last image row / column)
        EpiPix epx[TP];
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            const int m = m0 + prow0[b];
            const unsigned bits = taps[b];
            const int ry = !((bits >> 1) & 1u) ? 0 : (!((bits >> 7) & 1u) ? 2 : 1), rx = !((bits >> 3) & 1u) ? 0 : (!((bits >> 5) & 1u) ? 2 : 1);
            epx[b] = EpiPix{m < p.M, (size_t)m, 3 * ry + rx};
            if (p.y_s2d && epx[b].live) {
                const int r = m % (p.Ho * p.Wo);
                const int ho = r / p.Wo;
                epx[b].row = (size_t)s2d_row(m, ho, r - ho * p.Wo, p.Wo);
            }
        }
        epi_dispatch(emode, [&](auto MODE_) {
            constexpr int MODE = decltype(MODE_)::v;
            if constexpr (MODE != EPI_GENERIC) epi_direct_stores<MODE, NARROW, TC, TP>(p, acc, c0 + wc * TC * 16, kg, epx, s1, s2);
        });
        if constexpr (PP) {
            if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
        }
        if (p.stats) epi_direct_stats<TC, WP, BN>(p, s1, s2, reinterpret_cast<float *>(smem_n16p), wp, wc, kg, l15, tid, c0, (size_t)tile_m);
        return;
    }
    if constexpr (PP) {
        if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
    }

    // ---- staged epilogue (every other launch): accumulators -> LDS (fp32, swizzled granules) -> compact coalesced loop over
    // consecutive output pixels ----
    constexpr int G = BN / 4, RPI = NT / G;
    static_assert(NT % G == 0, "one thread per granule");
    float *Ct = reinterpret_cast<float *>(smem_n16p);   // the launcher sizes the LDS for 256 * BN floats at least
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
            const int gg = (wc * TC * 16 + epi_cout_of_row(a * 16 + kg * 4)) >> 2;   // the granule of the lane's 4 couts
            *reinterpret_cast<n_f32x4 *>(Ct + ml * BN + ((gg ^ (ml & 15)) << 2)) = acc[a][b];
        });
    });
    __syncthreads();
    int ho = 0, wo = 0;
    const bool track = p.bias9 || p.y_s2d;   // the row loop needs the pixel's image coordinates
    if (track) {
        const int mm = m0 + r0 < p.M ? m0 + r0 : 0;
        const int r = mm % (p.Ho * p.Wo);
        ho = r / p.Wo;
        wo = r - ho * p.Wo;
    }
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
                const int ry = ho == 0 ? 0 : (ho == p.Ho - 1 ? 2 : 1), rx = wo == 0 ? 0 : (wo == p.Wo - 1 ? 2 : 1);
                epi_row<decltype(MODE_)::v>(p, ec, p.y_s2d ? s2d_row(m, ho, wo, p.Wo) : m, c, v, 3 * ry + rx);
            }
            if (track) {
                wo += RPI;
                while (wo >= p.Wo) {
                    wo -= p.Wo;
                    if (++ho == p.Ho) ho = 0;
                }
            }
        }
    });
    if (p.stats) {
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem_n16p);  // [RPI][2][BN]
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

// 8-row window pieces per buffer for this conv, or 0 when two buffers do not fit beside the 128-cout weight ring
static int n16_win_pieces(const ConvArgs &a) {
    const int np = (256 + 2 * a.W + 2 + 1 + 7) / 8;   // + 1: the last row stays zero (masked taps read it)
    return np <= 54 ? np : 0;
}

template <int BN, int WP, int WC, bool PP, int XBUFS = 2>
static int launch_win(const ConvArgs &a, hipStream_t st) {
    const int np = n16_win_pieces(a);
    size_t lds = (size_t)XBUFS * np * 1024 + 3 * (size_t)BN * 128 + 1024;
    if (lds < (size_t)256 * BN * 4) lds = (size_t)256 * BN * 4;   // the epilogue's accumulator tile
    const dim3 grid(a.tiles_m * a.tiles_n, 1, 1), block(WP * WC * 64);
    const WinGeo geo{make_fastdiv((unsigned)a.tiles_n), make_fastdiv((unsigned)(a.H * a.W)), make_fastdiv((unsigned)a.W)};
    if (a.narrow == CER_STORE_F16) {
        auto k = conv_n16_win_kernel<BN, WP, WC, true, PP, XBUFS>;
        if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a, np, geo);
    } else {
        auto k = conv_n16_win_kernel<BN, WP, WC, false, PP, XBUFS>;
        if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a, np, geo);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

bool conv_n16_win_ok(const ConvArgs &a) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.dil_h != 1 || a.dil_w != 1 || a.pad_t != 1 || a.pad_l != 1 || a.Ho != a.H ||
        a.Wo != a.W || a.H < 2 || a.W < 2 || (a.Cin & 63) || a.split_k != 1 || n16_win_pieces(a) == 0)
        return false;
    return (long long)(512 + 2 * a.W) * a.x_ld * 2 < (1ll << 31) && (long long)128 * a.Kpad * 2 < (1ll << 31);
}

template <int BN, int WP, int WC, int XBUFS, bool PP>
static int launch_patch(const ConvArgs &a, hipStream_t st) {
    constexpr int XPIECES = 41;
    const size_t lds = (size_t)XBUFS * XPIECES * 1024 + 3 * (size_t)BN * 128 + 1024;
    const dim3 grid(a.tiles_m * a.tiles_n, 1, 1), block(WP * WC * 64);
    const PatchGeo geo{make_fastdiv((unsigned)a.tiles_n), make_fastdiv((unsigned)(a.W / 16)), make_fastdiv((unsigned)(a.H / 16))};
    if (a.narrow == CER_STORE_F16) {
        auto k = conv_n16_patch_kernel<BN, WP, WC, XBUFS, true, PP>;
        if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a, geo);
    } else {
        auto k = conv_n16_patch_kernel<BN, WP, WC, XBUFS, false, PP>;
        if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, grid, block, lds, st, a, geo);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

// tile ids: 71 = 16x16 patch x 64 couts, 4 waves, ONE window buffer (Cin == 64 only; 74 KiB of LDS -> two blocks per CU, so
// one block's window fetch overlaps the other's nine steps); 72 = 16x16 patch x 128 couts, 8 waves, two window buffers
bool conv_n16_patch_ok(const ConvArgs &a, int tile) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.dil_h != 1 || a.dil_w != 1 || a.pad_t != 1 || a.pad_l != 1 || a.Ho != a.H ||
        a.Wo != a.W || (a.H & 15) || (a.W & 15) || (a.Cin & 63) || a.split_k != 1)
        return false;
    if ((long long)a.H * a.W * a.x_ld * 2 >= (1ll << 31) || (long long)128 * a.Kpad * 2 >= (1ll << 31)) return false;
    if (tile == 71) return a.Cin == 64;
    return tile == 72 || tile == 78;
}

int conv_n16_patch_launch(int tile, const ConvArgs &a, hipStream_t st) {
    if (tile == 77) {
        if (!conv_n16_win_ok(a) || a.Cin != 64)
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow, single-window kernel): needs a 3x3 / stride 1 / pad 1 conv with "
                                                       "W <= 86 and Cin == 64, no split-K");
        return launch_win<64, 4, 1, false, 1>(a, st);
    }
    if (tile == 73 || tile == 76) {
        if (!conv_n16_win_ok(a))
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow, window kernel): needs a 3x3 / stride 1 / pad 1 conv with W <= 86, "
                                                       "Cin % 64 == 0, no split-K");
        return tile == 73 ? launch_win<64, 4, 2, false>(a, st) : launch_win<128, 4, 2, true>(a, st);   // 76: ping-pong phases
    }
    if (!conv_n16_patch_ok(a, tile))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow, patch kernel): needs a 3x3 / stride 1 / pad 1 conv on images whose "
                                                   "height and width are multiples of 16, Cin % 64 == 0 (tile 71: Cin == 64), no split-K");
    switch (tile) {
        case 71: return launch_patch<64, 4, 1, 1, false>(a, st);
        case 72: return launch_patch<128, 4, 2, 2, false>(a, st);
        case 78: return launch_patch<128, 4, 2, 2, true>(a, st);     // ping-pong variant of 72
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (narrow, patch kernel): unknown tile id");
    }
}

}  // namespace cer
