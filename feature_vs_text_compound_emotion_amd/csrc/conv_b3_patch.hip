// conv_b3_patch.hip -- the bf16x3 (split hi/lo operand) 3x3 / stride 1 / pad 1 convolution with the INPUT WINDOW RESIDENT
// IN LDS: the split-operand twin of conv_n16_patch.hip (read that file's header for the why and the PMC evidence).
//
//   * a block owns a 16 x 16 patch of output pixels of one image and BN output channels;
//   * per 32-channel chunk the 18 x 18 input window (halo included, out-of-image pixels = zeros from the range-checked
//     DMA) is fetched ONCE: two planes (hi, lo) of 324 rows x 64 bytes = 2 x 21 one-KiB LDS-DMA pieces; all nine taps
//     read their pixel fragments from it at the row shift (r + kh) * 18 + kw.  64-byte rows put four window pixels in a
//     256-byte bank row; the conflict-free swizzle for fragment rows that start at any alignment is a function of the
//     window COLUMN found by exhaustive search (tools/check_swizzle.py: B3_PATCH_F);
//   * the weights stream: one BN x 32 slice (two planes) per (chunk, tap) through a 3-slot ring, two steps ahead, counted
//     vmcnt; the NEXT chunk's window is fetched a whole chunk (nine steps) ahead into the second window buffer;
//   * a * b = a_hi * b_hi + a_hi * b_lo + a_lo * b_hi: three v_mfma_f32_16x16x32_bf16 per 16x16 tile and step; taps unrolled;
//   * PING-PONG PHASES: the two waves that share a SIMD (wave w and w + 4: the cout halves wc = 0 / 1) run half a step out of
//     phase.  A step is a READ phase (all 16 fragment reads of the step + the step's DMA issue) and an MFMA phase (48 MFMAs,
//     nothing else) with a block barrier after each; group 1 starts one phase late, so while one wave of a SIMD streams MFMAs
//     the other fetches.  Each group DMAs the weight rows of ITS cout half (nobody else reads them), which keeps the ring's
//     two-step latency budget in both groups.  (The earlier single-phase version -- reads two MFMA groups ahead, DMA pieces
//     between MFMA groups, one barrier per step -- was 1-8 % slower on every layer and was removed; it is in the history.)
//
// Against the flat 128-pixel tile this cuts the L2 -> LDS activation traffic of a 3x3 layer from nine fetches of every
// input line to 1.27 (the halo), the round-1 limiter of the bf16x3 kernel ("both operands' L2 -> LDS
// traffic"; roofline.traffic 2.2x the algorithmic bytes).
#include "conv_b3.h"

namespace cer {

// slot = chunk ^ B3_PATCH_F[window column], 2 bits per column (tools/check_swizzle.py)
constexpr unsigned long long B3_PATCH_F_TABLE = 0xaaa00a00ull;
__device__ __forceinline__ int b3_patch_f(int wx) { return (int)((B3_PATCH_F_TABLE >> (2 * wx)) & 3ull); }

// block -> (patch, cout tile) -> (image, patch row, patch column) through reciprocals of the launch constants (conv_common.h)
struct B3PatchGeo { FastDiv tiles_n, pxn, pyn; };

// SINGLE: 4 waves (WC = 1, each wave 64 pixels x 64 couts), ONE window buffer that is re-filled at every chunk boundary, the
// single-phase step (reads two MFMA groups ahead, DMA between MFMA groups, one barrier per step): 67 KiB of LDS, so TWO blocks
// share a CU and cover each other's window fetches and epilogue -- the K-short 64-cout layers.
template <int BN, int WP, int WC, bool SINGLE>
__global__ __launch_bounds__(WP * WC * 64, 2) void conv_b3_patch_kernel(ConvArgs p, B3PatchGeo geo) {
    constexpr int NW = WP * WC, NT = NW * 64;
    constexpr int PH = 16, PWD = 16, WW = 18, WROWS = 18 * 18;
    constexpr int XPP = (WROWS + 15) / 16;                        // 21 one-KiB pieces (16 rows of 64 bytes) per plane
    constexpr int XPW = (XPP + NW - 1) / NW;                      // window pieces per wave, plane and chunk
    constexpr int XPL = XPP * 1024, XBYTES = 2 * XPL;             // plane / buffer bytes
    constexpr int WPL = BN * 64, WSLICE = 2 * WPL;                // weight plane / slice bytes
    constexpr int WPIECES = 2 * BN / 16;                          // weight pieces per step (both planes)
    static_assert(WPIECES % NW == 0, "weight-slice pieces are dealt round-robin to the waves");
    constexpr int WQ = WPIECES / NW;                              // weight pieces per wave and step
    constexpr int TP = PH / WP, TC = BN / (16 * WC);
    static_assert(PH % WP == 0 && (SINGLE || XPW <= 9) && TP >= 2 && TC % 2 == 0, "geometry");
    static_assert(SINGLE ? WC == 1 : (WP == 4 && WC == 2 && (BN / 16) % 4 == 0), "ping-pong: waves w and w + 4 share a SIMD and split the couts");
    constexpr int WOFF = (SINGLE ? 1 : 2) * XBYTES, SINK = WOFF + 3 * WSLICE;    // LDS map: window(s) | weight ring | 1 KiB sink
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_b3p[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_b3p);

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
    const int patch = (int)fdiv((unsigned)bid, geo.tiles_n), tile_n = bid - patch * p.tiles_n;
    const int prw = (int)fdiv((unsigned)patch, geo.pxn), px = patch - prw * (int)geo.pxn.d;
    const int n = (int)fdiv((unsigned)prw, geo.pyn), py = prw - n * (int)geo.pyn.d;
    const int c0 = tile_n * BN;
    const int cin_steps = p.cin_steps;

    // ---- DMA assignment: in a piece lane l owns row l / 4 (16 rows), LDS slot l % 4.  32-bit address arithmetic
    // (conv_b3_patch_ok bounds an image and a weight panel by 2^31 bytes): tools/exp_stamp.py measured 2.9 us of a block's life
    // in the 64-bit / emulated-division version of this prologue on the narrow twin of this kernel ----
    const int prow = lane >> 2, slot = lane & 3;
    unsigned x_off[XPW];
    bool x_real[XPW];
    {
        const int iy0 = py * PH - 1, ix0 = px * PWD - 1, pitch = p.x_ld * 2;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int q = wave + NW * i;
            const int row = q * 16 + prow;
            const int wy = row / WW, wx = row - wy * WW;
            const int iy = iy0 + wy, ix = ix0 + wx;
            const bool inb = q < XPP && row < WROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            x_real[i] = q < XPP;
            x_off[i] = inb ? (unsigned)((iy * p.W + ix) * pitch) + (unsigned)((slot ^ b3_patch_f(wx)) << 4) : OOB;
        }
    }
    unsigned w_off[WQ];
    int w_plane[WQ], w_dst[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        // the group's own cout half, BN / 32 pieces per plane dealt to its four waves
        const int j = SINGLE ? wave + NW * i : (wave & 3) + 4 * i;
        constexpr int PPL = SINGLE ? BN / 16 : BN / 32;             // pieces per plane in this wave's pool
        w_plane[i] = j / PPL;
        const int pc = (SINGLE ? 0 : (wave >> 2) * PPL) + j % PPL;  // piece (16 cout rows) within the plane
        const int row = pc * 16 + prow;                                                   // LDS row of the slice
        const int grow = row / (TC * 16) * (TC * 16) + epi_cout_of_row(row % (TC * 16));  // the cout it holds (conv_common.h)
        w_dst[i] = w_plane[i] * WPL + pc * 1024;
        w_off[i] = c0 + grow < p.Cout ? (unsigned)(grow * p.Kpad * 2) + (unsigned)((slot ^ swz16((row >> 2) & 3)) << 4) : OOB;
    }
    const size_t img = (size_t)n * p.H * p.W * p.x_ld * 2;
    const char *xh = reinterpret_cast<const char *>(p.x_hi) + img, *xl = reinterpret_cast<const char *>(p.x_lo) + img;
    const size_t wpan = (size_t)c0 * p.Kpad * 2;
    const char *wh = reinterpret_cast<const char *>(p.w_hi) + wpan, *wl = reinterpret_cast<const char *>(p.w_lo) + wpan;

    // window piece i (both planes) of chunk cc into window buffer cc & 1 -- or two zero pieces into the sink, which keeps
    // the per-step DMA count of a wave constant (the counted vmcnt relies on it)
    auto issue_x = [&](int i, int cc) {
        const bool real = x_real[i] && cc < cin_steps;
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xh) + (size_t)cc * 64, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xl) + (size_t)cc * 64, 0, (int)OOB, 0x00020000);
        unsigned char *dst = real ? smem + (SINGLE ? 0 : (cc & 1) * XBYTES) + (wave + NW * i) * 1024 : smem + SINK;
        const int vo = (int)(real ? x_off[i] : OOB);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_ptr_t)(real ? dst + XPL : dst), 16, vo, 0, 0, 0);
    };
    auto issue_w = [&](int cc, int tap, int ring) {
        const bool real = cc < cin_steps;
        const size_t koff = ((size_t)tap * p.Cin + (size_t)cc * 32) * 2;
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            // (the plane is chosen on the POINTER: a ?: between two buffer resources made hipcc's host pass drop the kernel stub)
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

    // fragment address bases (hi plane; lo plane = + WPL / + XPL)
    const int arow = WOFF + (wc * TC * 16 + l15) * 64 + ((kg ^ swz16((l15 >> 2) & 3)) << 4);
    int bcol[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) bcol[kw] = (wp * WW + kw + l15) * 64 + ((kg ^ b3_patch_f(kw + l15)) << 4);

    // prologue: the whole window of chunk 0, weight slices of steps 0 and 1
#pragma unroll
    for (int i = 0; i < XPW; ++i) issue_x(i, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    if constexpr (!SINGLE) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WQ) : "memory");   // the window and slice 0 (slice 1 may still be in flight)
        __builtin_amdgcn_s_barrier();
        if (wc == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one phase behind group 0
    }

    // One step = one filter tap of one 32-channel chunk; per wave and step WQ weight pieces (slice of step s + 2) and, during
    // taps 0 .. XPW-1, the two planes of one window piece of the next chunk.  At the top of step s everything issued before
    // step s-1 must have landed: vmcnt(pieces of the previous step).
    for (int cc = 0; cc < cin_steps; ++cc) {
        const int xcur = SINGLE ? 0 : (cc & 1) * XBYTES;
        if constexpr (SINGLE) {
            if (cc > 0) {   // everyone is done with the previous chunk's window: re-fill the one buffer (the other block computes meanwhile)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < XPW; ++i) issue_x(i, cc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        static_for<9>([&](auto T) {
            constexpr int tap = decltype(T)::v, kh = tap / 3, kw = tap % 3;
            constexpr int ntap = (tap + 2) % 9, nring = (tap + 2) % 3;
            const int ncc = cc + (tap + 2 >= 9 ? 1 : 0);
            const unsigned char *Wr = smem + (tap % 3) * WSLICE;
            const unsigned char *Xb = smem + xcur + kh * (WW * 64);
            // (fragments travel as u32x4 and are re-typed at the MFMA: a lambda returning a __bf16 vector made hipcc's host pass
            // drop the kernel stub)
            auto lda = [&](int a, int pl) { return *reinterpret_cast<const u32x4 *>(Wr + pl * WPL + arow + a * 16 * 64); };
            auto ldb = [&](int b, int pl) { return *reinterpret_cast<const u32x4 *>(Xb + pl * XPL + bcol[kw] + b * WP * WW * 64); };
            if constexpr (!SINGLE) {
                // ---- READ phase: every fragment of the step, then the step's DMA (slice of step + 2, a window piece) ----
                u32x4 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
                for (int a = 0; a < TC; ++a) { ah[a] = lda(a, 0); al[a] = lda(a, 1); }
#pragma unroll
                for (int b = 0; b < TP; ++b) { bh[b] = ldb(b, 0); bl[b] = ldb(b, 1); }
                issue_w(ncc, ntap, nring);
                if constexpr (tap < XPW) issue_x(tap, cc + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA phase: consecutive MFMAs go to different accumulators (lo*hi, hi*lo, hi*hi per accumulator) ----
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int g = 0; g < TP; ++g)
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al[a]), as_bf16x8(bh[g]), acc[a][g], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int g = 0; g < TP; ++g)
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bl[g]), acc[a][g], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int g = 0; g < TP; ++g)
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bh[g]), acc[a][g], 0, 0, 0);
                // everything this wave issued before this step's READ phase has landed (slice of step + 1, older window pieces)
                constexpr int cnt = WQ + (tap < XPW ? 2 : 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cnt) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // single-phase step: at its top the slice of this step has landed (issued two steps ago); the step issues the
                // slice of step + 2 after its first MFMA group
                if (cc == 0 && tap == 0) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WQ) : "memory");  // prologue: all but slice 1
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WQ) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                u32x4 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
                for (int a = 0; a < TC; ++a) { ah[a] = lda(a, 0); al[a] = lda(a, 1); }
                bh[0] = ldb(0, 0); bl[0] = ldb(0, 1);
                bh[1] = ldb(1, 0); bl[1] = ldb(1, 1);
                static_for<TP>([&](auto G) {
                    constexpr int g = decltype(G)::v;
                    if constexpr (g + 2 < TP) { bh[g + 2] = ldb(g + 2, 0); bl[g + 2] = ldb(g + 2, 1); }
                    if constexpr (g == 0) issue_w(ncc, ntap, nring);
#pragma unroll
                    for (int a = 0; a < TC; ++a) {
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al[a]), as_bf16x8(bh[g]), acc[a][g], 0, 0, 0);
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bl[g]), acc[a][g], 0, 0, 0);
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bh[g]), acc[a][g], 0, 0, 0);
                    }
                });
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * TC + 4, 0);
                static_for<TP>([&](auto G) {
                    constexpr int g = decltype(G)::v;
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * TC, 0);
                    if constexpr (g + 2 < TP) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if constexpr (g == 0) __builtin_amdgcn_sched_group_barrier(0x010, WQ, 0);
                });
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sink pieces of the last steps (zero fills: they return at once)

    // ---- direct epilogue (the encoder's specialised modes): accumulators -> global memory, no LDS round trip (conv_common.h) ----
    const int emode = epi_mode(p);
    if (emode != EPI_GENERIC && (p.Cout & 7) == 0) {
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
            if constexpr (MODE != EPI_GENERIC) epi_direct_stores<MODE, CER_STORE_BF16, TC, TP>(p, acc, c0 + wc * TC * 16, kg, epx, s1, s2);
        });
        if constexpr (!SINGLE) {
            if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
        }
        if (p.stats) epi_direct_stats<TC, WP, BN>(p, s1, s2, reinterpret_cast<float *>(smem_b3p), wp, wc, kg, l15, tid, c0, (size_t)patch);
        return;
    }
    if constexpr (!SINGLE) {
        if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
    }

    // ---- staged epilogue (every other launch): accumulators -> LDS (fp32, swizzled granules) -> compact coalesced loop, one patch
    // row per iteration ----
    constexpr int G = BN / 4, RPI = NT / G;
    static_assert(NT % G == 0 && RPI % 16 == 0 && 256 * BN * 4 <= SINK, "whole patch rows of 16 pixels per loop iteration; one pass");
    float *Ct = reinterpret_cast<float *>(smem_b3p);
    const int g = tid % G, r0 = tid / G, ox = r0 & 15;
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
            *reinterpret_cast<f32x4 *>(Ct + ml * BN + ((gg ^ (ml & 15)) << 2)) = acc[a][b];
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
            const f32x4 q = *reinterpret_cast<const f32x4 *>(Ct + ml * BN + ((g ^ (ml & 15)) << 2));
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
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem_b3p);  // [RPI][2][BN]
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
// conv_b3_win_kernel: the window-resident structure for ANY image size (3x3 / stride 1 / pad 1): the layers the patch kernel
// cannot take -- 56x56 and 28x28 at 224x224 input, everything at the reference's 40x40 crop.
//
// A block owns 256 CONSECUTIVE flattened output pixels m0 .. m0+255 (they may span image rows and images) and BN couts.  For a
// stride-1 "same" conv the input of output pixel m under tap (kh, kw) is pixel m + (kh-1)*W + (kw-1) of the same flat
// [N*H*W] array, so the window is the contiguous pixel range [m0 - W - 1, m0 + 255 + W + 1]: 256 + 2W + 2 rows of 64 bytes per
// plane, fetched once per 32-channel chunk and double buffered; a tap is a row shift kh*W + kw of the fragment address.
// Taps that fall off the image (zero padding; the previous / next image row or frame in the flat array) are masked per lane:
// such a lane reads the window's last row, which the DMA zero-fills.  Fragment rows start at any alignment; the swizzle
// slot = chunk ^ ((row & 4) >> 1) is conflict free for every alignment (tools/check_swizzle.py).  Everything else -- weight
// ring, counted vmcnt, unrolled taps, ping-pong phases -- is the patch kernel's.
// SINGLE: as in conv_b3_patch_kernel -- 4 waves, one window buffer re-filled per chunk, single-phase steps, two blocks per CU.
template <int BN, int WP, int WC, bool SINGLE>
__global__ __launch_bounds__(WP * WC * 64, 2) void conv_b3_win_kernel(ConvArgs p, int NP, WinGeo geo) {
    constexpr int NW = WP * WC, NT = NW * 64, BM = 256;
    constexpr int NPMAX = 27;                                     // window pieces (16 rows) per plane the LDS can hold twice
    constexpr int XPW = (NPMAX + NW - 1) / NW;                    // window pieces per wave, plane and chunk (upper bound)
    constexpr int WPL = BN * 64, WSLICE = 2 * WPL, WPIECES = 2 * BN / 16;
    static_assert(WPIECES % NW == 0, "weight-slice pieces are dealt round-robin to the waves");
    constexpr int WQ = WPIECES / NW;
    constexpr int TP = BM / (16 * WP), TC = BN / (16 * WC);
    static_assert((SINGLE || XPW <= 9) && TP >= 2 && TC % 2 == 0, "geometry");
    static_assert(SINGLE ? WC == 1 : (WP == 4 && WC == 2 && (BN / 16) % 4 == 0), "ping-pong: waves w and w + 4 share a SIMD and split the couts");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_b3p[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_b3p);
    const int XPL = NP * 1024, XBYTES = 2 * XPL, WOFF = (SINGLE ? 1 : 2) * XBYTES, SINK = WOFF + 3 * WSLICE;
    const int ZROW = NP * 16 - 1;                                 // always zero-filled by the DMA

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
    const int wstart = m0 - p.W - 1;                              // first window pixel (may be negative)

    // ---- DMA assignment (32-bit address arithmetic: conv_b3_win_ok bounds a window by 2^31 bytes, M = N H W is an int) ----
    const int prow = lane >> 2, slot = lane & 3;
    unsigned x_off[XPW];
    bool x_real[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int q = wave + NW * i;
        const int row = q * 16 + prow;
        const int pix = wstart + row;
        const bool inb = q < NP && row < rows_needed && pix >= 0 && pix < p.M;
        x_real[i] = q < NP;
        x_off[i] = inb ? (unsigned)(row * p.x_ld * 2) + (unsigned)((slot ^ ((row & 4) >> 1)) << 4) : OOB;
    }
    unsigned w_off[WQ];
    int w_plane[WQ], w_dst[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        // the group's own cout half, BN / 32 pieces per plane dealt to its four waves
        const int j = SINGLE ? wave + NW * i : (wave & 3) + 4 * i;
        constexpr int PPL = SINGLE ? BN / 16 : BN / 32;             // pieces per plane in this wave's pool
        w_plane[i] = j / PPL;
        const int pc = (SINGLE ? 0 : (wave >> 2) * PPL) + j % PPL;  // piece (16 cout rows) within the plane
        const int row = pc * 16 + prow;                                                   // LDS row of the slice
        const int grow = row / (TC * 16) * (TC * 16) + epi_cout_of_row(row % (TC * 16));  // the cout it holds (conv_common.h)
        w_dst[i] = w_plane[i] * WPL + pc * 1024;
        w_off[i] = c0 + grow < p.Cout ? (unsigned)(grow * p.Kpad * 2) + (unsigned)((slot ^ swz16((row >> 2) & 3)) << 4) : OOB;
    }
    // 64-bit base of the window's first pixel (never dereferenced where it points outside the tensor: those lanes are OOB)
    const long long wbase = (long long)wstart * p.x_ld * 2;
    const char *xh = reinterpret_cast<const char *>(p.x_hi) + wbase, *xl = reinterpret_cast<const char *>(p.x_lo) + wbase;
    const size_t wpan = (size_t)c0 * p.Kpad * 2;
    const char *wh = reinterpret_cast<const char *>(p.w_hi) + wpan, *wl = reinterpret_cast<const char *>(p.w_lo) + wpan;

    auto issue_x = [&](int i, int cc) {
        const bool real = x_real[i] && cc < cin_steps;
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xh) + (size_t)cc * 64, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xl) + (size_t)cc * 64, 0, (int)OOB, 0x00020000);
        unsigned char *dst = real ? smem + (SINGLE ? 0 : (cc & 1) * XBYTES) + (wave + NW * i) * 1024 : smem + SINK;
        const int vo = (int)(real ? x_off[i] : OOB);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_ptr_t)(real ? dst + XPL : dst), 16, vo, 0, 0, 0);
    };
    auto issue_w = [&](int cc, int tap, int ring) {
        const bool real = cc < cin_steps;
        const size_t koff = ((size_t)tap * p.Cin + (size_t)cc * 32) * 2;
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

    // per pixel tile (t = b * WP + wp): the lane's window row under tap (0,0) and the 9-bit mask of the taps inside its image
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
    const int arow = (wc * TC * 16 + l15) * 64 + ((kg ^ swz16((l15 >> 2) & 3)) << 4);

#pragma unroll
    for (int i = 0; i < XPW; ++i) issue_x(i, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    if constexpr (!SINGLE) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WQ) : "memory");   // the window and slice 0 (slice 1 may still be in flight)
        __builtin_amdgcn_s_barrier();
        if (wc == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one phase behind group 0
    }

    for (int cc = 0; cc < cin_steps; ++cc) {
        const int xcur = SINGLE ? 0 : (cc & 1) * XBYTES;
        if constexpr (SINGLE) {
            if (cc > 0) {   // everyone is done with the previous chunk's window: re-fill the one buffer (the other block computes meanwhile)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < XPW; ++i) issue_x(i, cc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        static_for<9>([&](auto T) {
            constexpr int tap = decltype(T)::v, kh = tap / 3, kw = tap % 3;
            constexpr int ntap = (tap + 2) % 9, nring = (tap + 2) % 3;
            const int ncc = cc + (tap + 2 >= 9 ? 1 : 0);
            const unsigned char *Wr = smem + WOFF + (tap % 3) * WSLICE;
            const unsigned char *Xb = smem + xcur;
            const int toff = kh * p.W + kw;
            auto lda = [&](int a, int pl) { return *reinterpret_cast<const u32x4 *>(Wr + pl * WPL + arow + a * 16 * 64); };
            int baddr[TP];
#pragma unroll
            for (int b = 0; b < TP; ++b) {
                const int row = ((taps[b] >> tap) & 1u) ? prow0[b] + toff : ZROW;
                baddr[b] = row * 64 + ((kg ^ ((row & 4) >> 1)) << 4);
            }
            auto ldb = [&](int b, int pl) { return *reinterpret_cast<const u32x4 *>(Xb + pl * XPL + baddr[b]); };
            if constexpr (!SINGLE) {
                // ---- READ phase: every fragment of the step, then the step's DMA (slice of step + 2, a window piece) ----
                u32x4 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
                for (int a = 0; a < TC; ++a) { ah[a] = lda(a, 0); al[a] = lda(a, 1); }
#pragma unroll
                for (int b = 0; b < TP; ++b) { bh[b] = ldb(b, 0); bl[b] = ldb(b, 1); }
                issue_w(ncc, ntap, nring);
                if constexpr (tap < XPW) issue_x(tap, cc + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA phase: consecutive MFMAs go to different accumulators (lo*hi, hi*lo, hi*hi per accumulator) ----
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int g = 0; g < TP; ++g)
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al[a]), as_bf16x8(bh[g]), acc[a][g], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int g = 0; g < TP; ++g)
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bl[g]), acc[a][g], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TC; ++a)
#pragma unroll
                    for (int g = 0; g < TP; ++g)
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bh[g]), acc[a][g], 0, 0, 0);
                // everything this wave issued before this step's READ phase has landed (slice of step + 1, older window pieces)
                constexpr int cnt = WQ + (tap < XPW ? 2 : 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cnt) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            } else {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WQ) : "memory");   // this step's slice has landed
                __builtin_amdgcn_s_barrier();
                u32x4 ah[TC], al[TC], bh[TP], bl[TP];
#pragma unroll
                for (int a = 0; a < TC; ++a) { ah[a] = lda(a, 0); al[a] = lda(a, 1); }
                bh[0] = ldb(0, 0); bl[0] = ldb(0, 1);
                bh[1] = ldb(1, 0); bl[1] = ldb(1, 1);
                static_for<TP>([&](auto G) {
                    constexpr int g = decltype(G)::v;
                    if constexpr (g + 2 < TP) { bh[g + 2] = ldb(g + 2, 0); bl[g + 2] = ldb(g + 2, 1); }
                    if constexpr (g == 0) issue_w(ncc, ntap, nring);
#pragma unroll
                    for (int a = 0; a < TC; ++a) {
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al[a]), as_bf16x8(bh[g]), acc[a][g], 0, 0, 0);
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bl[g]), acc[a][g], 0, 0, 0);
                        acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah[a]), as_bf16x8(bh[g]), acc[a][g], 0, 0, 0);
                    }
                });
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * TC + 4, 0);
                static_for<TP>([&](auto G) {
                    constexpr int g = decltype(G)::v;
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * TC, 0);
                    if constexpr (g + 2 < TP) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if constexpr (g == 0) __builtin_amdgcn_sched_group_barrier(0x010, WQ, 0);
                });
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- direct epilogue (the encoder's specialised modes): accumulators -> global memory, no LDS round trip (conv_common.h) ----
    const int emode = epi_mode(p);
    if (emode != EPI_GENERIC && (p.Cout & 7) == 0) {
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
            if constexpr (MODE != EPI_GENERIC) epi_direct_stores<MODE, CER_STORE_BF16, TC, TP>(p, acc, c0 + wc * TC * 16, kg, epx, s1, s2);
        });
        if constexpr (!SINGLE) {
            if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
        }
        if (p.stats) epi_direct_stats<TC, WP, BN>(p, s1, s2, reinterpret_cast<float *>(smem_b3p), wp, wc, kg, l15, tid, c0, (size_t)tile_m);
        return;
    }
    if constexpr (!SINGLE) {
        if (wc == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last phase boundary
    }

    // ---- staged epilogue (every other launch): accumulators -> LDS (fp32, swizzled granules) -> compact coalesced loop over
    // consecutive output rows ----
    constexpr int G = BN / 4, RPI = NT / G;
    static_assert(NT % G == 0, "one thread per granule");
    float *Ct = reinterpret_cast<float *>(smem_b3p);   // the launcher sizes the LDS for 256 * BN floats at least
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
            *reinterpret_cast<f32x4 *>(Ct + ml * BN + ((gg ^ (ml & 15)) << 2)) = acc[a][b];
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
            const f32x4 q = *reinterpret_cast<const f32x4 *>(Ct + ml * BN + ((g ^ (ml & 15)) << 2));
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
        float *red = reinterpret_cast<float *>(smem_b3p);  // [RPI][2][BN]
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

// window pieces per plane for this conv, or 0 when the window does not fit twice beside the weight ring
static int b3_win_pieces(const ConvArgs &a) {
    const int np = (256 + 2 * a.W + 2 + 1 + 15) / 16;   // + 1: the last row stays zero (masked taps read it)
    return np <= 27 ? np : 0;                           // 4 * 27 KiB + the 48 KiB weight ring + sink <= 160 KiB: W <= 86
}

template <int BN, int WP, int WC, bool SINGLE = false>
static int launch_b3_win(const ConvArgs &a, hipStream_t st) {
    const int np = b3_win_pieces(a);
    size_t lds = (size_t)(SINGLE ? 2 : 4) * np * 1024 + 3 * (size_t)2 * BN * 64 + 1024;
    if (lds < (size_t)256 * BN * 4) lds = (size_t)256 * BN * 4;   // the epilogue's accumulator tile
    auto k = conv_b3_win_kernel<BN, WP, WC, SINGLE>;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const WinGeo geo{make_fastdiv((unsigned)a.tiles_n), make_fastdiv((unsigned)(a.H * a.W)), make_fastdiv((unsigned)a.W)};
    CER_LAUNCH(k, dim3(a.tiles_m * a.tiles_n, 1, 1), dim3(WP * WC * 64), lds, st, a, np, geo);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

template <int BN, int WP, int WC, bool SINGLE = false>
static int launch_b3_patch(const ConvArgs &a, hipStream_t st) {
    const size_t lds = (size_t)(SINGLE ? 1 : 2) * 2 * 21 * 1024 + 3 * (size_t)2 * BN * 64 + 1024;
    auto k = conv_b3_patch_kernel<BN, WP, WC, SINGLE>;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const B3PatchGeo geo{make_fastdiv((unsigned)a.tiles_n), make_fastdiv((unsigned)(a.W / 16)), make_fastdiv((unsigned)(a.H / 16))};
    CER_LAUNCH(k, dim3(a.tiles_m * a.tiles_n, 1, 1), dim3(WP * WC * 64), lds, st, a, geo);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

bool conv_b3_patch_ok(const ConvArgs &a) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.dil_h != 1 || a.dil_w != 1 || a.pad_t != 1 || a.pad_l != 1 || a.Ho != a.H ||
        a.Wo != a.W || (a.H & 15) || (a.W & 15) || (a.Cin & 31) || a.split_k != 1)
        return false;
    return (long long)a.H * a.W * a.x_ld * 2 < (1ll << 31) && (long long)128 * a.Kpad * 2 < (1ll << 31);
}

bool conv_b3_win_ok(const ConvArgs &a) {
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.dil_h != 1 || a.dil_w != 1 || a.pad_t != 1 || a.pad_l != 1 || a.Ho != a.H ||
        a.Wo != a.W || a.H < 2 || a.W < 2 || (a.Cin & 31) || a.split_k != 1 || b3_win_pieces(a) == 0)
        return false;
    return (long long)(512 + 2 * a.W) * a.x_ld * 2 < (1ll << 31) && (long long)128 * a.Kpad * 2 < (1ll << 31);
}

int conv_b3_patch_launch(int tile, const ConvArgs &a, hipStream_t st) {
    if (tile == 56 || tile == 53) {
        if (!conv_b3_win_ok(a))
            return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3, window kernel): needs a 3x3 / stride 1 / pad 1 conv with W <= 86, "
                                                       "Cin % 32 == 0, no split-K");
        if (tile == 53) return launch_b3_win<64, 4, 1, true>(a, st);   // single-phase, one window buffer, two blocks per CU
        return launch_b3_win<128, 4, 2>(a, st);
    }
    if (!conv_b3_patch_ok(a))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (bf16x3, patch kernel): needs a 3x3 / stride 1 / pad 1 conv on images whose "
                                                   "height and width are multiples of 16, Cin % 32 == 0, no split-K");
    switch (tile) {
        case 59: return launch_b3_patch<64, 4, 1, true>(a, st);      // single-phase, one window buffer, two blocks per CU
        case 58: return launch_b3_patch<128, 4, 2>(a, st);
        default: return cer_set_error(CER_ERR_INVALID_ARG, "conv2d (bf16x3, patch kernel): unknown tile id");
    }
}

}  // namespace cer
