// conv_n16_winp.hip -- PERSISTENT form of the ping-pong window kernel (conv_n16_win_kernel<128, 4, 2, ., true, 2>, conv_n16_patch.hip):
// 3x3 / stride 1 / pad 1 on narrow operands, any image size with W <= 86, 256 consecutive pixels x 128 couts per tile.
//
// Why (phase stamps of the one-tile-per-block kernel on 256 -> 256 @56x56, tools/exp_stamp_win.py, profiles/round3_stamps_win.txt):
// inside its K loop that kernel keeps the matrix pipes ~0.8 busy (36 steps of ~0.7 us), but a block lives 32 us of which
// 4.6 .. 6.5 us are its prologue (arguments, addresses, the first window's flight, the phase offset of the second group) and
// ~2 us its epilogue -- and with 143 KiB of LDS there is ONE block per CU, so nothing runs under them: MFMA busy 0.60 over
// the launch (profiles/round3_mfma_util_bf16_hw224_L64.txt).
//
// Here one block per CU walks over its tiles (all of one cout tile: the weight panel stays the same).  The DMA slots that the
// last chunk of a tile used to fill with dummy pieces -- "the next chunk's window", "the slices of steps 37 and 38" -- carry the
// NEXT TILE's first window and its slices 0 and 1, so the counted-vmcnt step pipeline runs on across the tile boundary and the
// next tile starts with everything resident.  Between two tiles: the phase groups re-align (one barrier), the accumulators
// go out through the direct epilogue (conv_common.h; statistics scratch in the idle ring slot), the groups take their offset
// again.  Launches with a generic epilogue stay on conv_n16_win_kernel.
#include "conv_n16.h"

namespace cer {

template <int BN, bool F16>
__global__ __launch_bounds__(512, 2) void conv_n16_winp_kernel(ConvArgs p, int NP, WinGeo geo) {
    constexpr int WP = 4, WC = 2, NW = WP * WC, BM = 256;
    constexpr int NPMAX = 54;                                       // 8-row window pieces per buffer the LDS can hold twice (W <= 86)
    constexpr int XPW = (NPMAX + NW - 1) / NW;
    constexpr int WSLICE = BN * 128, RING = 3;
    constexpr int WQ = BN / (8 * NW);
    constexpr int TP = BM / (16 * WP), TC = BN / (16 * WC);
    static_assert(XPW <= 9 && TC % 2 == 0 && (BN / 16) % 4 == 0 && BN % (8 * NW) == 0, "geometry");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NGRP = 2 * TP;
    constexpr int NARROW = F16 ? CER_STORE_F16 : CER_STORE_BF16;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_winp[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_winp);
    const int XBYTES = NP * 1024, WOFF = 2 * XBYTES, SINK = WOFF + RING * WSLICE;
    const int ZROW = NP * 8 - 1;                                    // past the rows the taps address: always zero-filled

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave % WP, wc = wave / WP;
    const int kg = lane >> 4, l15 = lane & 15;
    const int prow = lane >> 3, slot = lane & 7;

    // ---- this block's tiles: XCD x owns a contiguous range of the pixel tiles; its blocks (k = 0 .. G/8 - 1) take cout tile
    // k % tiles_n and every (G / 8 / tiles_n)-th tile of the range from k / tiles_n (the launcher makes G a multiple of 8 tiles_n)
    const int xcd = blockIdx.x & 7, kblk = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int kq = (int)fdiv((unsigned)kblk, geo.tiles_n), tile_n = kblk - kq * p.tiles_n;
    const int tstride = nbx / p.tiles_n;
    const int q8 = p.tiles_m >> 3, r8 = p.tiles_m & 7;
    const int tlo = xcd * q8 + (xcd < r8 ? xcd : r8), tcnt = q8 + (xcd < r8 ? 1 : 0);
    if (kq >= tcnt) return;                                         // (whole block: no barrier has been reached)
    const int c0 = tile_n * BN;
    const int cin_steps = p.cin_steps;
    const int rows_needed = BM + 2 * p.W + 2;

    // ---- weights: the same panel for every tile of the block ----
    unsigned w_off[WQ];
    int w_piece[WQ];
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
        w_piece[i] = (wave >> 2) * (BN / 16) + (wave & 3) + 4 * i;    // the group's own cout half dealt to its four waves
        const int row = w_piece[i] * 8 + prow;                                          // LDS row of the slice
        const int grow = row / (TC * 16) * (TC * 16) + epi_cout_of_row(row % (TC * 16));  // the cout it holds (conv_common.h)
        w_off[i] = c0 + grow < p.Cout ? (unsigned)(grow * p.Kpad * 2) + (unsigned)((slot ^ ((row >> 1) & 7)) << 4) : OOB;
    }
    const char *wpanel = reinterpret_cast<const char *>(p.w_hi) + (size_t)c0 * p.Kpad * 2;
    auto issue_w = [&](bool real, int cc, int tap, int ring) {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(wpanel) + ((size_t)tap * p.Cin + (size_t)cc * 64) * 2, 0, (int)OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            unsigned char *dst = real ? smem + WOFF + ring * WSLICE + w_piece[i] * 1024 : smem + SINK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (n_lds_ptr_t)dst, 16, (int)(real ? w_off[i] : OOB), 0, 0, 0);
        }
    };

    // ---- windows: piece i of a tile = rows 8 (wave + 8 i) .. + 7 of [m0 - W - 1, m0 + 256 + W]; the byte offsets from the window's
    // first pixel are the same for every tile, only the range check against the tensor's ends depends on it ----
    auto window_offsets = [&](int m0, unsigned (&xo)[XPW]) {
        const int wstart = m0 - p.W - 1;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int q = wave + NW * i;
            const int row = q * 8 + prow;
            const int pix = wstart + row;
            const bool inb = q < NP && row < rows_needed && pix >= 0 && pix < p.M;
            xo[i] = inb ? (unsigned)(row * p.x_ld * 2) + (unsigned)((slot ^ (row & 6)) << 4) : OOB;
        }
    };
    // piece i of channel chunk cc of the tile at m0 into window buffer `buf` (an absent piece: zeros into the sink -- the counted
    // vmcnt relies on constant per-step counts)
    auto issue_x = [&](bool real, int i, int m0, const unsigned (&xo)[XPW], int cc, int buf) {
        real = real && wave + NW * i < NP;
        const char *xwin = reinterpret_cast<const char *>(p.x_hi) + (long long)(m0 - p.W - 1) * p.x_ld * 2;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xwin) + (size_t)cc * 128, 0, (int)OOB, 0x00020000);
        unsigned char *dst = real ? smem + buf * XBYTES + (wave + NW * i) * 1024 : smem + SINK;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (n_lds_ptr_t)dst, 16, (int)(real ? xo[i] : OOB), 0, 0, 0);
    };

    // per pixel tile b: the lane's window row under tap (0, 0) (the same for every tile) and, per tile, the 9-bit mask of the
    // taps that stay inside its image
    int prow0[TP];
#pragma unroll
    for (int b = 0; b < TP; ++b) prow0[b] = (b * WP + wp) * 16 + l15;
    auto tap_masks = [&](int m0, unsigned (&taps)[TP]) {
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            const int m = m0 + prow0[b];
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
    };
    const int arow = (wc * TC * 16 + l15) * 128 + ((kg ^ ((l15 >> 1) & 7)) << 4);
    const int emode = epi_mode(p);

    // (one instance of the walk per epilogue mode: a single loop body that carried all six epilogues spilled 272 registers)
    auto walk = [&](auto MODE_) {
    constexpr int MODE = decltype(MODE_)::v;
    // ---- the first tile's window and slices 0 / 1 ----
    int k = kq, m0 = (tlo + k) * BM;
    unsigned x_off[XPW], taps[TP];
    window_offsets(m0, x_off);
    tap_masks(m0, taps);
#pragma unroll
    for (int i = 0; i < XPW; ++i) issue_x(true, i, m0, x_off, 0, 0);
    issue_w(true, 0, 0, 0);
    issue_w(true, 0, 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WQ) : "memory");       // the window and slice 0 (slice 1 may still be in flight)
    int par = 0;                                                    // window buffer of the current tile's chunk 0

    n_f32x4 acc[TC][TP];
    for (;;) {
        const bool has_next = k + tstride < tcnt;
        const int m0n = (tlo + k + tstride) * BM;
        unsigned x_offn[XPW];
        if (has_next) window_offsets(m0n, x_offn);
#pragma unroll
        for (int a = 0; a < TC; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
        __builtin_amdgcn_s_barrier();
        if (wc == 1) __builtin_amdgcn_s_barrier();                  // group 1 runs one phase behind group 0

        for (int cc = 0; cc < cin_steps; ++cc) {
            const int xcur = ((par + cc) & 1) * XBYTES;
            const bool last = cc == cin_steps - 1;
            static_for<9>([&](auto T) {
                constexpr int tap = decltype(T)::v, kh = tap / 3, kw = tap % 3;
                constexpr int ntap = (tap + 2) % 9, nring = (tap + 2) % 3;
                constexpr bool wraps = tap + 2 >= 9;                 // the slice of step + 2 belongs to the next chunk
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
                // ---- READ phase: every fragment of the step, then the step's DMA (slice of step + 2, a window piece) ----
                n_u32x4 af[2][TC], bf[NGRP];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int a = 0; a < TC; ++a) af[kk][a] = lda(a, kk);
#pragma unroll
                for (int g = 0; g < NGRP; ++g) bf[g] = ldb(g % TP, g / TP);
                // (past the tile's last chunk: the next tile's chunk 0 -- the same panel, its own window)
                if constexpr (wraps) issue_w(!last || has_next, last ? 0 : cc + 1, ntap, nring);
                else issue_w(true, cc, ntap, nring);
                if constexpr (tap < XPW) {
                    if (!last) issue_x(true, tap, m0, x_off, cc + 1, (par + cc + 1) & 1);
                    else issue_x(has_next, tap, m0n, x_offn, 0, (par + cc + 1) & 1);
                }
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
                constexpr int cnt = WQ + (tap < XPW ? 1 : 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cnt) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if (wc == 0) __builtin_amdgcn_s_barrier();                  // pairs with group 1's last phase boundary: the groups are aligned

        // ---- direct epilogue: accumulators -> global memory (conv_common.h); statistics scratch: ring slot 2, which the last step
        // read and nobody fills before the next tile's first READ phase ----
        {
            float s1[TC / 2][8], s2[TC / 2][8];
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
#ifndef WINP_NO_STORES
            epi_direct_stores<MODE, NARROW, TC, TP>(p, acc, c0 + wc * TC * 16, kg, epx, s1, s2);
#endif
            if (p.stats)
                epi_direct_stats<TC, WP, BN>(p, s1, s2, reinterpret_cast<float *>(smem + WOFF + 2 * WSLICE), wp, wc, kg, l15, tid, c0, (size_t)(tlo + k));
        }
        if (!has_next) break;
        // ---- the next tile: its window (chunk 0) and slices 0 / 1 were issued during the last chunk above ----
        k += tstride;
        m0 = m0n;
        par = (par + cin_steps) & 1;
#pragma unroll
        for (int i = 0; i < XPW; ++i) x_off[i] = x_offn[i];
        tap_masks(m0, taps);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // (conv_n16_winp_ok: the narrow launches' specialised modes)
    if (emode == EPI_RAW_N16) walk(IdxC<EPI_RAW_N16>{});
    else if (emode == EPI_B9_PRELU_N16) walk(IdxC<EPI_B9_PRELU_N16>{});
    else if (emode == EPI_BIAS_RES_N16) walk(IdxC<EPI_BIAS_RES_N16>{});
    else if (emode == EPI_RAW_F32) walk(IdxC<EPI_RAW_F32>{});
}

bool conv_n16_winp_ok(const ConvArgs &a) {
    if (!conv_n16_win_ok(a) || (a.Cout & 7) || a.tiles_n > 4 || (a.tiles_n & (a.tiles_n - 1))) return false;
    const int mode = epi_mode(a);
    if (mode != EPI_RAW_N16 && mode != EPI_B9_PRELU_N16 && mode != EPI_BIAS_RES_N16 && mode != EPI_RAW_F32) return false;
    const int np = (256 + 2 * a.W + 2 + 1 + 7) / 8;
    return np <= 54;
}

int conv_n16_winp_launch(const ConvArgs &a, hipStream_t st) {
    if (!conv_n16_winp_ok(a))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow, persistent window kernel): needs a 3x3 / stride 1 / pad 1 conv with W <= 86, "
                                                   "Cin % 64 == 0, Cout % 8 == 0 in at most four cout tiles (1, 2 or 4) and one of the specialised "
                                                   "epilogues");
    int dev = 0, cus = 0;
    CER_HIP_CHECK(hipGetDevice(&dev));
    CER_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int unit = 8 * a.tiles_n;
    const int grid = cus / unit * unit;
    if (grid <= 0) return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow, persistent window kernel): fewer CUs than 8 x cout tiles");
    const int np = (256 + 2 * a.W + 2 + 1 + 7) / 8;
    const size_t lds = (size_t)2 * np * 1024 + 3 * (size_t)128 * 128 + 1024;
    const WinGeo geo{make_fastdiv((unsigned)a.tiles_n), make_fastdiv((unsigned)(a.H * a.W)), make_fastdiv((unsigned)a.W)};
    if (a.narrow == CER_STORE_F16) {
        auto k = conv_n16_winp_kernel<128, true>;
        CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, dim3(grid), dim3(512), lds, st, a, np, geo);
    } else {
        auto k = conv_n16_winp_kernel<128, false>;
        CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CER_LAUNCH(k, dim3(grid), dim3(512), lds, st, a, np, geo);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

}  // namespace cer
