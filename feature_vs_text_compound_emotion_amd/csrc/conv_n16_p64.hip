// conv_n16_p64.hip -- PERSISTENT 16x16-patch kernel for the K-short narrow layers (3x3 / stride 1 / pad 1, Cin == 64, Cout a
// multiple of 64: the 64 -> 64 @224x224 and 64 -> 128 @112x112 layers of the 224x224 pyramid).
//
// Why (in-kernel stamps of conv_n16_patch_kernel<64> on 64 -> 64 @224x224, tools/exp_stamp.py, profiles/round3_stamps_tile71.txt):
// a block of the one-patch-per-block kernel lives 13.7 us for 2.2 us of matrix work -- 4 us of prologue (kernel arguments,
// ~600 instructions of address arithmetic, 15 DMA issues, the window's flight), a K loop that shares its SIMDs with the second
// block of the CU running the SAME phase (the two blocks of a CU stay within a few us of each other), then the output stores and
// the exit.  K = 576 is too short to amortise any of it, and two blocks per CU do not overlap what they both do at once.
//
// Here ONE block per CU (4 waves, one per SIMD) walks over its share of the patches:
//   * the 64 x 576 weight panel of its cout tile is loaded ONCE into LDS (72 KiB, the nine tap slices of the ring kernels);
//   * two window buffers (2 x 41 KiB): the window of patch i + 1 is fetched by LDS-DMA during the first five taps of patch i;
//   * no weight traffic, no ring, no barrier inside the K loop -- one block barrier per patch (window i + 1 has landed,
//     everyone is done with window i);
//   * two accumulator sets: the epilogue of patch i - 1 (direct stores from the accumulators, conv_common.h) is issued between
//     the MFMAs of the last four taps of patch i, so the matrix pipes do not idle behind it; the batch statistics are ONE running
//     sum per lane over all of the block's patches (the consumer sums the per-patch rows: all but one of the block's are zeros); its stores are the wave's YOUNGEST vector-memory operations at the end of the patch, so the counted
//     vmcnt(stores) that waits for window i + 1 does not wait for them;
//   * blocks of one XCD walk neighbouring patches at the same time (halo rows shared in that XCD's L2).
// Launches whose epilogue is not one of the specialised modes (conv_common.h: epi_mode) go to conv_n16_patch_kernel<64>.
#include "conv_n16.h"

namespace cer {

namespace {
constexpr int P64_BN = 64, P64_NW = 4, P64_XPIECES = 41, P64_XBYTES = P64_XPIECES * 1024;
constexpr int P64_WSLICE = P64_BN * 128, P64_WOFF = 2 * P64_XBYTES, P64_SINK = P64_WOFF + 9 * P64_WSLICE, P64_RED = P64_SINK + 1024;
constexpr int P64_LDS = P64_RED + P64_NW * 2 * P64_BN * 4;       // 160 768 bytes of the 163 840
}  // namespace

template <bool F16>
__global__ __launch_bounds__(256, 1) void conv_n16_p64_kernel(ConvArgs p, PatchGeo geo) {
    constexpr int BN = P64_BN, NW = P64_NW, WP = 4, TP = 4, TC = 4, PH = 16, PWD = 16, WW = 18, WROWS = 18 * 18;
    constexpr int XPIECES = P64_XPIECES, XPW = (XPIECES + NW - 1) / NW, XBYTES = P64_XBYTES;
    constexpr int WOFF = P64_WOFF, WSLICE = P64_WSLICE, SINK = P64_SINK, RED = P64_RED;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NARROW = F16 ? CER_STORE_F16 : CER_STORE_BF16;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_p64[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_p64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave;
    const int kg = lane >> 4, l15 = lane & 15;
    const int prow = lane >> 3, slot = lane & 7;

    // ---- this block's patches: XCD x owns a contiguous range of the patches; its blocks (k = 0 .. G/8 - 1) take cout tile
    // k % tiles_n and every (G / 8 / tiles_n)-th patch of the range from k / tiles_n (the launcher makes G a multiple of 8 tiles_n)
    const int xcd = blockIdx.x & 7, kblk = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int kq = (int)fdiv((unsigned)kblk, geo.tiles_n), tile_n = kblk - kq * p.tiles_n;
    const int pstride = nbx / p.tiles_n;
    const int q8 = p.tiles_m >> 3, r8 = p.tiles_m & 7;
    const int plo = xcd * q8 + (xcd < r8 ? xcd : r8), pcnt = q8 + (xcd < r8 ? 1 : 0);
    const int c0 = tile_n * BN;
    if (kq >= pcnt) return;                                   // (whole block: no barrier has been reached)

    // ---- the weight panel of the cout tile, once: nine slices [64 couts][64 channels], rows dealt as epi_cout_of_row says ----
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(p.w_hi) + (size_t)c0 * p.Kpad * 2), 0, (int)OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int idx = wave + NW * i, tap = idx >> 3, pc = idx & 7;
            const int row = pc * 8 + prow;
            const int grow = epi_cout_of_row(row);
            const unsigned off = c0 + grow < p.Cout ? (unsigned)(grow * p.Kpad * 2 + tap * 128) + (unsigned)((slot ^ ((row >> 1) & 7)) << 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (n_lds_ptr_t)(smem + WOFF + tap * WSLICE + pc * 1024), 16, (int)off, 0, 0, 0);
        }
    }

    // ---- per-patch window addresses (32-bit: conv_n16_p64_ok bounds an image by 2^31 bytes).  The lane's pixel (wy, wx) of every
    // piece and its byte offset from the window's first pixel do not depend on the patch: an interior patch adds one scalar ----
    const int pitch = p.x_ld * 2;
    int wyx[XPW];                                              // wy * 32 + wx, or -1 for the lanes past the 324 window pixels
    unsigned xrel[XPW];                                        // (wy * W + wx) * pitch + the lane's swizzled 16-byte slot
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int qp = wave + NW * i;
        const int row = qp * 8 + prow;
        const int wy = row / WW, wx = row - wy * WW;
        const bool live = qp < XPIECES && row < WROWS;
        wyx[i] = live ? wy * 32 + wx : -1;
        xrel[i] = live ? (unsigned)((wy * p.W + wx) * pitch) + (unsigned)((slot ^ patch_f(wx)) << 4) : OOB;
    }
    struct Patch { int n, py, px, id; };
    auto patch_of = [&](int k) {
        Patch t;
        t.id = plo + k;
        const int prw = (int)fdiv((unsigned)t.id, geo.pxn);
        t.px = t.id - prw * (int)geo.pxn.d;
        t.n = (int)fdiv((unsigned)prw, geo.pyn);
        t.py = prw - t.n * (int)geo.pyn.d;
        return t;
    };
    auto window_offsets = [&](const Patch &t, unsigned (&xo)[XPW]) {
        const int iy0 = t.py * PH - 1, ix0 = t.px * PWD - 1;
        const int base = (iy0 * p.W + ix0) * pitch;            // (negative for the first patch of an image: lanes that use it are in range)
        const bool interior = t.py > 0 && t.py < (int)geo.pyn.d - 1 && t.px > 0 && t.px < (int)geo.pxn.d - 1;
        if (interior) {
#pragma unroll
            for (int i = 0; i < XPW; ++i) xo[i] = wyx[i] >= 0 ? (unsigned)(base + (int)xrel[i]) : OOB;
        } else {
#pragma unroll
            for (int i = 0; i < XPW; ++i) {
                const int iy = iy0 + (wyx[i] >> 5), ix = ix0 + (wyx[i] & 31);
                const bool inb = wyx[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                xo[i] = inb ? (unsigned)(base + (int)xrel[i]) : OOB;
            }
        }
    };
    auto issue_x = [&](int i, const Patch &t, const unsigned (&xo)[XPW], int buf) {
        const bool real = wave + NW * i < XPIECES;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(p.x_hi) + (size_t)t.n * p.H * p.W * p.x_ld * 2), 0, (int)OOB, 0x00020000);
        unsigned char *dst = real ? smem + buf * XBYTES + (wave + NW * i) * 1024 : smem + SINK;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (n_lds_ptr_t)dst, 16, (int)(real ? xo[i] : OOB), 0, 0, 0);
    };

    // ---- fragment address bases (conv_n16_patch_kernel's images) ----
    const int arow = WOFF + l15 * 128 + ((kg ^ ((l15 >> 1) & 7)) << 4);
    int bcol[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) bcol[kw] = (wp * WW + kw + l15) * 128 + ((kg ^ patch_f(kw + l15)) << 4);

    const int emode = epi_mode(p);
    const int cw = c0;                                         // the wave's first cout (one cout group of 64)
    float *red = reinterpret_cast<float *>(smem + RED);        // [WP][2][BN]

    // ---- epilogue pieces of one patch: chunk (j, b) = cout pair j x output row b * 4 + wp; statistics; final sum ----
    // (Cout is a multiple of 64: every lane's 8 couts exist.  STATS: only the raw-output launches carry batch statistics)
    float s1[2][8], s2[2][8];
    auto epi_chunk = [&](auto MODE_, auto STATS_, auto J, auto B, const n_f32x4 (&acc)[TC][TP], const Patch &t, const float (&aa)[8],
                         const float (&bb)[8]) {
        constexpr int MODE = decltype(MODE_)::v, j = decltype(J)::v, b = decltype(B)::v;
        constexpr bool STATS = decltype(STATS_)::v;
        const int c = cw + j * 32 + kg * 8;
        const int gy = t.py * PH + b * WP + wp, gx = t.px * PWD + l15;             // gy is wave-uniform
        const int m = (t.n * p.H + gy) * p.W + gx;
        int cs = 4;
        if constexpr (MODE == EPI_B9_PRELU_N16) {
            const int ry = gy == 0 ? 0 : (gy == p.H - 1 ? 2 : 1), rx = gx == 0 ? 0 : (gx == p.W - 1 ? 2 : 1);
            cs = 3 * ry + rx;
        }
        const float v[8] = {acc[2 * j][b][0], acc[2 * j][b][1], acc[2 * j][b][2], acc[2 * j][b][3],
                            acc[2 * j + 1][b][0], acc[2 * j + 1][b][1], acc[2 * j + 1][b][2], acc[2 * j + 1][b][3]};
        if constexpr (STATS) {   // the BLOCK's running sums over all of its patches (see the walk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[j][e] += v[e];
                s2[j][e] += v[e] * v[e];
            }
        }
        epi_direct8<MODE, NARROW>(p, aa, bb, (size_t)(p.y_s2d ? s2d_row(m, gy, gx, p.W) : m), c, cs, v);
    };
    auto stats_to_lds = [&](int par) {                         // the wave's totals over all its pixels -> red[par][wp]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[j][e] = row16_sum(s1[j][e]);
                s2[j][e] = row16_sum(s2[j][e]);
            }
        if (l15 == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float *d1 = red + ((par * WP + wp) * 2 + 0) * BN + j * 32 + kg * 8, *d2 = d1 + BN;
                *reinterpret_cast<float4 *>(d1) = make_float4(s1[j][0], s1[j][1], s1[j][2], s1[j][3]);
                *reinterpret_cast<float4 *>(d1 + 4) = make_float4(s1[j][4], s1[j][5], s1[j][6], s1[j][7]);
                *reinterpret_cast<float4 *>(d2) = make_float4(s2[j][0], s2[j][1], s2[j][2], s2[j][3]);
                *reinterpret_cast<float4 *>(d2 + 4) = make_float4(s2[j][4], s2[j][5], s2[j][6], s2[j][7]);
            }
        }
    };
    auto stats_final = [&](int par, const Patch &t) {          // after the block barrier that follows stats_to_lds(par)
        if (tid < 2 * BN) {
            const int st = tid >> 6, c = tid & 63;
            if (c0 + c < p.Cout) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < WP; ++w) a += red[((par * WP + w) * 2 + st) * BN + c];
                p.stats[((size_t)t.id * 2 + st) * p.Cout + c0 + c] = a;
            }
        }
    };

    // ---- one patch: nine taps on accC; the window of `nxt` is fetched during taps 0..4; the epilogue of `prv` (accP) rides on
    // taps 5..8.  HAS_PREV / has_next: the first / last patch of the block ----
    auto body = [&](auto MODE_, auto STATS_, auto HAS_PREV_, n_f32x4 (&accC)[TC][TP], n_f32x4 (&accP)[TC][TP], int cur, bool has_next,
                    const Patch &nxt, const Patch &prv, int it, const float (&aa)[2][8], const float (&bb)[2][8]) {
        constexpr int MODE = decltype(MODE_)::v;
        constexpr bool HAS_PREV = decltype(HAS_PREV_)::v, STATS = decltype(STATS_)::v;
        unsigned xo[XPW];
        if (has_next) window_offsets(nxt, xo);
        if constexpr (HAS_PREV && STATS) {
            // The statistics consumer sums the [patches][2][Cout] rows: the block keeps ONE running sum per lane over all of its
            // patches (no per-patch DPP reduction, LDS round trip or final sum), writes zeros into the rows of all its patches but
            // the last, and its totals into that one.  (Issued here: older than the window DMA below, see the counted vmcnt.)
            if (tid < 2 * BN) p.stats[((size_t)prv.id * 2 + (tid >> 6)) * p.Cout + c0 + (tid & 63)] = 0.f;
        }
        const unsigned char *Xb = smem + cur * XBYTES;
        // fragments of a HALF tap (32 of the 64 channels), read one half tap ahead: 8 reads in flight behind 16 MFMAs
        n_u32x4 af[2][TC], bf[2][TP];
        auto read_half = [&](auto Hh, int set) {
            constexpr int h = decltype(Hh)::v, tap = h / 2, kk = h % 2, kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int a = 0; a < TC; ++a)
                af[set][a] = *reinterpret_cast<const n_u32x4 *>(smem + tap * WSLICE + ((arow + a * 16 * 128) ^ (kk << 6)));
#pragma unroll
            for (int b = 0; b < TP; ++b)
                bf[set][b] = *reinterpret_cast<const n_u32x4 *>(Xb + kh * (WW * 128) + ((bcol[kw] + b * WP * WW * 128) ^ (kk << 6)));
        };
        read_half(IdxC<0>{}, 0);
        static_for<18>([&](auto Hh) {
            constexpr int h = decltype(Hh)::v, tap = h / 2, kk = h % 2, set = h & 1;
            if constexpr (h < 17) read_half(IdxC<h + 1>{}, set ^ 1);
#pragma unroll
            for (int b = 0; b < TP; ++b)
#pragma unroll
                for (int a = 0; a < TC; ++a) {
                    if constexpr (h == 0) {
                        const n_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        accC[a][b] = mfma_n16<F16>(af[set][a], bf[set][b], zero);
                    } else {
                        accC[a][b] = mfma_n16<F16>(af[set][a], bf[set][b], accC[a][b]);
                    }
                }
            // the next window: 3 + 2 + 2 + 2 + 2 pieces during taps 0..4 (an absent piece goes to the sink: constant counts)
            if constexpr (tap < 5 && kk == 0) {
                constexpr int first = tap == 0 ? 0 : 2 * tap + 1, cnt = tap == 0 ? 3 : 2;
                if (has_next) {
#pragma unroll
                    for (int i = first; i < first + cnt; ++i) issue_x(i, nxt, xo, cur ^ 1);
                }
            }
            if constexpr (h == 9) __builtin_amdgcn_sched_barrier(0);   // every DMA above, every store below: the counted vmcnt
            if constexpr (HAS_PREV && h >= 10) {
                constexpr int ch = h - 10;                              // one of the eight chunks per half tap: j = ch / 4, b = ch % 4
                epi_chunk(MODE_, STATS_, IdxC<ch / 4>{}, IdxC<ch % 4>{}, accP, prv, aa[ch / 4], bb[ch / 4]);
            }
            // issue order: 4 MFMAs, then two of the next half tap's eight fragment reads
            if constexpr (h < 17) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
        });
        // window `nxt` has landed (the stores of the epilogue above are younger than every DMA piece and may stay in flight:
        // one store per chunk in the 16-bit modes, two in the fp32 mode, loads besides in the residual mode -> drain everything)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HAS_PREV && (MODE == EPI_RAW_N16 || MODE == EPI_B9_PRELU_N16)) {
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        } else if constexpr (HAS_PREV && MODE == EPI_RAW_F32) {
            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- the walk ----
    n_f32x4 accA[TC][TP], accB[TC][TP];
    auto walk = [&](auto MODE_, auto STATS_) {
        constexpr int MODE = decltype(MODE_)::v;
        constexpr bool STATS = decltype(STATS_)::v;
        float aa[2][8], bb[2][8];                               // PReLU slopes / bias of the lane's 16 couts: once per block
        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[j][e] = s2[j][e] = 0.f;
        }
        epi_direct_consts<MODE>(p, cw + kg * 8, aa[0], bb[0]);
        epi_direct_consts<MODE>(p, cw + 32 + kg * 8, aa[1], bb[1]);
        Patch cur = patch_of(kq), prv = cur, nxt = cur;
        {
            unsigned xo[XPW];
            window_offsets(cur, xo);
#pragma unroll
            for (int i = 0; i < XPW; ++i) issue_x(i, cur, xo, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the weight panel, the constants and the first window
        __builtin_amdgcn_s_barrier();
        int k = kq, it = 0;
        bool has_next = k + pstride < pcnt;
        if (has_next) nxt = patch_of(k + pstride);
        body(MODE_, STATS_, IdxC<0>{}, accA, accB, 0, has_next, nxt, prv, it, aa, bb);
        while (has_next) {
            prv = cur; cur = nxt; k += pstride; ++it;
            has_next = k + pstride < pcnt;
            if (has_next) nxt = patch_of(k + pstride);
            body(MODE_, STATS_, IdxC<1>{}, accB, accA, 1, has_next, nxt, prv, it, aa, bb);
            if (!has_next) break;
            prv = cur; cur = nxt; k += pstride; ++it;
            has_next = k + pstride < pcnt;
            if (has_next) nxt = patch_of(k + pstride);
            body(MODE_, STATS_, IdxC<1>{}, accA, accB, 0, has_next, nxt, prv, it, aa, bb);
        }
        // the last patch's epilogue: its accumulators are accA after an even number of patches before it, accB otherwise
        const bool lastA = (it & 1) == 0;
        static_for<8>([&](auto CH) {
            constexpr int ch = decltype(CH)::v;
            if (lastA) epi_chunk(MODE_, STATS_, IdxC<ch / 4>{}, IdxC<ch % 4>{}, accA, cur, aa[ch / 4], bb[ch / 4]);
            else epi_chunk(MODE_, STATS_, IdxC<ch / 4>{}, IdxC<ch % 4>{}, accB, cur, aa[ch / 4], bb[ch / 4]);
        });
        if constexpr (STATS) {   // the block's totals into the row of its last patch
            stats_to_lds(0);
            __syncthreads();
            stats_final(0, cur);
        }
    };
    // (conv_n16_p64_ok: statistics come with the raw-output modes only)
    if (emode == EPI_RAW_N16) {
        if (p.stats) walk(IdxC<EPI_RAW_N16>{}, IdxC<1>{});
        else walk(IdxC<EPI_RAW_N16>{}, IdxC<0>{});
    } else if (emode == EPI_RAW_F32) {
        if (p.stats) walk(IdxC<EPI_RAW_F32>{}, IdxC<1>{});
        else walk(IdxC<EPI_RAW_F32>{}, IdxC<0>{});
    } else if (emode == EPI_B9_PRELU_N16) {
        walk(IdxC<EPI_B9_PRELU_N16>{}, IdxC<0>{});
    } else if (emode == EPI_BIAS_RES_N16) {
        walk(IdxC<EPI_BIAS_RES_N16>{}, IdxC<0>{});
    }
}

// The persistent kernel takes a launch when its epilogue is one of the direct 16-bit / fp32 modes, the geometry is the patch
// kernels', Cin == 64, whole cout tiles of 64, and there are enough patches for every block to amortise its weight panel.
bool conv_n16_p64_ok(const ConvArgs &a) {
    if (!conv_n16_patch_ok(a, 71) || (a.Cout & 63) || a.tiles_n > 4 || (a.tiles_n & (a.tiles_n - 1))) return false;
    const int mode = epi_mode(a);
    if (mode == EPI_RAW_F32 || mode == EPI_RAW_N16) return true;
    return (mode == EPI_B9_PRELU_N16 || mode == EPI_BIAS_RES_N16) && !a.stats;
}

int conv_n16_p64_launch(const ConvArgs &a, hipStream_t st) {
    if (!conv_n16_p64_ok(a))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow, persistent patch kernel): needs a 3x3 / stride 1 / pad 1 conv on images whose "
                                                   "height and width are multiples of 16, Cin == 64, Cout % 64 == 0 (64, 128 or 256) "
                                                   "and one of the specialised epilogues");
    int dev = 0, cus = 0;
    CER_HIP_CHECK(hipGetDevice(&dev));
    CER_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int unit = 8 * a.tiles_n;
    const int grid = cus / unit * unit;                     // one block per CU, a multiple of 8 XCDs x cout tiles
    if (grid <= 0) return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d (narrow, persistent patch kernel): fewer CUs than 8 x cout tiles");
    const PatchGeo geo{make_fastdiv((unsigned)a.tiles_n), make_fastdiv((unsigned)(a.W / 16)), make_fastdiv((unsigned)(a.H / 16))};
    if (a.narrow == CER_STORE_F16) {
        auto k = conv_n16_p64_kernel<true>;
        CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, P64_LDS));
        CER_LAUNCH(k, dim3(grid), dim3(256), P64_LDS, st, a, geo);
    } else {
        auto k = conv_n16_p64_kernel<false>;
        CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, P64_LDS));
        CER_LAUNCH(k, dim3(grid), dim3(256), P64_LDS, st, a, geo);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

}  // namespace cer
