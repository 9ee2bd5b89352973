// conv_n16_p64.hip -- PERSISTENT 16x16-patch kernel for the K-short narrow layers (3x3 / stride 1 / pad 1, Cin == 64, Cout a
// multiple of 64: the 64 -> 64 @224x224 and 64 -> 128 @112x112 layers of the 224x224 pyramid).
//
// Why (in-kernel stamps of conv_n16_patch_kernel<64> on 64 -> 64 @224x224, tools/exp_stamp.py, profiles/round3_stamps_tile71.txt):
// a block of the one-patch-per-block kernel lives 13.7 us for 2.2 us of matrix work -- 4 us of prologue (kernel arguments,
// ~600 instructions of address arithmetic, 15 DMA issues, the window's flight), a K loop that shares its SIMDs with the second
// block of the CU running the SAME phase (the two blocks of a CU stay within a few us of each other), then the output stores and
// the exit.  K = 576 is too short to amortise any of it, and two blocks per CU do not overlap what they both do at once.
//
// Here ONE block per CU (8 waves) walks over its share of the patches:
//   * the 64 x 576 weight panel of its cout tile is loaded ONCE into LDS (72 KiB, the nine tap slices of the ring kernels):
//     no weight traffic, no ring, no barrier inside a patch's K loop;
//   * two wave groups in ping-pong over PATCHES: group g = wave / 4 owns window buffer g (41 KiB) and every second patch of
//     the block.  Time runs in phases with one block barrier each; in a phase one group is in its K phase (nine taps as 18 half
//     taps: MFMAs and fragment reads, no vector memory at all) while the other is in its E phase: the window DMA of its next
//     patch into its own, now dead, buffer -- issued FIRST --, then the direct epilogue of the patch it has just finished
//     (conv_common.h), then a counted vmcnt that waits for the window but not for the stores, which drain during the group's next
//     K phase.  A first version with ONE wave per SIMD and two accumulator sets (the previous patch's epilogue between the MFMAs
//     of the current one) lost the matrix pipe behind every vector-memory issue -- ~150 cycles each, 19 per wave and patch
//     (profiles/round3_stamps_p64.txt: 18 half taps 3.5 us bare, 4.5 with the DMA, 5.2 with DMA and stores, for 2.2 us of MFMAs;
//     3.54 ms for 1024 frames against 3.25 ms now and 4.30 ms for conv_n16_patch_kernel<64>);
//   * the batch statistics are ONE running sum per lane over all of the block's patches (the consumer sums the per-patch rows:
//     all of the block's rows but one are zeros), reduced once at the end;
//   * blocks of one XCD walk neighbouring patches at the same time (halo rows shared in that XCD's L2).
// 64 -> 64 @224x224 moves 26 GB per 1024 frames: at 3.25 ms = 4.0 TB/s of algorithmic bytes the launch is within ~20 % of what
// the BatchNorm apply pass (pure streaming) reaches on this chip -- HBM-bound, 0.47 of the narrow MFMA peak.
// Launches whose epilogue is not one of the specialised modes (conv_common.h: epi_mode) go to conv_n16_patch_kernel<64>.
#include "conv_n16.h"

namespace cer {

namespace {
constexpr int P64_BN = 64, P64_XPIECES = 41, P64_XBYTES = P64_XPIECES * 1024;
constexpr int P64_WSLICE = P64_BN * 128, P64_WOFF = 2 * P64_XBYTES, P64_SINK = P64_WOFF + 9 * P64_WSLICE;
constexpr int P64_LDS = P64_SINK + 1024;                          // 158 720 bytes of the 163 840
}  // namespace

template <bool F16>
__global__ __launch_bounds__(512, 2) void conv_n16_p64_kernel(ConvArgs p, PatchGeo geo) {
    constexpr int BN = P64_BN, NWG = 4, WP = 4, TP = 4, TC = 4, PH = 16, PWD = 16, WW = 18, WROWS = 18 * 18;
    constexpr int XPIECES = P64_XPIECES, XPW = (XPIECES + NWG - 1) / NWG, XBYTES = P64_XBYTES;
    constexpr int WOFF = P64_WOFF, WSLICE = P64_WSLICE, SINK = P64_SINK;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NARROW = F16 ? CER_STORE_F16 : CER_STORE_BF16;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_p64[];
    unsigned char *smem = reinterpret_cast<unsigned char *>(smem_p64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wp = wave & 3;
    const int kg = lane >> 4, l15 = lane & 15;
    const int prow = lane >> 3, slot = lane & 7;

    // ---- this block's patches (as in conv_n16_p64_kernel); group g takes the block's patches g, g + 2, ... ----
    const int xcd = blockIdx.x & 7, kblk = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int kq = (int)fdiv((unsigned)kblk, geo.tiles_n), tile_n = kblk - kq * p.tiles_n;
    const int pstride = nbx / p.tiles_n;
    const int q8 = p.tiles_m >> 3, r8 = p.tiles_m & 7;
    const int plo = xcd * q8 + (xcd < r8 ? xcd : r8), pcnt = q8 + (xcd < r8 ? 1 : 0);
    const int c0 = tile_n * BN;
    if (kq >= pcnt) return;                                   // (whole block: no barrier has been reached)
    const int nblk = (pcnt - kq + pstride - 1) / pstride;     // patches of the block
    const int nmine = (nblk - grp + 1) >> 1;                  // ... of this group

    // ---- the weight panel of the cout tile, once: 72 pieces over the 8 waves ----
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(p.w_hi) + (size_t)c0 * p.Kpad * 2), 0, (int)OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int idx = wave + 8 * i, tap = idx >> 3, pc = idx & 7;
            const int row = pc * 8 + prow;
            const int grow = epi_cout_of_row(row);
            const unsigned off = c0 + grow < p.Cout ? (unsigned)(grow * p.Kpad * 2 + tap * 128) + (unsigned)((slot ^ ((row >> 1) & 7)) << 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (n_lds_ptr_t)(smem + WOFF + tap * WSLICE + pc * 1024), 16, (int)off, 0, 0, 0);
        }
    }

    // ---- window addresses: piece i of the group's wave wp = rows 8 (wp + 4 i) .. + 7 of the 324 window pixels ----
    const int pitch = p.x_ld * 2;
    int wyx[XPW];
    unsigned xrel[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int qp = wp + NWG * i;
        const int row = qp * 8 + prow;
        const int wy = row / WW, wx = row - wy * WW;
        const bool live = qp < XPIECES && row < WROWS;
        wyx[i] = live ? wy * 32 + wx : -1;
        xrel[i] = live ? (unsigned)((wy * p.W + wx) * pitch) + (unsigned)((slot ^ patch_f(wx)) << 4) : OOB;
    }
    struct Patch { int n, py, px, id; };
    auto patch_of = [&](int jj) __attribute__((always_inline)) {                             // the group's jj-th patch
        Patch t;
        t.id = plo + kq + (2 * jj + grp) * pstride;
        const int prw = (int)fdiv((unsigned)t.id, geo.pxn);
        t.px = t.id - prw * (int)geo.pxn.d;
        t.n = (int)fdiv((unsigned)prw, geo.pyn);
        t.py = prw - t.n * (int)geo.pyn.d;
        return t;
    };
    auto issue_window = [&](const Patch &t) __attribute__((always_inline)) {                 // all of the wave's pieces of patch t into the group's buffer
        const int iy0 = t.py * PH - 1, ix0 = t.px * PWD - 1;
        const int base = (iy0 * p.W + ix0) * pitch;
        const bool interior = t.py > 0 && t.py < (int)geo.pyn.d - 1 && t.px > 0 && t.px < (int)geo.pxn.d - 1;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(p.x_hi) + (size_t)t.n * p.H * p.W * p.x_ld * 2), 0, (int)OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            bool inb = wyx[i] >= 0;
            if (!interior) {
                const int iy = iy0 + (wyx[i] >> 5), ix = ix0 + (wyx[i] & 31);
                inb = inb && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            }
            const unsigned off = inb ? (unsigned)(base + (int)xrel[i]) : OOB;
            const bool real = wp + NWG * i < XPIECES;
            unsigned char *dst = real ? smem + grp * XBYTES + (wp + NWG * i) * 1024 : smem + SINK;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (n_lds_ptr_t)dst, 16, (int)(real ? off : OOB), 0, 0, 0);
        }
    };

    const int arow = WOFF + l15 * 128 + ((kg ^ ((l15 >> 1) & 7)) << 4);
    int bcol[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) bcol[kw] = (wp * WW + kw + l15) * 128 + ((kg ^ patch_f(kw + l15)) << 4);
    const unsigned char *Xb = smem + grp * XBYTES;
    const int emode = epi_mode(p);

    n_f32x4 acc[TC][TP];
    float s1[2][8], s2[2][8];

    // ---- K phase: nine taps as 18 half taps, fragments read one half tap ahead; nothing but MFMAs and LDS reads ----
    auto k_phase = [&]() __attribute__((always_inline)) {
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
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // (the hints below count DS reads from here: these eight first)
        static_for<18>([&](auto Hh) {
            constexpr int h = decltype(Hh)::v, set = h & 1;
            if constexpr (h < 17) read_half(IdxC<h + 1>{}, set ^ 1);
#pragma unroll
            for (int b = 0; b < TP; ++b)
#pragma unroll
                for (int a = 0; a < TC; ++a) {
                    if constexpr (h == 0) {
                        const n_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[a][b] = mfma_n16<F16>(af[set][a], bf[set][b], zero);
                    } else {
                        acc[a][b] = mfma_n16<F16>(af[set][a], bf[set][b], acc[a][b]);
                    }
                }
            // issue order: the next half tap's eight fragment reads FIRST (a whole half tap of MFMAs covers their latency), then the
            // 16 MFMAs.  (Reads spread between the MFMA groups came out one group late: the compiler's lgkmcnt(1..4) before every
            // group of four MFMAs waited for reads it had just issued -- 5.0 us per K phase for 2.7 us of MFMAs in the stamps.)
            if constexpr (h < 17) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        });
    };

    auto walk = [&](auto MODE_, auto STATS_) __attribute__((always_inline)) {
        constexpr int MODE = decltype(MODE_)::v;
        constexpr bool STATS = decltype(STATS_)::v;
        constexpr int NST = MODE == EPI_RAW_F32 ? 16 : 8;       // stores per wave and patch (whole cout tiles)
        float aa[2][8], bb[2][8];
        epi_direct_consts<MODE>(p, c0 + kg * 8, aa[0], bb[0]);
        epi_direct_consts<MODE>(p, c0 + 32 + kg * 8, aa[1], bb[1]);
        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[j][e] = s2[j][e] = 0.f;
        }
        // E phase of the group's patch t: its accumulators out; `nxt`: the group's next patch (window DMA first, stores after)
        const int last_id = plo + kq + (nblk - 1) * pstride;  // the block's last patch: its statistics row takes the block's totals
        auto e_phase = [&](const Patch &t, bool has_next, const Patch &nxt) __attribute__((always_inline)) {
            if (has_next) issue_window(nxt);
            __builtin_amdgcn_sched_barrier(0);                  // every DMA piece above, every store below (the counted vmcnt)
            if constexpr (STATS) {
                // one running sum per lane over all of the group's patches; zero rows for all its patches but the last (the totals
                // of the block go into one row at the end: the consumer sums the rows)
                if (t.id != last_id && (tid & 255) < 2 * BN)
                    p.stats[((size_t)t.id * 2 + ((tid & 255) >> 6)) * p.Cout + c0 + (tid & 63)] = 0.f;
            }
            static_for<8>([&](auto CH) {
                constexpr int ch = decltype(CH)::v, j = ch / 4, b = ch % 4;
                const int c = c0 + j * 32 + kg * 8;
                const int gy = t.py * PH + b * WP + wp, gx = t.px * PWD + l15;
                const int m = (t.n * p.H + gy) * p.W + gx;
                int cs = 4;
                if constexpr (MODE == EPI_B9_PRELU_N16) {
                    const int ry = gy == 0 ? 0 : (gy == p.H - 1 ? 2 : 1), rx = gx == 0 ? 0 : (gx == p.W - 1 ? 2 : 1);
                    cs = 3 * ry + rx;
                }
                const float v[8] = {acc[2 * j][b][0], acc[2 * j][b][1], acc[2 * j][b][2], acc[2 * j][b][3],
                                    acc[2 * j + 1][b][0], acc[2 * j + 1][b][1], acc[2 * j + 1][b][2], acc[2 * j + 1][b][3]};
                if constexpr (STATS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s1[j][e] += v[e];
                        s2[j][e] += v[e] * v[e];
                    }
                }
                const size_t orow = (size_t)(p.y_s2d ? s2d_row(m, gy, gx, p.W) : m);
                if constexpr (MODE == EPI_B9_PRELU_N16) {   // (border rows are loaded only in waves that hold a border pixel: conv_common.h)
                    if (__ballot(cs != 4) != 0ull) epi_direct8<MODE, NARROW, true>(p, aa[j], bb[j], orow, c, cs, v);
                    else epi_direct8<MODE, NARROW, false>(p, aa[j], bb[j], orow, c, 4, v);
                } else {
                    epi_direct8<MODE, NARROW, false>(p, aa[j], bb[j], orow, c, cs, v);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            // the next window has landed; at least NST younger operations (this phase's stores) may stay in flight
            // (more operations than NST -- the zero row, the residual loads, a border pixel's bias row -- only make the wait stricter)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        };

        // ---- prologue: both groups fetch their first window; the weight panel ----
        Patch cur = patch_of(0), nxt = cur;
        if (nmine > 0) issue_window(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phases: group g computes in the phases ph = g (mod 2), its E phase follows; 2 * ceil(nblk / 2) + 1 phases, one barrier each
        const int nph = 2 * ((nblk + 1) >> 1) + 1;
        for (int ph = 0; ph < nph; ++ph) {
            const bool kph = ((ph ^ grp) & 1) == 0;
            if (kph) {
                const int jj = (ph - grp) >> 1;                 // (group 1: phases 1, 3, ...)
                if (ph >= grp && jj < nmine) k_phase();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                const int jj = (ph - 1 - grp) >> 1;             // the patch computed in the previous phase
                if (ph - 1 >= grp && jj < nmine) {
                    const bool has_next = jj + 1 < nmine;
                    if (has_next) nxt = patch_of(jj + 1);
                    e_phase(cur, has_next, nxt);
                    cur = nxt;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (STATS) {
            // the block's totals (8 waves: [8][2][64] floats in window buffer 0, dead by now) into the row of the block's last patch
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[j][e] = row16_sum(s1[j][e]);
                    s2[j][e] = row16_sum(s2[j][e]);
                }
            float *red = reinterpret_cast<float *>(smem);
            if (l15 == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float *d1 = red + (wave * 2 + 0) * BN + j * 32 + kg * 8, *d2 = d1 + BN;
                    *reinterpret_cast<float4 *>(d1) = make_float4(s1[j][0], s1[j][1], s1[j][2], s1[j][3]);
                    *reinterpret_cast<float4 *>(d1 + 4) = make_float4(s1[j][4], s1[j][5], s1[j][6], s1[j][7]);
                    *reinterpret_cast<float4 *>(d2) = make_float4(s2[j][0], s2[j][1], s2[j][2], s2[j][3]);
                    *reinterpret_cast<float4 *>(d2 + 4) = make_float4(s2[j][4], s2[j][5], s2[j][6], s2[j][7]);
                }
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const int st = tid >> 6, c = tid & 63;
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) a += red[(w * 2 + st) * BN + c];
                p.stats[((size_t)last_id * 2 + st) * p.Cout + c0 + c] = a;
            }
        }
    };
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
        CER_LAUNCH(k, dim3(grid), dim3(512), P64_LDS, st, a, geo);
    } else {
        auto k = conv_n16_p64_kernel<false>;
        CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, P64_LDS));
        CER_LAUNCH(k, dim3(grid), dim3(512), P64_LDS, st, a, geo);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

}  // namespace cer
