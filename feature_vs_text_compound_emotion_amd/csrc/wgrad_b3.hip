// wgrad_b3.hip -- weight gradient of a 2-D convolution on the bf16 matrix cores with split (hi/lo) operands ("bf16x3"):
//
//   dW[co][ci][kh][kw] = sum over output pixels r = (n, ho, wo) of  dZ[r][co] * X[n, ho*s - pad + kh, wo*s - pad + kw][ci]
//
// This is a TN GEMM whose reduction index is the PIXEL, the slow axis of both NHWC operands, while
// v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane.  The tiles therefore stay in their natural [pixel][channel]
// order in LDS (coalesced 16-byte global loads, split into hi / lo bf16 on the way in) and the operands are fetched with
// gfx950's transposing LDS read ds_read_b64_tr_b16: per 16-lane group it takes a 4-row x 16-column block of 16-bit elements
// and hands lane i column i of the four rows -- two of them are one MFMA operand (8 pixels of one channel).
//
//   * block = 4 waves, output tile 128 couts x 128 cins of ONE filter tap (channel counts that are not multiples of 128 use
//     part of it: columns past Cout / Cin are loaded as zeros); wave (wi, wj) owns 64 x 64 = 4 x 4 MFMA tiles;
//   * K step = 32 pixels: dZ tile [32][128] and the tap-shifted X tile [32][128] (zero rows where the tap leaves the
//     image), each as two bf16 planes of 8 KiB, double buffered (64 KiB: two blocks per CU);
//   * LDS image: plain 256-byte rows, 16-byte chunk ch of row r at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))):
//     conflict free for the transposed reads of a 16x16x32 operand (cdna_hip_programming.md T10, image (b));
//   * a * b = a_hi b_hi + a_hi b_lo + a_lo b_hi: three MFMAs per 16x16 tile and step (<= 2^-15 relative per product);
//   * the pixel range is split over gridDim.z blocks (the layers this serves have 25 600 .. 102 400 pixels and only
//     4 .. 16 output tiles per tap); partial results go to a workspace and a second kernel adds them in a fixed order.
//
// Replaces the fp32-MFMA kernel of wgrad.hip (64 x 64 tiles, one block per tile and tap walking ALL pixels: 144 blocks,
// 40 TFLOP/s) for the released encoder units (reference base/parameter_control.py:55-103).
#include "conv_b3.h"

namespace cer {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 as_bf16(const s16x8 v) { return __builtin_bit_cast(bf16x8, v); }
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;

struct Wgrad2dArgs {
    const float *dz, *x;                        // fp32 operands (split into hi / lo bf16 by the loader) ...
    const uint16_t *dz_hi, *dz_lo, *x_hi, *x_lo;   // ... or operands that are split tensors already (SPLIT_IN)
    float *out;          // dw, or the partial slabs [splits][Cout * Cin * KH * KW]
    int R, Cout, Cin, KHW, KW, H, W, Ho, Wo, stride, pad_t, pad_l;
    int rows_per_split;  // multiple of 32
    int tiles, splits;   // output tiles per tap, row splits
};

__device__ __forceinline__ int wg_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// TS = output tile edge: 128 (wave = 64 x 64 = 4 x 4 MFMA tiles) or 64 (wave = 32 x 32: the 64-channel layers and the stem
// without the 2-4x of zero columns a 128-wide tile would multiply)
// SPLIT_IN: dZ and X arrive as split tensors (hi / lo bf16 planes, what the released units' forward and data-gradient convs
// consume anyway): the loader copies 8 bytes per plane instead of converting -- the conversion of fp32 operands, redone by
// every (tile, tap) block, was half of the kernel's issue slots (PMC: "active" 51 % at MFMA busy 23 %).
// DMA (split operands, TS = 128, channel counts % 8 == 0): the planes already are what the LDS tiles hold, so the stage is
// filled by LDS-DMA instead of through registers -- a 1-KiB piece is 4 pixel rows x 256 bytes, lane l owns row l / 16 and LDS
// slot l % 16 and fetches the source chunk slot ^ f(row) (the swizzle of wg_off on the SOURCE address, as in the conv
// kernels); padding taps and rows past the split are out-of-range offsets = zeros.  8 DMA instructions per wave and step
// replace 16 eight-byte loads + 16 LDS stores + their index arithmetic.
template <int TS, bool SPLIT_IN, bool DMA = false>
__global__ __launch_bounds__(256, 2) void conv2d_wgrad_b3_kernel(Wgrad2dArgs p) {
    static_assert(!DMA || (SPLIT_IN && TS == 128), "the DMA loader copies split planes into full 256-byte rows");
    // KG pixel groups of 32 per stage: the 64-wide tile fills only half of a 256-byte LDS row, so a second group of 32 pixels
    // lives in the other eight 16-byte slots of the same rows (chunk + 8; the XOR swizzle is a bijection on the 16 slots of a
    // row) -- 64 pixels and twice the MFMAs per barrier in the same 32 KiB stage (round 3: the 64-channel layers ran at 0.11)
    constexpr int KG = TS == 64 ? 2 : 1;
    constexpr int KS = 32 * KG, PLANE = 32 * 256, STAGE = 4 * PLANE;  // planes per stage: dZ hi, dZ lo, X hi, X lo
    constexpr int NTW = TS / 32;                                      // 16-wide MFMA tiles per wave and side
    constexpr int LPT = TS / 32 * KG;                                 // float4 loads per thread, operand and step
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: the DMA destinations are scalar
    const int wi = wave & 1, wj = wave >> 1;
    const int kg = lane >> 4, l15 = lane & 15;
    // XCD-aware work order: the tiles x taps blocks of one row split all read the same dZ / X rows (a few MiB), so they are
    // given to ONE XCD (block b runs on XCD b % 8) and hit its L2; dealt tile-major the same rows were fetched by all eight
    // L2s (3.8 GB of requests per 256 -> 256 launch); measured gain 1-3 % -- the kernel is bound by the loader's VALU work
    const int per = p.tiles * p.KHW;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int split = (j / per) * 8 + xcd, within = j % per;
    if (split >= p.splits) return;
    const int tile = within % p.tiles, tap = within / p.tiles, kh = tap / p.KW, kw = tap - kh * p.KW;
    const int tiles_ci = (p.Cin + TS - 1) / TS;
    const int co0 = (tile / tiles_ci) * TS, ci0 = (tile % tiles_ci) * TS;
    const int r_begin = split * p.rows_per_split;
    const int r_end = min(p.R, r_begin + p.rows_per_split);

    // ---- loader: thread = 4 consecutive channels of rows lrow + RSTEP i ----
    constexpr int TPR = TS / 4;                                       // threads per row
    const int col4 = (tid % TPR) * 4, lrow = tid / TPR;               // rows lrow + (256 / TPR) i
    constexpr int RSTEP = 256 / TPR;
    const int hw = p.Ho * p.Wo;
    float4 ra[LPT], rb[LPT];            // fp32 operands: 4 floats; split operands: .x/.y = the hi plane's 8 bytes, .z/.w = lo
    // (n, ho, wo) of this thread's rows, advanced by the 32 rows of a step without divisions (the index arithmetic of the
    // loader was half of the wave's issue slots: PMC "active" 51 % at MFMA busy 23 %)
    int pn[LPT], pho[LPT], pwo[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int r = r_begin + lrow + RSTEP * i;
        pn[i] = r / hw;
        const int q = r - pn[i] * hw;
        pho[i] = q / p.Wo;
        pwo[i] = q - pho[i] * p.Wo;
    }
    const int adv_h = KS / p.Wo, adv_w = KS - adv_h * p.Wo;
    auto load_step = [&](int r0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int r = r0 + lrow + RSTEP * i;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < r_end) {
                // (channel counts are multiples of 4; columns past Cout / Cin stay zero: 64-channel layers and the 3 -> 4
                // channel stem use part of the tile)
                const int hi = pho[i] * p.stride - p.pad_t + kh, wi_ = pwo[i] * p.stride - p.pad_l + kw;
                const bool xin = ci0 + col4 < p.Cin && (unsigned)hi < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.W;
                const size_t ao = (size_t)r * p.Cout + co0 + col4, bo = (((size_t)pn[i] * p.H + hi) * p.W + wi_) * p.Cin + ci0 + col4;
                if constexpr (SPLIT_IN) {
                    if (co0 + col4 < p.Cout) {
                        const float2 h = *reinterpret_cast<const float2 *>(p.dz_hi + ao), l = *reinterpret_cast<const float2 *>(p.dz_lo + ao);
                        a = make_float4(h.x, h.y, l.x, l.y);
                    }
                    if (xin) {
                        const float2 h = *reinterpret_cast<const float2 *>(p.x_hi + bo), l = *reinterpret_cast<const float2 *>(p.x_lo + bo);
                        b = make_float4(h.x, h.y, l.x, l.y);
                    }
                } else {
                    if (co0 + col4 < p.Cout) a = *reinterpret_cast<const float4 *>(p.dz + ao);
                    if (xin) b = *reinterpret_cast<const float4 *>(p.x + bo);
                }
            }
            ra[i] = a;
            rb[i] = b;
            pwo[i] += adv_w;
            pho[i] += adv_h;
            if (pwo[i] >= p.Wo) {
                pwo[i] -= p.Wo;
                ++pho[i];
            }
            while (pho[i] >= p.Ho) {
                pho[i] -= p.Ho;
                ++pn[i];
            }
        }
    };
    auto store_step = [&](int buf) {
        unsigned char *base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int prow_ = lrow + RSTEP * i, row = prow_ & 31;      // pixel row of the step -> LDS row, slot group prow_ >> 5
            const int byte = wg_off(row, (col4 >> 3) + 8 * (prow_ >> 5)) + 8 * ((col4 >> 2) & 1);
            if constexpr (SPLIT_IN) {
                *reinterpret_cast<float2 *>(base + byte) = make_float2(ra[i].x, ra[i].y);
                *reinterpret_cast<float2 *>(base + PLANE + byte) = make_float2(ra[i].z, ra[i].w);
                *reinterpret_cast<float2 *>(base + 2 * PLANE + byte) = make_float2(rb[i].x, rb[i].y);
                *reinterpret_cast<float2 *>(base + 3 * PLANE + byte) = make_float2(rb[i].z, rb[i].w);
            } else {
                const float va[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w}, vb[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
                ushort4 h, l;
                split_bf16(va[0], h.x, l.x); split_bf16(va[1], h.y, l.y); split_bf16(va[2], h.z, l.z); split_bf16(va[3], h.w, l.w);
                *reinterpret_cast<ushort4 *>(base + byte) = h;
                *reinterpret_cast<ushort4 *>(base + PLANE + byte) = l;
                split_bf16(vb[0], h.x, l.x); split_bf16(vb[1], h.y, l.y); split_bf16(vb[2], h.z, l.z); split_bf16(vb[3], h.w, l.w);
                *reinterpret_cast<ushort4 *>(base + 2 * PLANE + byte) = h;
                *reinterpret_cast<ushort4 *>(base + 3 * PLANE + byte) = l;
            }
        }
    };

    // ---- transposed operand reads: lane 4q + p of a 16-lane group supplies row r0 + q, columns 4p .. 4p+3 of the block;
    // group kg reads rows 8 kg .. 8 kg + 3 and 8 kg + 4 .. 8 kg + 7 (the 8 pixels of its k group) ----
    const int q4 = l15 >> 2, p4 = l15 & 3;
    int tr_a[KG][NTW][2], tr_b[KG][NTW][2];   // byte offsets inside a plane: [pixel group][16-channel tile of this wave][row half]
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 8 * kg + 4 * h + q4;
                tr_a[g][t][h] = wg_off(row, 8 * g + 2 * (wi * NTW + t) + (p4 >> 1)) + 8 * (p4 & 1);
                tr_b[g][t][h] = wg_off(row, 8 * g + 2 * (wj * NTW + t) + (p4 >> 1)) + 8 * (p4 & 1);
            }
    // (fragments travel as short vectors and are re-typed at the MFMA: a lambda returning a __bf16 vector made hipcc's host
    // pass drop the kernel stub, see conv_b3_patch.hip)
    auto frag = [&](const unsigned char *plane, const int off[2]) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + off[0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + off[1]));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return v;
    };

    f32x4 acc[NTW][NTW];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    // ---- DMA loader state: pieces wave + 4 i (i = 0, 1) of each plane; (n, ho, wo) of this lane's two rows ----
    constexpr unsigned OOB = 0x80000000u;
    const int drow = lane >> 4, dch = lane & 15;
    int dn[2], dho[2], dwo[2];
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r_begin + 4 * (wave + 4 * i) + drow;
            dn[i] = r / hw;
            const int q = r - dn[i] * hw;
            dho[i] = q / p.Wo;
            dwo[i] = q - dho[i] * p.Wo;
        }
    }
    auto issue_step = [&](int r0, int buf) {
        // uniform bases: dZ rows r0 .. r0 + 31 are contiguous; X offsets are taken from the first pixel of image n0 (the rows
        // of a step span a few images at most, so they fit 32 bits)
        const int n0 = __builtin_amdgcn_readfirstlane(r0 / hw);
        const size_t zb = (size_t)r0 * p.Cout * 2, xb = (size_t)n0 * p.H * p.W * p.Cin * 2;
        const __amdgpu_buffer_rsrc_t rzh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.dz_hi)) + zb, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rzl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.dz_lo)) + zb, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.x_hi)) + xb, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.x_lo)) + xb, 0, (int)OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + 4 * i, rl = 4 * piece + drow, r = r0 + rl;
            const int ch = dch ^ (((rl & 3) << 2) | ((rl >> 2) & 3));      // source chunk (8 channels) for this LDS slot
            const int hi = dho[i] * p.stride - p.pad_t + kh, wi_ = dwo[i] * p.stride - p.pad_l + kw;
            const bool live = r < r_end;
            const bool zin = live && co0 + ch * 8 < p.Cout;
            const bool xin = live && ci0 + ch * 8 < p.Cin && (unsigned)hi < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.W;
            const unsigned ao = zin ? (unsigned)(((size_t)rl * p.Cout + co0 + ch * 8) * 2) : OOB;
            const unsigned bo = xin ? (unsigned)(((((size_t)(dn[i] - n0) * p.H + hi) * p.W + wi_) * p.Cin + ci0 + ch * 8) * 2) : OOB;
            unsigned char *dst = smem + buf * STAGE + piece * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rzh, (lds_ptr_t)dst, 16, (int)ao, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rzl, (lds_ptr_t)(dst + PLANE), 16, (int)ao, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_ptr_t)(dst + 2 * PLANE), 16, (int)bo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_ptr_t)(dst + 3 * PLANE), 16, (int)bo, 0, 0, 0);
            dwo[i] += adv_w;
            dho[i] += adv_h;
            if (dwo[i] >= p.Wo) {
                dwo[i] -= p.Wo;
                ++dho[i];
            }
            while (dho[i] >= p.Ho) {
                dho[i] -= p.Ho;
                ++dn[i];
            }
        }
    };

    const int steps = (r_end - r_begin + KS - 1) / KS;
    if (steps > 0) {
        if constexpr (DMA) {
            issue_step(r_begin, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            load_step(r_begin);
            store_step(0);
        }
    }
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if constexpr (DMA) {
            if (s + 1 < steps) issue_step(r_begin + (s + 1) * KS, buf ^ 1);   // buf ^ 1 was released by the last barrier
        } else {
            if (s + 1 < steps) load_step(r_begin + (s + 1) * KS);
        }
        const unsigned char *base = smem + buf * STAGE;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            s16x8 bh[NTW], bl[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                bh[t] = frag(base + 2 * PLANE, tr_b[g][t]);
                bl[t] = frag(base + 3 * PLANE, tr_b[g][t]);
            }
#pragma unroll
            for (int a = 0; a < NTW; ++a) {
                const s16x8 ah = frag(base, tr_a[g][a]), al = frag(base + PLANE, tr_a[g][a]);
#pragma unroll
                for (int b = 0; b < NTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(al), as_bf16(bh[b]), acc[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(ah), as_bf16(bl[b]), acc[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(ah), as_bf16(bh[b]), acc[a][b], 0, 0, 0);
            }
            // issue order (round 3): X fragments + the first dZ pair, then the reads of pair a + 1 in front of the MFMAs of pair a
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * NTW + 4, 0);
#pragma unroll
            for (int a = 0; a < NTW; ++a) {
                if (a + 1 < NTW) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NTW, 0);
            }
        }
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the next stage has landed
        } else {
            if (s + 1 < steps) store_step(buf ^ 1);
        }
        __syncthreads();
    }

    // D[i = cout][j = cin]: lane -> cin (l15), register r -> cout 4 kg + r
    float *out = p.out + (size_t)split * p.Cout * p.Cin * p.KHW;
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
            const int ci = ci0 + (wj * NTW + b) * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wi * NTW + a) * 16 + 4 * kg + r;
                if (co < p.Cout && ci < p.Cin) out[((size_t)co * p.Cin + ci) * p.KHW + tap] = acc[a][b][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide variant (round 3) for the 256- / 512-channel layers: output tile 256 couts x 256 cins of one tap, 8 waves (2 cout halves
// x 4 cin quarters, wave = 128 x 64 = 8 x 4 MFMA tiles).  The 128 x 128 kernel above is bound by what it pulls from L2 into LDS:
// 32 KiB per K step for 192 MFMAs -- 8 TB/s chip-wide at its measured 0.31 of the bf16x3 ceiling (PMC: MFMA busy 32 %, 36 % of
// the wave cycles waiting on vmcnt / the barrier), half of what gather-to-LDS loops reach on this part.  Doubling both tile
// edges moves 64 KiB per step for 768 MFMAs (half the bytes per MFMA) and does twice the MFMA work per barrier.
// LDS: 8 planes of [32 pixels][128 channels] per stage (dZ hi / lo x 2 cout halves, X hi / lo x 2 cin halves; the swizzle of
// wg_off per plane), two stages = 128 KiB, one block of 8 waves per CU.  A wave issues ONE 1-KiB piece of every plane per
// step (its 4 pixel rows), so it tracks a single (n, ho, wo) position.  Register budget at 2 waves per SIMD (<= 256): 128
// accumulators + the 4 X fragment pairs of the step (32) + one dZ fragment pair at a time (8, the compiler double-buffers).
template <int DUMMY>
__global__ __launch_bounds__(512, 1) void conv2d_wgrad_b3_wide_kernel(Wgrad2dArgs p) {
    constexpr int KS = 32, PLANE = KS * 256, STAGE = 8 * PLANE, TS = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;                     // cout half (= dZ plane), cin quarter
    const int kg = lane >> 4, l15 = lane & 15;
    const int per = p.tiles * p.KHW;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int split = (j / per) * 8 + xcd, within = j % per;
    if (split >= p.splits) return;
    const int tile = within % p.tiles, tap = within / p.tiles, kh = tap / p.KW, kw = tap - kh * p.KW;
    const int tiles_ci = (p.Cin + TS - 1) / TS;
    const int co0 = (tile / tiles_ci) * TS, ci0 = (tile % tiles_ci) * TS;
    const int r_begin = split * p.rows_per_split;
    const int r_end = min(p.R, r_begin + p.rows_per_split);
    const int hw = p.Ho * p.Wo;

    const int q4 = l15 >> 2, p4 = l15 & 3;
    int tr_a[8][2], tr_b[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * kg + 4 * h + q4;
#pragma unroll
        for (int t = 0; t < 8; ++t) tr_a[t][h] = wg_off(row, 2 * t + (p4 >> 1)) + 8 * (p4 & 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) tr_b[t][h] = wg_off(row, 2 * ((wj & 1) * 4 + t) + (p4 >> 1)) + 8 * (p4 & 1);
    }
    auto frag = [&](const unsigned char *plane, const int off[2]) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + off[0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + off[1]));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return v;
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    constexpr unsigned OOB = 0x80000000u;
    const int drow = lane >> 4, dch = lane & 15;
    const int rl = 4 * wave + drow;                               // this lane's pixel row inside a step
    const int ch = dch ^ (((rl & 3) << 2) | ((rl >> 2) & 3));     // source chunk (8 channels) for this LDS slot
    int dn, dho, dwo;
    {
        const int r = r_begin + rl;
        dn = r / hw;
        const int q = r - dn * hw;
        dho = q / p.Wo;
        dwo = q - dho * p.Wo;
    }
    const int adv_h = KS / p.Wo, adv_w = KS - adv_h * p.Wo;
    auto issue_step = [&](int r0, int buf) {
        const int n0 = __builtin_amdgcn_readfirstlane(r0 / hw);
        const size_t zb = (size_t)r0 * p.Cout * 2, xb = (size_t)n0 * p.H * p.W * p.Cin * 2;
        const __amdgpu_buffer_rsrc_t rzh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.dz_hi)) + zb, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rzl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.dz_lo)) + zb, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.x_hi)) + xb, 0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.x_lo)) + xb, 0, (int)OOB, 0x00020000);
        const int r = r0 + rl;
        const int hi = dho * p.stride - p.pad_t + kh, wi_ = dwo * p.stride - p.pad_l + kw;
        const bool live = r < r_end;
        const bool xpix = live && (unsigned)hi < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.W;
        const size_t xrow = (((size_t)(dn - n0) * p.H + hi) * p.W + wi_) * p.Cin;
        unsigned char *dst = smem_w + buf * STAGE + wave * 1024;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int co = co0 + pl * 128 + ch * 8, ci = ci0 + pl * 128 + ch * 8;
            const unsigned ao = (live && co < p.Cout) ? (unsigned)(((size_t)rl * p.Cout + co) * 2) : OOB;
            const unsigned bo = (xpix && ci < p.Cin) ? (unsigned)((xrow + ci) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rzh, (lds_ptr_t)(dst + (0 + pl) * PLANE), 16, (int)ao, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rzl, (lds_ptr_t)(dst + (2 + pl) * PLANE), 16, (int)ao, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_ptr_t)(dst + (4 + pl) * PLANE), 16, (int)bo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (lds_ptr_t)(dst + (6 + pl) * PLANE), 16, (int)bo, 0, 0, 0);
        }
        dwo += adv_w;
        dho += adv_h;
        if (dwo >= p.Wo) {
            dwo -= p.Wo;
            ++dho;
        }
        while (dho >= p.Ho) {
            dho -= p.Ho;
            ++dn;
        }
    };

    const int steps = (r_end - r_begin + KS - 1) / KS;
    if (steps > 0) {
        issue_step(r_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) issue_step(r_begin + (s + 1) * KS, buf ^ 1);
        const unsigned char *base = smem_w + buf * STAGE;
        const unsigned char *zh = base + wi * PLANE, *zl = base + (2 + wi) * PLANE;
        const unsigned char *xh = base + (4 + (wj >> 1)) * PLANE, *xl = base + (6 + (wj >> 1)) * PLANE;
        s16x8 bh[4], bl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bh[t] = frag(xh, tr_b[t]);
            bl[t] = frag(xl, tr_b[t]);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const s16x8 ah = frag(zh, tr_a[a]), al = frag(zl, tr_a[a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(al), as_bf16(bh[b]), acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(ah), as_bf16(bl[b]), acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16(ah), as_bf16(bh[b]), acc[a][b], 0, 0, 0);
        }
        // issue order: the X fragments and the first dZ pair, then the reads of pair a + 1 in front of the 12 MFMAs of pair a
        // (left to itself the scheduler hoists all 48 reads to the top: every wave of the block -- they leave the barrier
        // together -- then reads while no matrix pipe works, and multiplies while the LDS idles)
        __builtin_amdgcn_sched_group_barrier(0x100, 16 + 4, 0);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (a + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float *out = p.out + (size_t)split * p.Cout * p.Cin * p.KHW;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ci = ci0 + wj * 64 + b * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wi * 128 + a * 16 + 4 * kg + r;
                if (co < p.Cout && ci < p.Cin) out[((size_t)co * p.Cin + ci) * p.KHW + tap] = acc[a][b][r];
            }
        }
}

__global__ void wgrad_reduce_kernel(const float4 *__restrict__ part, float4 *__restrict__ dw, size_t n4, int splits) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = part[i];
    for (int k = 1; k < splits; ++k) {
        const float4 v = part[(size_t)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dw[i] = s;
}

static int wgrad_b3_splits(long long R, int tiles, int taps, int ts = 128) {
    // The pixel range is cut into `s` splits; s * tiles * taps blocks run in rounds of `slots` resident blocks (2 per CU; the
    // wide tile: 1 per CU), each block walking R / s / 32 K steps, and a reduce kernel adds the s partial slabs.  Pick the
    // s that minimises rounds x steps (+ the reduce's share): e.g. 256 -> 256 @56x56, wide tile: 9 blocks per split --
    // "enough blocks for two rounds" took s = 64 = 576 blocks = 2.25 -> THREE rounds of 1568 steps; s = 56 = 504 blocks runs
    // two rounds of 1792 (-24 %).
    const long long per = (long long)tiles * taps, slots = ts == 256 ? 256 : 512;
    const long long max_s = (R + 255) / 256;
    long long best = 1;
    double best_cost = 1e300;
    for (long long s = 1; s <= max_s && s <= 4096; ++s) {
        const long long rows = ((R + s - 1) / s + 31) / 32 * 32, steps = rows / 32;
        if (steps < 8 && s > 1) break;
        // the kernel deals the splits round-robin over the 8 XCDs (split = 8 * (j / per) + xcd, so the blocks of a split
        // share one L2): an XCD holds ceil(s / 8) * per blocks for its slots / 8 places -- s = 28 with 9 blocks per split is
        // 36 blocks on four 32-CU XCDs = TWO rounds, not the one that 252 blocks / 256 places suggests (measured: 2x)
        const long long rounds = (((s + 7) / 8) * per + slots / 8 - 1) / (slots / 8);
        // a K step of a resident block costs ~1 unit; the reduce reads s slabs of the output: ~ s * (outputs / bytes-per-step-unit)
        const double cost = (double)rounds * (steps + 24) + 0.02 * (double)s;
        if (cost < best_cost) {
            best_cost = cost;
            best = s;
        }
    }
    return (int)best;
}

}  // namespace cer

using namespace cer;

// 64 x 64 tiles when a 128-wide tile would be at most half full on either side; 256 x 256 (the wide kernel: split operands
// only) when both channel counts fill it
static int wgrad_b3_tile(int Cout, int Cin, bool split_in = false) {
    if (split_in && Cout % 256 == 0 && Cin % 256 == 0) return 256;
    return (Cout <= 64 || Cin <= 64) ? 64 : 128;
}

extern "C" size_t cer_conv2d_wgrad_b3_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW) {
    if (N <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 0;
    // (the split-operand entry point may pick the wide tile, which needs fewer splits: size for the larger of the two)
    size_t need = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int ts = wgrad_b3_tile(Cout, Cin, pass == 1);
        const int splits = wgrad_b3_splits((long long)N * Ho * Wo, ((Cout + ts - 1) / ts) * ((Cin + ts - 1) / ts), KH * KW, ts);
        const size_t b = splits > 1 ? (size_t)splits * Cout * Cin * KH * KW * sizeof(float) : 0;
        need = b > need ? b : need;
    }
    return need;
}

static int wgrad_b3_run(const float *dz, const float *x, const uint16_t *dz_hi, const uint16_t *dz_lo, const uint16_t *x_hi,
                        const uint16_t *x_lo, float *dw, int N, int H, int W, int Ho, int Wo, int Cout, int Cin, int KH, int KW,
                        int stride, int pad_t, int pad_l, void *workspace, size_t workspace_bytes, void *stream) {
    const bool split_in = dz_hi != nullptr;
    if ((split_in ? (!dz_lo || !x_hi || !x_lo) : (!dz || !x)) || !dw || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 ||
        Cin <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad_t < 0 || pad_l < 0 || (long long)N * Ho * Wo >= (1ll << 31) || KH * KW > 65535)
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d_wgrad_b3: bad argument");
    if ((Cout & 3) || (Cin & 3))
        return cer_set_error(CER_ERR_UNSUPPORTED, "conv2d_wgrad_b3: Cout and Cin must be multiples of 4 (use cer_conv2d_wgrad)");
    int ts = wgrad_b3_tile(Cout, Cin, split_in);
    if (ts == 256 && !((long long)(32 + 2ll * H * W) * Cin * 2 < (1ll << 31) && 32ll * Cout * 2 < (1ll << 31))) ts = 128;
    const int R = N * Ho * Wo, tiles = ((Cout + ts - 1) / ts) * ((Cin + ts - 1) / ts), taps = KH * KW;
    const int splits = wgrad_b3_splits(R, tiles, taps, ts);
    const size_t n = (size_t)Cout * Cin * taps;
    if (splits > 1 && (!workspace || workspace_bytes < (size_t)splits * n * sizeof(float)))
        return cer_set_error(CER_ERR_WORKSPACE, "conv2d_wgrad_b3: workspace too small");
    Wgrad2dArgs a{dz, x, dz_hi, dz_lo, x_hi, x_lo, splits > 1 ? (float *)workspace : dw, R, Cout, Cin, taps, KW, H, W, Ho, Wo,
                  stride, pad_t, pad_l, 0, tiles, splits};
    a.rows_per_split = ts == 64 ? ((R + splits - 1) / splits + 63) / 64 * 64 : ((R + splits - 1) / splits + 31) / 32 * 32;
    const dim3 grid((unsigned)((splits + 7) / 8 * 8 * tiles * taps));
    hipStream_t st = (hipStream_t)stream;
    if (ts == 256) {
        constexpr int lds = 2 * 8 * 32 * 256;
        auto k = conv2d_wgrad_b3_wide_kernel<0>;
        CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CER_LAUNCH(k, grid, dim3(512), lds, st, a);
    } else if (ts == 64) {
        if (split_in) CER_LAUNCH((conv2d_wgrad_b3_kernel<64, true>), grid, dim3(256), 0, st, a);
        else CER_LAUNCH((conv2d_wgrad_b3_kernel<64, false>), grid, dim3(256), 0, st, a);
    } else {
        // split planes with whole 16-byte channel chunks: LDS-DMA loader (offsets within a step stay far below 2^31)
        const bool dma = split_in && (Cout & 7) == 0 && (Cin & 7) == 0 && (long long)(32 + 2ll * H * W) * Cin * 2 < (1ll << 31) &&
                         32ll * Cout * 2 < (1ll << 31);
        if (dma) CER_LAUNCH((conv2d_wgrad_b3_kernel<128, true, true>), grid, dim3(256), 0, st, a);
        else if (split_in) CER_LAUNCH((conv2d_wgrad_b3_kernel<128, true>), grid, dim3(256), 0, st, a);
        else CER_LAUNCH((conv2d_wgrad_b3_kernel<128, false>), grid, dim3(256), 0, st, a);
    }
    if (splits > 1)
        CER_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (const float4 *)workspace, (float4 *)dw,
                   n / 4, splits);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_conv2d_wgrad_b3(const float *dz, const float *x, float *dw, int N, int H, int W, int Ho, int Wo, int Cout,
                                   int Cin, int KH, int KW, int stride, int pad_t, int pad_l, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    return wgrad_b3_run(dz, x, nullptr, nullptr, nullptr, nullptr, dw, N, H, W, Ho, Wo, Cout, Cin, KH, KW, stride, pad_t, pad_l, workspace,
                        workspace_bytes, stream);
}

extern "C" int cer_conv2d_wgrad_b3s(const uint16_t *dz_hi, const uint16_t *dz_lo, const uint16_t *x_hi, const uint16_t *x_lo,
                                    float *dw, int N, int H, int W, int Ho, int Wo, int Cout, int Cin, int KH, int KW, int stride,
                                    int pad_t, int pad_l, void *workspace, size_t workspace_bytes, void *stream) {
    if (!dz_hi) return cer_set_error(CER_ERR_INVALID_ARG, "conv2d_wgrad_b3s: NULL operand");
    return wgrad_b3_run(nullptr, nullptr, dz_hi, dz_lo, x_hi, x_lo, dw, N, H, W, Ho, Wo, Cout, Cin, KH, KW, stride, pad_t, pad_l, workspace,
                        workspace_bytes, stream);
}
