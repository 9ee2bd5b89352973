// frames.hip -- the video-frame input transform of the reference's Dataset on the GPU
// (/root/reference/base/dataset.py:487-508, base/transforms3D.py:33-143): uint8 frames
// [n,H,W,3] -> GroupScale(size) -> GroupRandomCrop / GroupCenterCrop(crop) ->
// GroupRandomHorizontalFlip -> ToTorchFormatTensor (/255, CHW) -> Normalize(mean, std), fused.
//
// GroupScale is torchvision Resize == PIL Image.resize(BILINEAR): an antialiased triangle filter in
// Pillow's 8-bit fixed-point arithmetic (coefficients scaled by 2^22, horizontal pass into a uint8
// intermediate, then the vertical pass).  The kernel reproduces that arithmetic bit for bit; the
// integer coefficient tables are computed by the host (frames.py) exactly as Pillow's
// precompute_coeffs does and handed over as device arrays.
//
// HBM-bound: 3*H*W bytes in, 12*crop*crop bytes out per frame.  One block owns a band of ROWS_PER_BLOCK
// output rows of one frame: the input rows the band needs are fetched ONCE with 16-byte coalesced
// loads into LDS, resampled horizontally (only the cropped columns, already flipped) into a uint8 LDS
// image, then vertically into the normalised fp32 output.
#include <stdint.h>

#include "cer_internal.h"

namespace cer {

constexpr int FR_PREC = 32 - 8 - 2;  // Pillow's PRECISION_BITS
constexpr int FR_BAND = 8;           // output rows per block
constexpr int FR_FAST_TAPS = 13;     // horizontal taps of the dword fast path (256 -> 48: ksize 13)

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

struct FramesArgs {
    const uint8_t *frames;
    const int32_t *hb, *hk, *vb, *vk;  // bounds [size][2] = (first, count), coefficients [size][ks]
    const int32_t *crop_xyf;           // [groups][3] = (x1, y1, flip)
    float *out;                        // [n][3][crop][crop]
    uint8_t *out_u8;                   // optional [n][crop][crop][3]
    int n, H, W, hks, vks, size, crop, frames_per_group, max_rows;
    float mean, stdv;
};

__global__ __launch_bounds__(256) void frames_transform_kernel(FramesArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fr_smem[];
    const int f = blockIdx.y, band = blockIdx.x, tid = threadIdx.x;
    const int32_t *cx = p.crop_xyf + (size_t)(f / p.frames_per_group) * 3;
    const int x1 = cx[0], y1 = cx[1], flip = cx[2];
    const int yy0 = band * FR_BAND, yy1 = min(p.crop, yy0 + FR_BAND);
    const int r0 = p.vb[2 * (y1 + yy0)];
    const int r1 = p.vb[2 * (y1 + yy1 - 1)] + p.vb[2 * (y1 + yy1 - 1) + 1];  // one past the last input row
    const int rows = r1 - r0, rowb = p.W * 3;
    uint8_t *src = fr_smem;                                        // [rows][W*3]
    uint8_t *tmp = fr_smem + (((size_t)p.max_rows * rowb + 15) & ~(size_t)15);  // [rows][crop][3]
    // coefficient rows of this block's columns / output rows, zero-padded to 16 taps (13 dependent global loads per
    // output were the bulk of the kernel's memory instructions)
    int32_t *hkl = reinterpret_cast<int32_t *>(tmp + (((size_t)p.max_rows * p.crop * 3 + 15) & ~(size_t)15));  // [crop][16]
    int32_t *vkl = hkl + p.crop * 16;                                                                          // [FR_BAND][16]
    const bool ktab = p.hks <= 16 && p.vks <= 16;
    if (ktab) {
        for (int i = tid; i < p.crop * 16; i += 256) {
            const int xc = i >> 4, j = i & 15;
            const int col = x1 + (flip ? p.crop - 1 - xc : xc);
            hkl[i] = j < p.hks ? p.hk[(size_t)col * p.hks + j] : 0;
        }
        for (int i = tid; i < (yy1 - yy0) * 16; i += 256) {
            const int ry = i >> 4, j = i & 15;
            vkl[i] = j < p.vks ? p.vk[(size_t)(y1 + yy0 + ry) * p.vks + j] : 0;
        }
    }
    const uint8_t *g = p.frames + ((size_t)f * p.H + r0) * rowb;
    const size_t nbytes = (size_t)rows * rowb;
    if (((reinterpret_cast<uintptr_t>(g) | nbytes) & 15) == 0) {
        // four loads in flight per thread before the first LDS store (a load-store-per-iteration loop serialises the
        // HBM latency: one band is only ~10 loads per thread)
        const uint4 *g4 = reinterpret_cast<const uint4 *>(g);
        uint4 *s4 = reinterpret_cast<uint4 *>(src);
        const size_t n16 = nbytes / 16;
        for (size_t i = tid; i < n16; i += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * 256 < n16) v[u] = g4[i + u * 256];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * 256 < n16) s4[i + u * 256] = v[u];
        }
    } else {
        for (size_t i = tid; i < nbytes; i += 256) src[i] = g[i];
    }
    __syncthreads();
    // horizontal pass, column already flipped
    const int per_row = p.crop * 3;
    if (((rowb & 3) == 0) && p.hks <= FR_FAST_TAPS && ktab) {
        // fast path: one thread per (row, cropped column) does all three channels.  Its taps are the 3*cnt <= 39
        // CONSECUTIVE bytes from xmin*3 of the row: 11 aligned dword LDS reads, v_alignbyte to drop the per-lane
        // misalignment, after which every (tap, channel) byte sits at a compile-time position.
        for (int i = tid; i < rows * p.crop; i += 256) {
            const int r = i / p.crop, xc = i - r * p.crop;
            const int col = x1 + (flip ? p.crop - 1 - xc : xc);
            const int xmin = p.hb[2 * col];
            const int4 *k4 = reinterpret_cast<const int4 *>(hkl + xc * 16);
            const int4 ka = k4[0], kb = k4[1], kc = k4[2], kd = k4[3];
            const int k[16] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w, kc.x, kc.y, kc.z, kc.w, kd.x, kd.y, kd.z, kd.w};
            const int b0 = r * rowb + xmin * 3, mis = b0 & 3;
            const uint32_t *w32 = reinterpret_cast<const uint32_t *>(src + (b0 - mis));
            constexpr int NW32 = (FR_FAST_TAPS * 3 + 3 + 3) / 4;  // dwords that cover 39 bytes at any misalignment
            uint32_t raw[NW32 + 1];
            const int last = (rows * rowb - (b0 - mis) + 3) / 4;  // dwords available before the end of the staged rows
#pragma unroll
            for (int q = 0; q <= NW32; ++q) raw[q] = q < last ? w32[q] : 0u;
            uint32_t al[NW32];
#pragma unroll
            for (int q = 0; q < NW32; ++q) al[q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], (uint32_t)mis);
            int a0 = 1 << (FR_PREC - 1), a1 = a0, a2 = a0;
#pragma unroll
            for (int j = 0; j < FR_FAST_TAPS; ++j) {
                const int kj = k[j];  // zero beyond the tap count
                a0 += (int)((al[(3 * j + 0) >> 2] >> (8 * ((3 * j + 0) & 3))) & 255u) * kj;
                a1 += (int)((al[(3 * j + 1) >> 2] >> (8 * ((3 * j + 1) & 3))) & 255u) * kj;
                a2 += (int)((al[(3 * j + 2) >> 2] >> (8 * ((3 * j + 2) & 3))) & 255u) * kj;
            }
            uint8_t *t = tmp + (size_t)i * 3;
            t[0] = (uint8_t)clip8(a0 >> FR_PREC);
            t[1] = (uint8_t)clip8(a1 >> FR_PREC);
            t[2] = (uint8_t)clip8(a2 >> FR_PREC);
        }
    } else {
        for (int i = tid; i < rows * per_row; i += 256) {  // generic: (row, cropped column, channel), byte reads
            const int r = i / per_row, q = i - r * per_row;
            const int xc = q / 3, c = q - xc * 3;
            const int col = x1 + (flip ? p.crop - 1 - xc : xc);
            const int xmin = p.hb[2 * col], cnt = p.hb[2 * col + 1];
            const int32_t *k = p.hk + (size_t)col * p.hks;
            const uint8_t *s = src + (size_t)r * rowb + xmin * 3 + c;
            int acc = 1 << (FR_PREC - 1);
            for (int j = 0; j < cnt; ++j) acc += (int)s[3 * j] * k[j];
            tmp[i] = (uint8_t)clip8(acc >> FR_PREC);
        }
    }
    __syncthreads();
    // vertical pass + ToTensor + Normalize
    for (int i = tid; i < (yy1 - yy0) * per_row; i += 256) {
        const int ry = i / per_row, q = i - ry * per_row;
        const int c = q / p.crop, xc = q - c * p.crop;  // channel-major so that the fp32 stores coalesce
        const int yy = yy0 + ry, vy = y1 + yy;
        const int ymin = p.vb[2 * vy], cnt = p.vb[2 * vy + 1];
        const int32_t *k = ktab ? vkl + ry * 16 : p.vk + (size_t)vy * p.vks;
        const uint8_t *s = tmp + ((size_t)(ymin - r0) * p.crop + xc) * 3 + c;
        int acc = 1 << (FR_PREC - 1);
        for (int j = 0; j < cnt; ++j) acc += (int)s[(size_t)j * per_row] * k[j];
        const int v = clip8(acc >> FR_PREC);
        p.out[(((size_t)f * 3 + c) * p.crop + yy) * p.crop + xc] = ((float)v / 255.0f - p.mean) / p.stdv;
        if (p.out_u8) p.out_u8[(((size_t)f * p.crop + yy) * p.crop + xc) * 3 + c] = (uint8_t)v;
    }
}

}  // namespace cer

using namespace cer;

extern "C" int cer_frames_transform(const uint8_t *frames, int n_frames, int H, int W, const int32_t *hbounds,
                                    const int32_t *hcoef, int hksize, const int32_t *vbounds, const int32_t *vcoef,
                                    int vksize, int size, int crop, const int32_t *crop_xyf, int frames_per_group,
                                    int max_band_rows, float mean, float stdv, float *out, uint8_t *out_u8,
                                    void *stream) {
    if (!frames || !hbounds || !hcoef || !vbounds || !vcoef || !crop_xyf || !out)
        return cer_set_error(CER_ERR_INVALID_ARG, "frames_transform: NULL pointer");
    if (n_frames <= 0 || H <= 0 || W <= 0 || size <= 0 || crop <= 0 || crop > size || hksize <= 0 || vksize <= 0 ||
        frames_per_group <= 0 || max_band_rows <= 0 || max_band_rows > H || stdv == 0.f)
        return cer_set_error(CER_ERR_INVALID_ARG, "frames_transform: bad geometry");
    if (n_frames > 65535) return cer_set_error(CER_ERR_UNSUPPORTED, "frames_transform: more than 65535 frames per call");
    const size_t lds = (((size_t)max_band_rows * W * 3 + 15) & ~(size_t)15) + (((size_t)max_band_rows * crop * 3 + 15) & ~(size_t)15) +
                       (size_t)(crop + FR_BAND) * 16 * sizeof(int32_t);
    if (lds > 160 * 1024) return cer_set_error(CER_ERR_UNSUPPORTED, "frames_transform: a band of input rows exceeds the 160 KiB LDS");
    FramesArgs a{};
    a.frames = frames; a.hb = hbounds; a.hk = hcoef; a.vb = vbounds; a.vk = vcoef; a.crop_xyf = crop_xyf;
    a.out = out; a.out_u8 = out_u8; a.n = n_frames; a.H = H; a.W = W; a.hks = hksize; a.vks = vksize;
    a.size = size; a.crop = crop; a.frames_per_group = frames_per_group; a.max_rows = max_band_rows;
    a.mean = mean; a.stdv = stdv;
    auto k = frames_transform_kernel;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CER_LAUNCH(k, dim3((crop + FR_BAND - 1) / FR_BAND, n_frames), dim3(256), lds, (hipStream_t)stream, a);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_frames_band_rows(void) { return FR_BAND; }
