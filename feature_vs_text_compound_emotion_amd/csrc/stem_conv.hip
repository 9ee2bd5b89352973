// stem_conv.hip -- the IR-50 input layer, Conv2d(3, 64, 3x3, stride 1, pad 1) + BatchNorm2d + PReLU
// (reference models/arcface_model.py:130-132, :148), as a DIRECT convolution on the vector ALUs.
//
// Why not the implicit-GEMM kernel: with Cin = 3 the layer is 27 multiply-adds per output value -- 89 GFMA for 1024 frames
// of 224x224, 1.2-2.3 ms of VALU time -- against 13.2 GB of output.  It is a WRITE-bound layer, and the fp32 MFMA kernel
// (K padded to 32, a scalar NCHW gather per element, 16-byte stores strided by the row pitch) took 8.8-10.5 ms for it.  In
// model.train() the batch-statistics BatchNorm needed the raw result twice more (bn_apply: 13.2 GB read + the activated
// tensor written), 16 ms in all.  Because the conv is this cheap it is cheaper to RECOMPUTE it than to store it:
//   pass 1 (no output tensor): conv -> per-block sum / sum of squares of the raw result (the BatchNorm statistics);
//   pass 2 (scale, shift now known): conv again -> * scale + shift -> PReLU -> the activated tensor in the consumer's storage
//           (fp32 / split hi+lo bf16 / one narrow plane) + the statistics of that output (the next unit's pre-conv BatchNorm).
// The raw tensor never exists: 26 GB of HBM traffic less per step.  Eval mode is pass 2 alone (BatchNorm scale folded into
// the weights, shift as the bias).
//
// A block owns `rows` output rows of one frame: the rows + 2 input rows of the three colour planes are staged in LDS with
// their zero border (NCHW fp32 frames are read as they are), a thread owns 4 couts (its 108 weights live in registers as
// float2 pairs for v_pk_fma_f32) and walks the pixels 16 apart, so the 16 threads of a pixel write its 64 couts as one
// 256-byte line and a wave writes four consecutive pixels; the window values are LDS broadcasts.
#include "conv_common.h"

namespace cer {

typedef float sc_f32x2 __attribute__((ext_vector_type(2)));

struct StemArgs {
    const float *x;                      // [N, 3, H, W]
    const float *w;                      // [64][Kpad], K index = (kh * 3 + kw) * 3 + ci (cer_conv_kpad(3, 3, 3) = 32)
    const float *scale, *shift, *alpha;  // per cout, any may be NULL (1 / 0 / no activation)
    float *y;                            // fp32 output or NULL
    uint16_t *y_hi, *y_lo;               // split output (both) or one narrow plane (y_hi, storage != 0) or NULL
    float *stats;                        // [blocks][2][64] or NULL: of the RAW result (no output given) or of the output
    int N, H, W, Kpad, narrow, rows, blocks_per_image;
};

template <bool APPLY>
__global__ __launch_bounds__(256) void stem_conv3x3_kernel(StemArgs a) {
    extern __shared__ float stem_xs[];                 // [rows + 2][3][W + 2]
    __shared__ float red[2][16][64];
    const int tid = threadIdx.x, g = tid & 15, ps = tid >> 4;
    const int n = blockIdx.x / a.blocks_per_image, r0 = (blockIdx.x % a.blocks_per_image) * a.rows;
    const int RB = a.H - r0 < a.rows ? a.H - r0 : a.rows;
    const int WPD = a.W + 2;
    const int total = (RB + 2) * 3 * WPD;
    for (int i = tid; i < total; i += 256) {
        const int col = i % WPD, t = i / WPD, ci = t % 3, row = t / 3;
        const int gy = r0 - 1 + row, gx = col - 1;
        float v = 0.f;
        if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) v = a.x[(((size_t)n * 3 + ci) * a.H + gy) * a.W + gx];
        stem_xs[i] = v;
    }
    sc_f32x2 w01[27], w23[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        w01[k] = sc_f32x2{a.w[(size_t)(4 * g + 0) * a.Kpad + k], a.w[(size_t)(4 * g + 1) * a.Kpad + k]};
        w23[k] = sc_f32x2{a.w[(size_t)(4 * g + 2) * a.Kpad + k], a.w[(size_t)(4 * g + 3) * a.Kpad + k]};
    }
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, al[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (APPLY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (a.scale) sc[j] = a.scale[4 * g + j];
            if (a.shift) sh[j] = a.shift[4 * g + j];
            if (a.alpha) al[j] = a.alpha[4 * g + j];
        }
    }
    __syncthreads();
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int rr = 0; rr < RB; ++rr) {
        const float *rowp = stem_xs + rr * 3 * WPD;
        for (int px = ps; px < a.W; px += 16) {
            sc_f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const float v = rowp[(kh * 3 + ci) * WPD + px + kw];
                        const sc_f32x2 vv = {v, v};
                        const int k = (kh * 3 + kw) * 3 + ci;
                        acc01 = __builtin_elementwise_fma(vv, w01[k], acc01);
                        acc23 = __builtin_elementwise_fma(vv, w23[k], acc23);
                    }
            float o[4] = {acc01[0], acc01[1], acc23[0], acc23[1]};
            if constexpr (APPLY) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = o[j] * sc[j] + sh[j];
                    o[j] = o[j] >= 0.f ? o[j] : o[j] * al[j];
                }
                const size_t off = (((size_t)n * a.H + r0 + rr) * a.W + px) * 64 + 4 * g;
                if (a.y) *reinterpret_cast<float4 *>(a.y + off) = make_float4(o[0], o[1], o[2], o[3]);
                if (a.y_hi) {
                    if (a.narrow) store_narrow4(a.y_hi + off, o, a.narrow);
                    else store_split4(a.y_hi + off, a.y_lo + off, o);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s1[j] += o[j];
                s2[j] += o[j] * o[j];
            }
        }
    }
    if (!a.stats) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][ps][4 * g + j] = s1[j];
        red[1][ps][4 * g + j] = s2[j];
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[which][q][c];
        a.stats[((size_t)blockIdx.x * 2 + which) * 64 + c] = t;
    }
}

constexpr int STEM_ROWS = 4;   // output rows per block: 6 staged input rows x 3 planes, 16 KiB at W = 224

}  // namespace cer

using namespace cer;

extern "C" int cer_stem_conv3x3_stats_rows(int N, int H) {
    if (N <= 0 || H <= 0) return 0;
    return N * ((H + STEM_ROWS - 1) / STEM_ROWS);
}

extern "C" int cer_stem_conv3x3(const float *x, const float *w, int Kpad, const float *scale, const float *shift, const float *alpha,
                                float *y, uint16_t *y_hi, uint16_t *y_lo, int storage, float *stats, int N, int H, int W,
                                void *stream) {
    if (!x || !w || N <= 0 || H <= 0 || W <= 0 || Kpad < 27)
        return cer_set_error(CER_ERR_INVALID_ARG, "stem_conv3x3: x, w, positive sizes and Kpad >= 27 are required");
    const bool apply = y || y_hi;
    if (!apply && (!stats || scale || shift || alpha || y_lo))
        return cer_set_error(CER_ERR_INVALID_ARG, "stem_conv3x3: without an output tensor this is the statistics pass (stats only)");
    if (storage != CER_STORE_NONE && storage != CER_STORE_BF16 && storage != CER_STORE_F16)
        return cer_set_error(CER_ERR_INVALID_ARG, "stem_conv3x3: unknown storage type");
    if (y_hi && ((storage == CER_STORE_NONE) != (y_lo != nullptr)))
        return cer_set_error(CER_ERR_INVALID_ARG, "stem_conv3x3: a split output needs both planes, a narrow one only y_hi");
    if ((long long)N * H * W >= (1ll << 31)) return cer_set_error(CER_ERR_UNSUPPORTED, "stem_conv3x3: more than 2^31 pixels");
    const size_t lds = (size_t)(STEM_ROWS + 2) * 3 * (W + 2) * sizeof(float);
    if (lds > 56 * 1024) return cer_set_error(CER_ERR_UNSUPPORTED, "stem_conv3x3: frames wider than 790 pixels");
    StemArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.alpha = alpha; a.y = y; a.y_hi = y_hi; a.y_lo = y_lo; a.stats = stats;
    a.N = N; a.H = H; a.W = W; a.Kpad = Kpad; a.narrow = storage; a.rows = STEM_ROWS;
    a.blocks_per_image = (H + STEM_ROWS - 1) / STEM_ROWS;
    const dim3 grid((unsigned)(N * a.blocks_per_image)), block(256);
    if (apply) CER_LAUNCH(stem_conv3x3_kernel<true>, grid, block, lds, (hipStream_t)stream, a);
    else CER_LAUNCH(stem_conv3x3_kernel<false>, grid, block, lds, (hipStream_t)stream, a);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
