// wgrad.hip -- weight gradient of the causal dilated conv1d / linear layers of the
// trainable tail on the fp32 matrix cores.
//
//   dW[co][ci][j] = sum_r dZ[r][co] * X[r - (k-1-j)*dil][ci]      (source row inside the same
//                                                                 length-L sequence, else 0)
//
// This is a "TN" GEMM whose reduction index is the row r, which is the slow axis of both
// operands.  v_mfma_f32_32x32x2_f32 wants A[i][k] / B[k][j] with lane = (i or j) and the
// two k of a step in the lane halves, so the natural row-major LDS image [k = row][channel]
// is read with conflict-free ds_read_b32 (32 consecutive floats per half-wave): no
// transpose is needed anywhere.
#include "cer_internal.h"

namespace cer {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const float *dz, *x;
    float *dw;
    int R, L, Cout, Cin, k, dil, dz_ld, x_ld;
    // 2-D mode (H > 0): row r = output pixel (n, ho, wo), tap = kh*KW + kw, source pixel
    // (n, ho*stride - pad_t + kh, wo*stride - pad_l + kw) of the NHWC input, zero outside the image
    int H, W, Ho, Wo, KW, stride, pad_t, pad_l;
    int rows_per_split;  // multiple of WG_ROWS; blockIdx.z = tap + k * split, partial results in slabs of Cout * Cin * k floats
};

constexpr int WG_ROWS = 32;  // reduction rows per step
constexpr int WG_T = 64;     // output tile edge (couts x cins)
constexpr int WG_PITCH = 64;

__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(WgradArgs p) {
    __shared__ __attribute__((aligned(16))) float As[2][WG_ROWS][WG_PITCH];  // dZ tile  [row][cout]
    __shared__ __attribute__((aligned(16))) float Bs[2][WG_ROWS][WG_PITCH];  // X tile   [row][cin]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ti = wave & 1, tj = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int co0 = blockIdx.x * WG_T, ci0 = blockIdx.y * WG_T, tap = blockIdx.z % p.k, split = blockIdx.z / p.k;
    const int r_begin = split * p.rows_per_split, r_end = min(p.R, r_begin + p.rows_per_split);
    const int shift = (p.k - 1 - tap) * p.dil;
    const int kh2 = p.H > 0 ? tap / p.KW : 0, kw2 = p.H > 0 ? tap - kh2 * p.KW : 0;

    // staging: 32 rows x 16 float4 per operand = 512 float4 -> 2 per thread per operand
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    float4 ra[2], rb[2];
    auto load_step = [&](int r0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r0 + srow + 16 * i;
            float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
            if (r < r_end) {
                const int co = co0 + scol, ci = ci0 + scol;
                const float *ap = p.dz + (size_t)r * p.dz_ld + co;
                if (co + 3 < p.Cout && ((p.dz_ld & 3) == 0)) {
                    a = *reinterpret_cast<const float4 *>(ap);
                } else {
                    if (co < p.Cout) a.x = ap[0];
                    if (co + 1 < p.Cout) a.y = ap[1];
                    if (co + 2 < p.Cout) a.z = ap[2];
                    if (co + 3 < p.Cout) a.w = ap[3];
                }
                long long src = -1;
                if (p.H > 0) {
                    const int hw = p.Ho * p.Wo;
                    const int n = r / hw, q = r - n * hw;
                    const int ho = q / p.Wo, wo = q - ho * p.Wo;
                    const int hi = ho * p.stride - p.pad_t + kh2, wi = wo * p.stride - p.pad_l + kw2;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) src = ((long long)n * p.H + hi) * p.W + wi;
                } else if (r % p.L >= shift) {
                    src = r - shift;
                }
                if (src >= 0) {
                    const float *bp = p.x + (size_t)src * p.x_ld + ci;
                    if (ci + 3 < p.Cin && ((p.x_ld & 3) == 0)) {
                        b = *reinterpret_cast<const float4 *>(bp);
                    } else {
                        if (ci < p.Cin) b.x = bp[0];
                        if (ci + 1 < p.Cin) b.y = bp[1];
                        if (ci + 2 < p.Cin) b.z = bp[2];
                        if (ci + 3 < p.Cin) b.w = bp[3];
                    }
                }
            }
            ra[i] = a;
            rb[i] = b;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<float4 *>(&As[buf][srow + 16 * i][scol]) = ra[i];
            *reinterpret_cast<float4 *>(&Bs[buf][srow + 16 * i][scol]) = rb[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int steps = r_end > r_begin ? (r_end - r_begin + WG_ROWS - 1) / WG_ROWS : 0;
    load_step(r_begin);
    store_step(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) load_step(r_begin + (s + 1) * WG_ROWS);
#pragma unroll
        for (int kk = 0; kk < WG_ROWS / 2; ++kk) {
            const float a = As[buf][2 * kk + half][ti * 32 + l31];
            const float b = Bs[buf][2 * kk + half][tj * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (s + 1 < steps) store_step(buf ^ 1);
        __syncthreads();
    }
    // D[i = cout][j = cin]: lane -> cin, register r -> cout (r&3) + 8*(r>>2) + 4*half
    const int ci = ci0 + tj * 32 + l31;
    if (ci < p.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < p.Cout) p.dw[(size_t)split * p.Cout * p.Cin * p.k + ((size_t)co * p.Cin + ci) * p.k + tap] = acc[r];
        }
    }
}

}  // namespace cer

using namespace cer;

// Row splits: the tail's layers reduce over only R = B * L = 1024 rows with 32 .. 320 (tile, tap) blocks, each walking all
// rows; splitting the rows over blocks (partials in a workspace, added in a fixed order) fills the chip: 56 -> ~20 us per layer.
static int wgrad_splits(int R, long long blocks) {
    long long s = (512 + blocks - 1) / blocks;
    const long long max_s = (R + 127) / 128;
    if (s > max_s) s = max_s;
    return (int)(s < 1 ? 1 : s);
}

__global__ void wgrad_fold_kernel(const float *__restrict__ part, float *__restrict__ dw, size_t n, int splits) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = part[i];
    for (int k = 1; k < splits; ++k) s += part[(size_t)k * n + i];
    dw[i] = s;
}

// splits_out != NULL: leave the split-R partial slabs in the workspace (the caller's next kernel sums them) and report how many
static int wgrad_launch(WgradArgs a, float *dw, void *workspace, size_t workspace_bytes, hipStream_t st, int *splits_out = nullptr) {
    dim3 grid((a.Cout + WG_T - 1) / WG_T, (a.Cin + WG_T - 1) / WG_T, a.k);
    const size_t n = (size_t)a.Cout * a.Cin * a.k;
    int splits = wgrad_splits(a.R, (long long)grid.x * grid.y * grid.z);
    if (splits > 1 && (!workspace || workspace_bytes < (size_t)splits * n * sizeof(float))) splits = 1;   // no workspace: one pass
    a.rows_per_split = ((a.R + splits - 1) / splits + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    a.dw = splits > 1 ? (float *)workspace : dw;
    grid.z = a.k * splits;
    CER_LAUNCH(conv1d_wgrad_kernel, grid, dim3(256), 0, st, a);
    if (splits_out) *splits_out = splits;
    else if (splits > 1) CER_LAUNCH(wgrad_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float *)workspace, dw, n, splits);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" size_t cer_conv_wgrad_workspace_bytes(long long R, int Cout, int Cin, int k) {
    if (R <= 0 || Cout <= 0 || Cin <= 0 || k <= 0) return 0;
    const long long blocks = (long long)((Cout + WG_T - 1) / WG_T) * ((Cin + WG_T - 1) / WG_T) * k;
    const int splits = wgrad_splits((int)(R > 0x7fffffff ? 0x7fffffff : R), blocks);
    return splits > 1 ? (size_t)splits * Cout * Cin * k * sizeof(float) : 0;
}

extern "C" int cer_conv1d_wgrad(const float *dz, int dz_ld, const float *x, int x_ld, float *dw, int R, int L,
                                int Cout, int Cin, int k, int dil, void *workspace, size_t workspace_bytes, void *stream) {
    if (!dz || !x || !dw || R <= 0 || L <= 0 || Cout <= 0 || Cin <= 0 || k <= 0 || dil <= 0 || dz_ld < Cout ||
        x_ld < Cin || (R % L) != 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "conv1d_wgrad: bad argument (R must be a multiple of L)");
    WgradArgs a{dz, x, dw, R, L, Cout, Cin, k, dil, dz_ld, x_ld, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return wgrad_launch(a, dw, workspace, workspace_bytes, (hipStream_t)stream);
}

// The weight gradient of a weight-normed 1-D conv straight into (dv, dg): the weight-norm backward kernel (tail_kernels.hip) sums
// the split-R partial slabs itself -- two launches instead of weight gradient + fold + weight-norm backward.  workspace: at least
// max(cer_conv_wgrad_workspace_bytes(R, Cout, Cin, k), Cout * Cin * k floats).
extern "C" int cer_weight_norm_bwd_partials(const float *dw_parts, int splits, const float *v, const float *g, const float *norm, float *dv,
                                            float *dg, int rows, int E, void *stream);
extern "C" int cer_conv1d_wgrad_weight_norm_bwd(const float *dz, int dz_ld, const float *x, int x_ld, int R, int L, int Cout, int Cin, int k,
                                                int dil, const float *v, const float *g, const float *norm, float *dv, float *dg,
                                                void *workspace, size_t workspace_bytes, void *stream) {
    if (!dz || !x || !v || !g || !norm || !dv || !dg || R <= 0 || L <= 0 || Cout <= 0 || Cin <= 0 || k <= 0 || dil <= 0 || dz_ld < Cout ||
        x_ld < Cin || (R % L) != 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "conv1d_wgrad_weight_norm_bwd: bad argument (R must be a multiple of L)");
    const size_t n = (size_t)Cout * Cin * k;
    if (!workspace || workspace_bytes < n * sizeof(float) || workspace_bytes < cer_conv_wgrad_workspace_bytes(R, Cout, Cin, k))
        return cer_set_error(CER_ERR_WORKSPACE, "conv1d_wgrad_weight_norm_bwd: workspace too small");
    WgradArgs a{dz, x, (float *)workspace, R, L, Cout, Cin, k, dil, dz_ld, x_ld, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int splits = 1;
    const int rc = wgrad_launch(a, (float *)workspace, workspace, workspace_bytes, (hipStream_t)stream, &splits);
    if (rc) return rc;
    return cer_weight_norm_bwd_partials((const float *)workspace, splits, v, g, norm, dv, dg, Cout, Cin * k, stream);
}

extern "C" int cer_conv2d_wgrad(const float *dz, const float *x, float *dw, int N, int H, int W, int Ho, int Wo, int Cout,
                                int Cin, int KH, int KW, int stride, int pad_t, int pad_l, void *workspace, size_t workspace_bytes,
                                void *stream) {
    if (!dz || !x || !dw || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 ||
        stride <= 0 || pad_t < 0 || pad_l < 0 || (long long)N * Ho * Wo >= (1ll << 31) || KH * KW > 65535)
        return cer_set_error(CER_ERR_INVALID_ARG, "conv2d_wgrad: bad argument");
    WgradArgs a{dz, x, dw, N * Ho * Wo, 1, Cout, Cin, KH * KW, 1, Cout, Cin, H, W, Ho, Wo, KW, stride, pad_t, pad_l, 0};
    return wgrad_launch(a, dw, workspace, workspace_bytes, (hipStream_t)stream);
}
