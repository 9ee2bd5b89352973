// tail_kernels.hip -- bandwidth-bound kernels of the trainable tail (TCN glue,
// BatchNorm1d, LFAN cross-modal attention, LayerNorm, cross-entropy, dropout
// masks).  All tensors are channels-last rows: [R = B*L, C].
//
// These ops move a few MB per step; the design goal is one pass over the data
// per kernel, 64-wide wave reductions (no 32-lane idioms) and deterministic
// results (no float atomics: column sums use a fixed two-level tree).
#include <math.h>

#include "cer_internal.h"

namespace cer {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------- weight norm
// w = g * v / ||v||, one wave per output channel (row of E = Cin*k elements).
// Reference: torch.nn.utils.weight_norm(dim=0) as applied at
// models/temporal_convolutional_model.py:24,30.
// Optionally ALSO the two packed layouts the conv kernels read (cer_pack_conv_weight's, for a [Cout][Cin][k] 1-D filter):
// wp_f [Cout][Kpad_f] with column kh * Cin + c (forward) and wp_b [Cin][Kpad_b] with column (k - 1 - kh) * Cout + o (the
// flipped, transposed data-gradient filter), zero padded -- the TCN re-normalises its weights every step, and two pack launches
// per conv in the forward plus two in the backward were 48 of the tail's launches.
__global__ void weight_norm_fwd_kernel(const float *__restrict__ v, const float *__restrict__ g,
                                       float *__restrict__ w, float *__restrict__ norm, int rows, int E,
                                       float *__restrict__ wp_f, float *__restrict__ wp_b, int k, int Kpad_f, int Kpad_b) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *vr = v + (size_t)row * E;
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += vr[i] * vr[i];
    s = wave_sum(s);
    const float nrm = sqrtf(s), sc = g[row] / nrm;
    if (!wp_f) {
        for (int i = lane; i < E; i += 64) w[(size_t)row * E + i] = vr[i] * sc;
    } else {
        const int cin = E / k;
        for (int i = lane; i < E; i += 64) {
            const float val = vr[i] * sc;
            const int c = i / k, kh = i - c * k;
            w[(size_t)row * E + i] = val;
            wp_f[(size_t)row * Kpad_f + kh * cin + c] = val;
            wp_b[(size_t)c * Kpad_b + (k - 1 - kh) * rows + row] = val;
        }
        for (int i = E + lane; i < Kpad_f; i += 64) wp_f[(size_t)row * Kpad_f + i] = 0.f;
        const int padb = Kpad_b - k * rows;                    // padding columns of every wp_b row: dealt over the cout waves
        for (int j = row * 64 + lane; j < cin * padb; j += rows * 64) wp_b[(size_t)(j / padb) * Kpad_b + k * rows + j % padb] = 0.f;
    }
    if (lane == 0) norm[row] = nrm;
}

// dg = <dw, v>/||v||;  dv = g/||v|| * dw - g*<dw,v>/||v||^3 * v.  splits > 1: dw is the weight-gradient kernel's split-R
// partial slabs [splits][rows * E], summed here in the fixed order 0 .. splits-1 (what wgrad_fold_kernel would have done).
__global__ void weight_norm_bwd_kernel(const float *__restrict__ dw, const float *__restrict__ v,
                                       const float *__restrict__ g, const float *__restrict__ norm,
                                       float *__restrict__ dv, float *__restrict__ dg, int rows, int E, int splits) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *vr = v + (size_t)row * E, *dr = dw + (size_t)row * E;
    const size_t n = (size_t)rows * E;
    auto grad = [&](int i) {
        float d = dr[i];
        for (int k = 1; k < splits; ++k) d += dr[(size_t)k * n + i];
        return d;
    };
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += vr[i] * grad(i);
    s = wave_sum(s);
    const float nrm = norm[row], gg = g[row];
    const float a = gg / nrm, b = gg * s / (nrm * nrm * nrm);
    for (int i = lane; i < E; i += 64) dv[(size_t)row * E + i] = a * grad(i) - b * vr[i];
    if (lane == 0) dg[row] = s / nrm;
}

// ------------------------------------------------------------- column sums
// out[c] = sum_r a[r][c] * (b ? (b[r][c]-mean[c])*invstd[c] : 1)  (a == NULL with b: sum_r ((b-mean)*invstd)^2, the
// centred second moment of the row BatchNorm statistics).  Block = 32 columns x 8 row
// lanes; grid.x = column chunks, grid.y = row slabs (partials [grid.y][C] reduced by a
// second launch with b == NULL) -> deterministic.
__global__ void col_sum_kernel(const float *__restrict__ a, int a_ld, const float *__restrict__ b, int b_ld,
                               const float *__restrict__ mean, const float *__restrict__ invstd,
                               float *__restrict__ out, int R, int C, int rows_per_slab) {
    __shared__ float red[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(R, r0 + rows_per_slab);
    float s = 0.f;
    if (c < C) {
        const float mu = mean ? mean[c] : 0.f, is = invstd ? invstd[c] : 1.f;
        if (a) {
            for (int r = r0 + ry; r < r1; r += 8) {
                float v = a[(size_t)r * a_ld + c];
                if (b) v *= (b[(size_t)r * b_ld + c] - mu) * is;
                s += v;
            }
        } else {
            for (int r = r0 + ry; r < r1; r += 8) {
                const float d = (b[(size_t)r * b_ld + c] - mu) * is;
                s += d * d;
            }
        }
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cx];
        out[(size_t)blockIdx.y * C + c] = t;
    }
}

// The same with 4 channels per thread (16-byte loads; C and the row pitches multiples of 4): block = 32 channel quads (128
// channels) x 8 row lanes, four rows in flight per lane.  The scalar kernel above moves 128 bytes per half-wave and row and
// reached ~1 TB/s on the released encoder units' [6.4 M x 64] tensors (20 % of a whole-encoder training step).
__global__ __launch_bounds__(256) void col_sum4_kernel(const float *__restrict__ a, int a_ld, const float *__restrict__ b, int b_ld,
                                                        const float *__restrict__ mean, const float *__restrict__ invstd,
                                                        float *__restrict__ out, int R, int C, int rows_per_slab) {
    __shared__ float red[8][32][4];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cx) * 4;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(R, r0 + rows_per_slab);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mean) mu = *reinterpret_cast<const float4 *>(mean + c);
        if (invstd) is = *reinterpret_cast<const float4 *>(invstd + c);
        for (int r = r0 + ry; r < r1; r += 32) {
            float4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + 8 * u;
                const bool ok = rr < r1;
                va[u] = (a && ok) ? *reinterpret_cast<const float4 *>(a + (size_t)rr * a_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                vb[u] = (b && ok) ? *reinterpret_cast<const float4 *>(b + (size_t)rr * b_ld + c) : make_float4(mu.x, mu.y, mu.z, mu.w);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d[4] = {(vb[u].x - mu.x) * is.x, (vb[u].y - mu.y) * is.y, (vb[u].z - mu.z) * is.z, (vb[u].w - mu.w) * is.w};
                const float v[4] = {va[u].x, va[u].y, va[u].z, va[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += a ? (b ? v[e] * d[e] : v[e]) : d[e] * d[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ry][cx][e] = s[e];
    __syncthreads();
    if (ry == 0 && c < C) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += red[i][cx][e];
        *reinterpret_cast<float4 *>(out + (size_t)blockIdx.y * C + c) = make_float4(t[0], t[1], t[2], t[3]);
    }
}

// TWO column sums in one pass over the rows (round 3: the released encoder units' BatchNorm passes are column sums over
// 13-26 GB tensors, 8 % of a whole-encoder training step; each of these pairs used to be two passes):
//   a != NULL:  out1 = sum a,            out2 = sum a * (b - mean) * invstd      (BatchNorm backward: sum dy, sum dy * x_hat)
//   a == NULL:  out1 = sum (b - mean),   out2 = sum (b - mean)^2                 (batch statistics about the shift `mean`)
// Same block shape as col_sum4_kernel; partial rows go to out1 / out2 [slab][C].
__global__ __launch_bounds__(256) void col_sum_pair4_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                             const float *__restrict__ mean, const float *__restrict__ invstd,
                                                             float *__restrict__ out1, float *__restrict__ out2, int R, int C,
                                                             int rows_per_slab) {
    __shared__ float red[8][32][8];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cx) * 4;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(R, r0 + rows_per_slab);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mean) mu = *reinterpret_cast<const float4 *>(mean + c);
        if (invstd) is = *reinterpret_cast<const float4 *>(invstd + c);
        for (int r = r0 + ry; r < r1; r += 32) {
            float4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + 8 * u;
                const bool ok = rr < r1;
                va[u] = (a && ok) ? *reinterpret_cast<const float4 *>(a + (size_t)rr * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                vb[u] = ok ? *reinterpret_cast<const float4 *>(b + (size_t)rr * C + c) : make_float4(mu.x, mu.y, mu.z, mu.w);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d[4] = {(vb[u].x - mu.x) * is.x, (vb[u].y - mu.y) * is.y, (vb[u].z - mu.z) * is.z, (vb[u].w - mu.w) * is.w};
                const float v[4] = {va[u].x, va[u].y, va[u].z, va[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] += a ? v[e] : d[e];
                    s2[e] += a ? v[e] * d[e] : d[e] * d[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[ry][cx][e] = s1[e];
        red[ry][cx][4 + e] = s2[e];
    }
    __syncthreads();
    if (ry == 0 && c < C) {
        float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += red[i][cx][e];
        *reinterpret_cast<float4 *>(out1 + (size_t)blockIdx.y * C + c) = make_float4(t[0], t[1], t[2], t[3]);
        *reinterpret_cast<float4 *>(out2 + (size_t)blockIdx.y * C + c) = make_float4(t[4], t[5], t[6], t[7]);
    }
}

// ------------------------------------------------------------- TCN glue
// dz = dy * mask * leaky'(y)   (y = mask * leaky(z): sign(y) == sign(z) wherever mask != 0)
__global__ void act_mask_bwd_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ y,
                                    const float4 *__restrict__ mask, float4 *__restrict__ dz, size_t n4,
                                    float slope) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 g = dy[i], o = y[i], m = mask ? mask[i] : make_float4(1, 1, 1, 1);
    float4 r;
    r.x = g.x * m.x * (o.x > 0.f ? 1.f : slope);
    r.y = g.y * m.y * (o.y > 0.f ? 1.f : slope);
    r.z = g.z * m.z * (o.z > 0.f ? 1.f : slope);
    r.w = g.w * m.w * (o.w > 0.f ? 1.f : slope);
    dz[i] = r;
}

// TemporalBlock tail: out = leaky(a2 + res), a2 = mask2 * leaky(z2).
// du = dout * leaky'(out) (gradient of the residual branch and of a2); dz2 = du * mask2 * leaky'(a2)
__global__ void tblock_tail_bwd_kernel(const float4 *__restrict__ dout, const float4 *__restrict__ out,
                                       const float4 *__restrict__ a2, const float4 *__restrict__ mask2,
                                       float4 *__restrict__ du, float4 *__restrict__ dz2, size_t n4, float slope) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 g = dout[i], o = out[i], a = a2[i], m = mask2 ? mask2[i] : make_float4(1, 1, 1, 1);
    float4 u, z;
    u.x = g.x * (o.x > 0.f ? 1.f : slope); z.x = u.x * m.x * (a.x > 0.f ? 1.f : slope);
    u.y = g.y * (o.y > 0.f ? 1.f : slope); z.y = u.y * m.y * (a.y > 0.f ? 1.f : slope);
    u.z = g.z * (o.z > 0.f ? 1.f : slope); z.z = u.z * m.z * (a.z > 0.f ? 1.f : slope);
    u.w = g.w * (o.w > 0.f ? 1.f : slope); z.w = u.w * m.w * (a.w > 0.f ? 1.f : slope);
    du[i] = u;
    dz2[i] = z;
}

// ------------------------------------------------------------- BatchNorm over rows
// stats: mean / biased var per channel (two passes over an L2-resident tensor), running-stat
// update with the unbiased variance (torch semantics).  Block = 32 channels x 8 row lanes.
__global__ void bn_rows_stats_kernel(const float *__restrict__ x, int x_ld, float *__restrict__ save_mean,
                                     float *__restrict__ save_invstd, float *__restrict__ running_mean,
                                     float *__restrict__ running_var, int R, int C, float eps, float momentum) {
    __shared__ float red[8][33];
    __shared__ float mu_s[32];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (c < C)
        for (int r = ry; r < R; r += 8) s += x[(size_t)r * x_ld + c];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cx];
        mu_s[cx] = t / (float)R;
    }
    __syncthreads();
    const float mu = mu_s[cx];
    s = 0.f;
    if (c < C)
        for (int r = ry; r < R; r += 8) {
            float d = x[(size_t)r * x_ld + c] - mu;
            s += d * d;
        }
    __syncthreads();
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cx];
        const float var = t / (float)R;
        save_mean[c] = mu;
        save_invstd[c] = rsqrtf(var + eps);
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (float)R / (float)max(R - 1, 1);
        }
    }
}

// y = (x - mean) * invstd * w + b     (train: saved batch stats; eval: running stats via invstd_from_var)
__global__ void bn_rows_apply_kernel(const float *__restrict__ x, int x_ld, const float *__restrict__ mean,
                                     const float *__restrict__ invstd_or_var, int is_var, float eps,
                                     const float *__restrict__ w, const float *__restrict__ b,
                                     float *__restrict__ y, int y_ld, int R, int C) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * C) return;
    const int r = (int)(idx / C), c = (int)(idx - (size_t)r * C);
    const float is = is_var ? rsqrtf(invstd_or_var[c] + eps) : invstd_or_var[c];
    y[(size_t)r * y_ld + c] = (x[(size_t)r * x_ld + c] - mean[c]) * is * w[c] + b[c];
}

// dx = w*invstd/R * (R*dy - sum_dy - xhat * sum_dy_xhat)   (train); eval: dx = dy*w*invstd
__global__ void bn_rows_bwd_kernel(const float *__restrict__ dy, int dy_ld, const float *__restrict__ x, int x_ld,
                                   const float *__restrict__ mean, const float *__restrict__ invstd,
                                   const float *__restrict__ w, const float *__restrict__ sum_dy,
                                   const float *__restrict__ sum_dy_xhat, float *__restrict__ dx, int R, int C,
                                   int train) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * C) return;
    const int r = (int)(idx / C), c = (int)(idx - (size_t)r * C);
    const float is = invstd[c], g = dy[(size_t)r * dy_ld + c];
    if (train) {
        const float xh = (x[(size_t)r * x_ld + c] - mean[c]) * is;
        dx[idx] = w[c] * is * (g - (sum_dy[c] + xh * sum_dy_xhat[c]) / (float)R);
    } else {
        dx[idx] = g * w[c] * is;
    }
}

// ------------------------------------------------------------- LFAN cross-modal attention core
// Reference models/transformer.py:11-19,133-159.  For each (row, head): M x M softmax over
// MODALITIES with d = hd; vals = softmax(q k^T / sqrt(hd)) v + v.  qkv_m rows are
// [head][q(hd) | k(hd) | v(hd)].  One group of HD lanes per (row, head); lane = d.
constexpr int MAXM = 4;
struct FusionPtrs { const float *qkv[MAXM]; float *dqkv[MAXM]; };

template <int HD>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = HD / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int HD>
__global__ void lfan_attn_fwd_kernel(FusionPtrs P, float *__restrict__ vals, float *__restrict__ probs, int R,
                                     int H, int M) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / HD, d = threadIdx.x % HD;
    const bool live = gid < R * H;
    const int r = live ? gid / H : 0, h = live ? gid - r * H : 0;
    float q[MAXM], k[MAXM], v[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
            const float *p = P.qkv[m] + (size_t)r * (3 * HD * H) + h * 3 * HD;
            q[m] = p[d]; k[m] = p[HD + d]; v[m] = p[2 * HD + d];
        } else { q[m] = k[m] = v[m] = 0.f; }
    }
    const float scale = rsqrtf((float)HD);
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m >= M) break;
        float lg[MAXM], mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < MAXM; ++n) {
            lg[n] = (n < M) ? group_sum<HD>(q[m] * k[n]) * scale : -INFINITY;
            mx = fmaxf(mx, lg[n]);
        }
        float den = 0.f;
#pragma unroll
        for (int n = 0; n < MAXM; ++n) { lg[n] = (n < M) ? expf(lg[n] - mx) : 0.f; den += lg[n]; }
        float o = v[m];
#pragma unroll
        for (int n = 0; n < MAXM; ++n) {
            const float p = lg[n] / den;
            o += p * v[n];
            if (live && d == 0 && n < M) probs[(((size_t)r * H + h) * M + m) * M + n] = p;
        }
        if (live) vals[(size_t)r * (H * M * HD) + (h * M + m) * HD + d] = o;
    }
}

template <int HD>
__global__ void lfan_attn_bwd_kernel(FusionPtrs P, const float *__restrict__ dvals, const float *__restrict__ probs,
                                     int R, int H, int M) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / HD, d = threadIdx.x % HD;
    const bool live = gid < R * H;
    const int r = live ? gid / H : 0, h = live ? gid - r * H : 0;
    float q[MAXM], k[MAXM], v[MAXM], go[MAXM], dq[MAXM], dk[MAXM], dv[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        dq[m] = dk[m] = 0.f;
        if (m < M) {
            const float *p = P.qkv[m] + (size_t)r * (3 * HD * H) + h * 3 * HD;
            q[m] = p[d]; k[m] = p[HD + d]; v[m] = p[2 * HD + d];
            go[m] = dvals[(size_t)r * (H * M * HD) + (h * M + m) * HD + d];
        } else { q[m] = k[m] = v[m] = go[m] = 0.f; }
        dv[m] = go[m];  // the "+ V" residual
    }
    const float scale = rsqrtf((float)HD);
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m >= M) break;
        float p[MAXM], dp[MAXM], dot = 0.f;
#pragma unroll
        for (int n = 0; n < MAXM; ++n) {
            p[n] = (n < M) ? probs[(((size_t)r * H + h) * M + m) * M + n] : 0.f;
            dp[n] = (n < M) ? group_sum<HD>(go[m] * v[n]) : 0.f;
            dot += p[n] * dp[n];
            dv[n] += p[n] * go[m];
        }
#pragma unroll
        for (int n = 0; n < MAXM; ++n) {
            const float dl = p[n] * (dp[n] - dot) * scale;
            dq[m] += dl * k[n];
            dk[n] += dl * q[m];
        }
    }
    if (!live) return;
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m >= M) break;
        float *p = P.dqkv[m] + (size_t)r * (3 * HD * H) + h * 3 * HD;
        p[d] = dq[m]; p[HD + d] = dk[m]; p[2 * HD + d] = dv[m];
    }
}

// ------------------------------------------------------------- LayerNorm (one wave per row)
// y = LN(x * mask) * gamma + beta; saves mean / rstd per row.  Reference transformer.py:194-196.
__global__ void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mask,
                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                     float *__restrict__ y, int y_ld, float *__restrict__ save_mean,
                                     float *__restrict__ save_rstd, int R, int C, float eps) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float *xr = x + (size_t)row * C;
    const float *mr = mask ? mask + (size_t)row * C : nullptr;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c] * (mr ? mr[c] : 1.f);
    const float mu = wave_sum(s) / (float)C;
    s = 0.f;
    for (int c = lane; c < C; c += 64) { float d = xr[c] * (mr ? mr[c] : 1.f) - mu; s += d * d; }
    const float rstd = rsqrtf(wave_sum(s) / (float)C + eps);
    for (int c = lane; c < C; c += 64)
        y[(size_t)row * y_ld + c] = (xr[c] * (mr ? mr[c] : 1.f) - mu) * rstd * gamma[c] + beta[c];
    if (lane == 0 && save_mean) { save_mean[row] = mu; save_rstd[row] = rstd; }
}

// dx (w.r.t. the un-masked input) and dy*xhat (for the deterministic gamma-gradient column sum)
__global__ void layernorm_bwd_kernel(const float *__restrict__ dy, int dy_ld, const float *__restrict__ x,
                                     const float *__restrict__ mask, const float *__restrict__ gamma,
                                     const float *__restrict__ save_mean, const float *__restrict__ save_rstd,
                                     float *__restrict__ dx, float *__restrict__ dy_xhat, int R, int C) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float *xr = x + (size_t)row * C, *gr = dy + (size_t)row * dy_ld;
    const float *mr = mask ? mask + (size_t)row * C : nullptr;
    const float mu = save_mean[row], rstd = save_rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] * (mr ? mr[c] : 1.f) - mu) * rstd, g = gr[c] * gamma[c];
        s1 += g; s2 += g * xh;
    }
    s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float m = mr ? mr[c] : 1.f;
        const float xh = (xr[c] * m - mu) * rstd, g = gr[c] * gamma[c];
        dx[(size_t)row * C + c] = rstd * (g - s1 - xh * s2) * m;
        dy_xhat[(size_t)row * C + c] = gr[c] * xh;
    }
}

// ------------------------------------------------------------- cross entropy (mean) fwd + bwd
// Reference experiment.py:133 + trainer.py:380-383: labels arrive as float32 and are cast to long.
// Single block: loss = sum over valid rows (lse_r - z_r[y_r]) / n_valid; dlogits = (softmax - onehot) / n_valid.
// nn.CrossEntropyLoss semantics for the label values: ignore_index = -100 rows are skipped (zero gradient, not counted);
// any other label outside [0, C) -- or a NaN -- is an error: torch raises, here the row is never dereferenced, the loss and
// that row's gradient become NaN (the step fails visibly) and *bad_labels counts such rows for the host wrapper to raise on.
__global__ void cross_entropy_kernel(const float *__restrict__ logits, const float *__restrict__ labels,
                                     float *__restrict__ loss, float *__restrict__ dlogits, int *__restrict__ bad_labels,
                                     int R, int C) {
    __shared__ float red[16];
    __shared__ int redn[16], redb[16];
    __shared__ float s_inv;
    // pass 1: count valid / bad rows (the gradient scale 1 / n_valid is needed before any row is written)
    int nvalid = 0, nbad = 0;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const float lf = labels[r];
        const bool ignore = lf == -100.f;
        const bool ok = lf >= 0.f && lf < (float)C;  // false for NaN
        nvalid += ok ? 1 : 0;
        nbad += (!ok && !ignore) ? 1 : 0;
    }
    nvalid = wave_sum_int(nvalid);
    nbad = wave_sum_int(nbad);
    if ((threadIdx.x & 63) == 0) { redn[threadIdx.x >> 6] = nvalid; redb[threadIdx.x >> 6] = nbad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tn = 0, tb = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { tn += redn[i]; tb += redb[i]; }
        if (bad_labels) *bad_labels = tb;
        s_inv = tb > 0 ? __int_as_float(0x7fc00000) : 1.f / (float)tn;  // all rows ignored: 0/0 = NaN, as torch
        redn[0] = tb;
    }
    __syncthreads();
    const float inv = s_inv;
    float acc = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const float *z = logits + (size_t)r * C;
        const float lf = labels[r];
        const bool ok = lf >= 0.f && lf < (float)C;
        if (!ok) {  // ignored (zero gradient) or invalid (NaN gradient through inv)
            if (dlogits)
                for (int c = 0; c < C; ++c) dlogits[(size_t)r * C + c] = lf == -100.f ? 0.f : inv;
            continue;
        }
        const int y = (int)lf;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(z[c] - mx);
        const float lse = logf(den) + mx;
        acc += lse - z[y];
        if (dlogits)
            for (int c = 0; c < C; ++c) dlogits[(size_t)r * C + c] = (expf(z[c] - lse) - (c == y ? 1.f : 0.f)) * inv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
        *loss = t * inv;
    }
}

// ------------------------------------------------------------- dropout keep-mask (pre-scaled)
// Counter-based generator: Philox-like mixing of (seed, element index); the mask is a pure
// function of (seed, offset + i), so a step can be replayed.
__device__ __forceinline__ uint32_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

__global__ void dropout_mask_kernel(float *__restrict__ mask, size_t n, float p, uint64_t seed, uint64_t offset) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = mix64(seed * 0x9e3779b97f4a7c15ull + (offset + i) + 0x632be59bd9b4e019ull);
    const float f = (float)(u >> 8) * (1.0f / 16777216.0f);
    mask[i] = f >= p ? 1.f / (1.f - p) : 0.f;
}

// y[r, 0:C] (pitch y_ld) = x[r, 0:C] (pitch x_ld): writes a column slice of a wider buffer
__global__ void copy_cols_kernel(const float *__restrict__ x, int x_ld, float *__restrict__ y, int y_ld, int R, int C) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * C) return;
    const int r = (int)(idx / C), c = (int)(idx - (size_t)r * C);
    y[(size_t)r * y_ld + c] = x[(size_t)r * x_ld + c];
}

// y = x >= 0 ? x : slope * x
__global__ void leaky_relu_fwd_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, size_t n4, float slope) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = x[i];
    y[i] = make_float4(v.x >= 0.f ? v.x : v.x * slope, v.y >= 0.f ? v.y : v.y * slope, v.z >= 0.f ? v.z : v.z * slope,
                       v.w >= 0.f ? v.w : v.w * slope);
}

// CAN attention fusion gate (reference models/model.py:561-566): out = softmax(z) * c, rows of C.
// One wave per row; p is saved for the backward.
__global__ void softmax_gate_fwd_kernel(const float *__restrict__ z, const float *__restrict__ c, float *__restrict__ out,
                                        float *__restrict__ prob, int R, int C) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float *zr = z + (size_t)row * C, *cr = c + (size_t)row * C;
    float mx = -INFINITY;
    for (int i = lane; i < C; i += 64) mx = fmaxf(mx, zr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += expf(zr[i] - mx);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int i = lane; i < C; i += 64) {
        const float p = expf(zr[i] - mx) * inv;
        prob[(size_t)row * C + i] = p;
        out[(size_t)row * C + i] = p * cr[i];
    }
}

// dc = dout * p;  dz = p * (dout*c - sum(p * dout * c))
__global__ void softmax_gate_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ prob,
                                        const float *__restrict__ c, float *__restrict__ dz, float *__restrict__ dc,
                                        int R, int C) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const size_t o = (size_t)row * C;
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += prob[o + i] * dout[o + i] * c[o + i];
    s = wave_sum(s);
    for (int i = lane; i < C; i += 64) {
        const float p = prob[o + i], g = dout[o + i];
        dc[o + i] = g * p;
        dz[o + i] = p * (g * c[o + i] - s);
    }
}

}  // namespace cer

using namespace cer;

#define ST ((hipStream_t)stream)

extern "C" int cer_leaky_relu_fwd(const float *x, float *y, size_t n, float slope, void *stream) {
    if (!x || !y || n == 0 || (n & 3)) return cer_set_error(CER_ERR_INVALID_ARG, "leaky_relu_fwd: n must be a positive multiple of 4");
    CER_LAUNCH(leaky_relu_fwd_kernel, dim3(cer_blocks(n / 4, 256)), dim3(256), 0, ST, (const float4 *)x, (float4 *)y, n / 4, slope);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_softmax_gate_fwd(const float *z, const float *c, float *out, float *prob, int R, int C, void *stream) {
    if (!z || !c || !out || !prob || R <= 0 || C <= 0) return cer_set_error(CER_ERR_INVALID_ARG, "softmax_gate_fwd: bad argument");
    CER_LAUNCH(softmax_gate_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, z, c, out, prob, R, C);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_softmax_gate_bwd(const float *dout, const float *prob, const float *c, float *dz, float *dc, int R, int C,
                                    void *stream) {
    if (!dout || !prob || !c || !dz || !dc || R <= 0 || C <= 0) return cer_set_error(CER_ERR_INVALID_ARG, "softmax_gate_bwd: bad argument");
    CER_LAUNCH(softmax_gate_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, dout, prob, c, dz, dc, R, C);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_weight_norm_fwd(const float *v, const float *g, float *w, float *norm, int rows, int E, void *stream) {
    if (!v || !g || !w || !norm || rows <= 0 || E <= 0) return cer_set_error(CER_ERR_INVALID_ARG, "weight_norm_fwd: bad argument");
    CER_LAUNCH(weight_norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, v, g, w, norm, rows, E, (float *)nullptr, (float *)nullptr, 1, 0, 0);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_weight_norm_fwd_packed(const float *v, const float *g, float *w, float *norm, float *wp_fwd, float *wp_dgrad, int Cout,
                                          int Cin, int k, void *stream) {
    if (!v || !g || !w || !norm || !wp_fwd || !wp_dgrad || Cout <= 0 || Cin <= 0 || k <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "weight_norm_fwd_packed: bad argument");
    CER_LAUNCH(weight_norm_fwd_kernel, dim3((Cout + 3) / 4), dim3(256), 0, ST, v, g, w, norm, Cout, Cin * k, wp_fwd, wp_dgrad, k,
               cer_conv_kpad(k, 1, Cin), cer_conv_kpad(k, 1, Cout));
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_weight_norm_bwd(const float *dw, const float *v, const float *g, const float *norm, float *dv,
                                   float *dg, int rows, int E, void *stream) {
    if (!dw || !v || !g || !norm || !dv || !dg || rows <= 0 || E <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "weight_norm_bwd: bad argument");
    CER_LAUNCH(weight_norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, dw, v, g, norm, dv, dg, rows, E, 1);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_weight_norm_bwd_partials(const float *dw_parts, int splits, const float *v, const float *g, const float *norm, float *dv,
                                            float *dg, int rows, int E, void *stream) {
    if (!dw_parts || splits <= 0 || !v || !g || !norm || !dv || !dg || rows <= 0 || E <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "weight_norm_bwd_partials: bad argument");
    CER_LAUNCH(weight_norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, dw_parts, v, g, norm, dv, dg, rows, E, splits);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" size_t cer_col_sum_workspace_bytes(int R, int C) {
    int slabs = (R + 255) / 256;
    return slabs > 1 ? (size_t)2 * slabs * C * sizeof(float) : 0;   // two partial arrays: the paired sums below
}

// out1 / out2 = the two column sums of col_sum_pair4_kernel (dense rows, C % 4 == 0, more than one slab); false = not applicable
static bool col_sum_pair(const float *a, const float *b, const float *mean, const float *invstd, float *out1, float *out2, int R, int C,
                         void *workspace, size_t workspace_bytes, void *stream) {
    int rows_per_slab = 256;
    if ((R + 255) / 256 > 1024) rows_per_slab = ((R + 1023) / 1024 + 31) / 32 * 32;
    const int slabs = (R + rows_per_slab - 1) / rows_per_slab;
    if (slabs <= 1 || (C & 3) || !workspace || workspace_bytes < (size_t)2 * slabs * C * sizeof(float)) return false;
    float *p1 = (float *)workspace, *p2 = p1 + (size_t)slabs * C;
    CER_LAUNCH(col_sum_pair4_kernel, dim3((C + 127) / 128, slabs), dim3(256), 0, ST, a, b, mean, invstd, p1, p2, R, C, rows_per_slab);
    CER_LAUNCH(col_sum_kernel, dim3((C + 31) / 32, 1), dim3(256), 0, ST, (const float *)p1, C, (const float *)nullptr, 0,
               (const float *)nullptr, (const float *)nullptr, out1, slabs, C, slabs);
    CER_LAUNCH(col_sum_kernel, dim3((C + 31) / 32, 1), dim3(256), 0, ST, (const float *)p2, C, (const float *)nullptr, 0,
               (const float *)nullptr, (const float *)nullptr, out2, slabs, C, slabs);
    return true;
}

extern "C" int cer_col_sum(const float *a, int a_ld, const float *b, int b_ld, const float *mean,
                           const float *invstd, float *out, int R, int C, void *workspace, size_t workspace_bytes,
                           void *stream) {
    if ((!a && !(b && mean)) || !out || R <= 0 || C <= 0 || (a && a_ld < C) || (b && b_ld < C))
        return cer_set_error(CER_ERR_INVALID_ARG, "col_sum: bad argument");
    // row slabs: 256 rows each, but at most 1024 of them -- the second launch folds the slabs' partial rows with ONE block per
    // 32 columns, so 25 088 slabs (a [6.4 M x 64] tensor) would cost more there than the first pass
    int rows_per_slab = 256;
    if ((R + 255) / 256 > 1024) rows_per_slab = ((R + 1023) / 1024 + 31) / 32 * 32;
    const int slabs = (R + rows_per_slab - 1) / rows_per_slab;
    dim3 grid((C + 31) / 32, slabs);
    const bool vec4 = slabs > 1 && (C & 3) == 0 && (!a || (a_ld & 3) == 0) && (!b || (b_ld & 3) == 0);
    if (slabs == 1) {
        CER_LAUNCH(col_sum_kernel, grid, dim3(256), 0, ST, a, a_ld, b, b_ld, mean, invstd, out, R, C, rows_per_slab);
    } else {
        if (!workspace || workspace_bytes < (size_t)slabs * C * sizeof(float))
            return cer_set_error(CER_ERR_WORKSPACE, "col_sum: workspace too small");
        float *part = (float *)workspace;
        if (vec4)
            CER_LAUNCH(col_sum4_kernel, dim3((C + 127) / 128, slabs), dim3(256), 0, ST, a, a_ld, b, b_ld, mean, invstd, part, R, C,
                       rows_per_slab);
        else
            CER_LAUNCH(col_sum_kernel, grid, dim3(256), 0, ST, a, a_ld, b, b_ld, mean, invstd, part, R, C, rows_per_slab);
        CER_LAUNCH(col_sum_kernel, dim3((C + 31) / 32, 1), dim3(256), 0, ST, (const float *)part, C,
                           (const float *)nullptr, 0, (const float *)nullptr, (const float *)nullptr, out, slabs, C,
                           slabs);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_act_mask_bwd(const float *dy, const float *y, const float *mask, float *dz, size_t n, float slope,
                                void *stream) {
    if (!dy || !y || !dz || n == 0 || (n & 3)) return cer_set_error(CER_ERR_INVALID_ARG, "act_mask_bwd: n must be a positive multiple of 4");
    CER_LAUNCH(act_mask_bwd_kernel, dim3(cer_blocks(n / 4, 256)), dim3(256), 0, ST, (const float4 *)dy,
                       (const float4 *)y, (const float4 *)mask, (float4 *)dz, n / 4, slope);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_tblock_tail_bwd(const float *dout, const float *out, const float *a2, const float *mask2,
                                   float *du, float *dz2, size_t n, float slope, void *stream) {
    if (!dout || !out || !a2 || !du || !dz2 || n == 0 || (n & 3))
        return cer_set_error(CER_ERR_INVALID_ARG, "tblock_tail_bwd: n must be a positive multiple of 4");
    CER_LAUNCH(tblock_tail_bwd_kernel, dim3(cer_blocks(n / 4, 256)), dim3(256), 0, ST, (const float4 *)dout,
                       (const float4 *)out, (const float4 *)a2, (const float4 *)mask2, (float4 *)du, (float4 *)dz2,
                       n / 4, slope);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

// large R: the statistics as two deterministic column sums (the block-per-32-channels kernel above walks all R rows
// serially: 3-7 ms at the released encoder units' R = 25600 .. 102400 rows)
__global__ void bn_rows_mean_kernel(float *__restrict__ sum_to_mean, int R, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) sum_to_mean[c] /= (float)R;
}
__global__ void bn_rows_finish_kernel(const float *__restrict__ mean, float *__restrict__ ssd_to_invstd,
                                      float *__restrict__ running_mean, float *__restrict__ running_var, int R, int C,
                                      float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float var = ssd_to_invstd[c] / (float)R;   // sum (x - mean)^2
    var = var > 0.f ? var : 0.f;
    ssd_to_invstd[c] = rsqrtf(var + eps);
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (float)R / (float)max(R - 1, 1);
    }
}

// s1 = sum (x - k), s2 = sum (x - k)^2 with the shift row k  ->  mean, invstd, running buffers
__global__ void bn_rows_shifted_finish_kernel(float *__restrict__ s1_to_mean, float *__restrict__ s2_to_invstd,
                                              const float *__restrict__ shift, float *__restrict__ running_mean,
                                              float *__restrict__ running_var, int R, int C, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double d1 = (double)s1_to_mean[c] / R, d2 = (double)s2_to_invstd[c] / R;
    double var = d2 - d1 * d1;
    var = var > 0.0 ? var : 0.0;
    const float mean = (float)((double)shift[c] + d1);
    s1_to_mean[c] = mean;
    s2_to_invstd[c] = rsqrtf((float)var + eps);
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)var * (float)R / (float)max(R - 1, 1);
}

extern "C" size_t cer_bn_rows_fwd_workspace_bytes(int R, int C) { return R > 2048 ? cer_col_sum_workspace_bytes(R, C) : 0; }

extern "C" int cer_bn_rows_fwd(const float *x, int x_ld, const float *w, const float *b, float *running_mean,
                               float *running_var, float *save_mean, float *save_invstd, float *y, int y_ld, int R,
                               int C, int train, float eps, float momentum, void *workspace, size_t workspace_bytes,
                               void *stream) {
    // y == NULL (train mode only): statistics pass alone -- save_mean / save_invstd and the running-buffer update, no output
    if (!x || R <= 0 || C <= 0 || x_ld < C || !running_mean || !running_var || (y ? (!w || !b || y_ld < C) : !train))
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_rows_fwd: bad argument");
    const size_t n = (size_t)R * C;
    if (train) {
        if (!save_mean || !save_invstd) return cer_set_error(CER_ERR_INVALID_ARG, "bn_rows_fwd: train needs save buffers");
        if (R > 2048 && x_ld == C &&
            col_sum_pair(nullptr, x, x, nullptr, save_mean, save_invstd, R, C, workspace, workspace_bytes, stream)) {
            // ONE pass: sum (x - k) and sum (x - k)^2 about the shift k[c] = x[0][c], the column's own first value (any k is
            // exact algebra; a sample of the column lies within a few sigma of its mean, so E[d^2] - E[d]^2 does not cancel
            // whatever the column's offset -- the running mean would do only once it has converged)
            CER_LAUNCH(bn_rows_shifted_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, save_mean, save_invstd, x, running_mean,
                       running_var, R, C, eps, momentum);
        } else if (R > 2048) {
            int rc = cer_col_sum(x, x_ld, nullptr, 0, nullptr, nullptr, save_mean, R, C, workspace, workspace_bytes, stream);
            if (rc) return rc;
            CER_LAUNCH(bn_rows_mean_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, save_mean, R, C);
            rc = cer_col_sum(nullptr, 0, x, x_ld, save_mean, nullptr, save_invstd, R, C, workspace, workspace_bytes, stream);
            if (rc) return rc;
            CER_LAUNCH(bn_rows_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, (const float *)save_mean, save_invstd,
                       running_mean, running_var, R, C, eps, momentum);
        } else {
            CER_LAUNCH(bn_rows_stats_kernel, dim3((C + 31) / 32), dim3(256), 0, ST, x, x_ld, save_mean, save_invstd,
                               running_mean, running_var, R, C, eps, momentum);
        }
        if (y)
            CER_LAUNCH(bn_rows_apply_kernel, dim3(cer_blocks(n, 256)), dim3(256), 0, ST, x, x_ld,
                               (const float *)save_mean, (const float *)save_invstd, 0, eps, w, b, y, y_ld, R, C);
    } else {
        CER_LAUNCH(bn_rows_apply_kernel, dim3(cer_blocks(n, 256)), dim3(256), 0, ST, x, x_ld,
                           (const float *)running_mean, (const float *)running_var, 1, eps, w, b, y, y_ld, R, C);
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_bn_rows_bwd(const float *dy, int dy_ld, const float *x, int x_ld, const float *save_mean,
                               const float *save_invstd, const float *w, float *dx, float *dw, float *db, int R,
                               int C, int train, void *workspace, size_t workspace_bytes, void *stream) {
    if (!dy || !x || !save_mean || !save_invstd || !w || !dx || !dw || !db || R <= 0 || C <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "bn_rows_bwd: bad argument");
    if (!(dy_ld == C && x_ld == C && col_sum_pair(dy, x, save_mean, save_invstd, db, dw, R, C, workspace, workspace_bytes, stream))) {
        int rc = cer_col_sum(dy, dy_ld, nullptr, 0, nullptr, nullptr, db, R, C, workspace, workspace_bytes, stream);
        if (rc) return rc;
        rc = cer_col_sum(dy, dy_ld, x, x_ld, save_mean, save_invstd, dw, R, C, workspace, workspace_bytes, stream);
        if (rc) return rc;
    }
    CER_LAUNCH(bn_rows_bwd_kernel, dim3(cer_blocks((size_t)R * C, 256)), dim3(256), 0, ST, dy, dy_ld, x, x_ld,
                       save_mean, save_invstd, w, (const float *)db, (const float *)dw, dx, R, C, train);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

template <int HD>
static void launch_attn(bool bwd, const FusionPtrs &P, float *vals, const float *dvals, float *probs, int R, int H,
                        int M, hipStream_t st) {
    const size_t threads = (size_t)R * H * HD;
    dim3 grid(cer_blocks(threads, 256)), block(256);
    if (!bwd) CER_LAUNCH(lfan_attn_fwd_kernel<HD>, grid, block, 0, st, P, vals, probs, R, H, M);
    else CER_LAUNCH(lfan_attn_bwd_kernel<HD>, grid, block, 0, st, P, dvals, (const float *)probs, R, H, M);
}

static int attn_dispatch(bool bwd, const float *const *qkv, float *const *dqkv, float *vals, const float *dvals,
                         float *probs, int R, int H, int M, int hd, void *stream) {
    if (R <= 0 || H <= 0 || M <= 0 || M > MAXM || !probs)
        return cer_set_error(CER_ERR_INVALID_ARG, "lfan_attn: bad argument (1 <= modalities <= 4)");
    FusionPtrs P{};
    for (int m = 0; m < M; ++m) {
        if (!qkv[m] || (bwd && !dqkv[m])) return cer_set_error(CER_ERR_INVALID_ARG, "lfan_attn: NULL modality pointer");
        P.qkv[m] = qkv[m];
        P.dqkv[m] = bwd ? dqkv[m] : nullptr;
    }
    switch (hd) {
        case 8: launch_attn<8>(bwd, P, vals, dvals, probs, R, H, M, ST); break;
        case 16: launch_attn<16>(bwd, P, vals, dvals, probs, R, H, M, ST); break;
        case 32: launch_attn<32>(bwd, P, vals, dvals, probs, R, H, M, ST); break;
        case 64: launch_attn<64>(bwd, P, vals, dvals, probs, R, H, M, ST); break;
        default: return cer_set_error(CER_ERR_UNSUPPORTED, "lfan_attn: head dim must be 8, 16, 32 or 64");
    }
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_lfan_attn_fwd(const float *const *qkv, float *vals, float *probs, int R, int H, int M, int hd,
                                 void *stream) {
    if (!qkv || !vals) return cer_set_error(CER_ERR_INVALID_ARG, "lfan_attn_fwd: NULL argument");
    return attn_dispatch(false, qkv, nullptr, vals, nullptr, probs, R, H, M, hd, stream);
}

extern "C" int cer_lfan_attn_bwd(const float *const *qkv, const float *dvals, const float *probs,
                                 float *const *dqkv, int R, int H, int M, int hd, void *stream) {
    if (!qkv || !dvals || !dqkv) return cer_set_error(CER_ERR_INVALID_ARG, "lfan_attn_bwd: NULL argument");
    return attn_dispatch(true, qkv, dqkv, nullptr, dvals, const_cast<float *>(probs), R, H, M, hd, stream);
}

extern "C" int cer_layernorm_fwd(const float *x, const float *mask, const float *gamma, const float *beta, float *y,
                                 int y_ld, float *save_mean, float *save_rstd, int R, int C, float eps, void *stream) {
    if (!x || !gamma || !beta || !y || R <= 0 || C <= 0 || y_ld < C)
        return cer_set_error(CER_ERR_INVALID_ARG, "layernorm_fwd: bad argument");
    CER_LAUNCH(layernorm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, x, mask, gamma, beta, y, y_ld,
                       save_mean, save_rstd, R, C, eps);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_layernorm_bwd(const float *dy, int dy_ld, const float *x, const float *mask, const float *gamma,
                                 const float *save_mean, const float *save_rstd, float *dx, float *dgamma,
                                 float *dbeta, float *scratch, int R, int C, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    if (!dy || !x || !gamma || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !scratch || R <= 0 || C <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "layernorm_bwd: bad argument (scratch must hold R*C floats)");
    CER_LAUNCH(layernorm_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, dy, dy_ld, x, mask, gamma, save_mean,
                       save_rstd, dx, scratch, R, C);
    CER_HIP_CHECK(hipGetLastError());
    int rc = cer_col_sum(scratch, C, nullptr, 0, nullptr, nullptr, dgamma, R, C, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return cer_col_sum(dy, dy_ld, nullptr, 0, nullptr, nullptr, dbeta, R, C, workspace, workspace_bytes, stream);
}

extern "C" int cer_cross_entropy(const float *logits, const float *labels, float *loss, float *dlogits, int *bad_labels, int R,
                                 int C, void *stream) {
    if (!logits || !labels || !loss || R <= 0 || C <= 0) return cer_set_error(CER_ERR_INVALID_ARG, "cross_entropy: bad argument");
    CER_LAUNCH(cross_entropy_kernel, dim3(1), dim3(1024), 0, ST, logits, labels, loss, dlogits, bad_labels, R, C);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_dropout_mask(float *mask, size_t n, float p, uint64_t seed, uint64_t offset, void *stream) {
    if (!mask || n == 0 || !(p >= 0.f && p < 1.f)) return cer_set_error(CER_ERR_INVALID_ARG, "dropout_mask: need 0 <= p < 1");
    CER_LAUNCH(dropout_mask_kernel, dim3(cer_blocks(n, 256)), dim3(256), 0, ST, mask, n, p, seed, offset);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_copy_cols(const float *x, int x_ld, float *y, int y_ld, int R, int C, void *stream) {
    if (!x || !y || R <= 0 || C <= 0 || x_ld < C || y_ld < C) return cer_set_error(CER_ERR_INVALID_ARG, "copy_cols: bad argument");
    CER_LAUNCH(copy_cols_kernel, dim3(cer_blocks((size_t)R * C, 256)), dim3(256), 0, ST, x, x_ld, y, y_ld, R, C);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}


// (used by cer_bn_rows_bwd_split in conv_b3.hip) db = sum dy, dw = sum dy * x_hat in one pass when the shape allows, else two
extern "C" int cer_bn_bwd_sums(const float *dy, const float *x, const float *save_mean, const float *save_invstd, float *db, float *dw,
                               int R, int C, void *workspace, size_t workspace_bytes, void *stream) {
    if (!dy || !x || !save_mean || !save_invstd || !db || !dw || R <= 0 || C <= 0) return cer_set_error(CER_ERR_INVALID_ARG, "bn_bwd_sums: bad argument");
    if (col_sum_pair(dy, x, save_mean, save_invstd, db, dw, R, C, workspace, workspace_bytes, stream)) {
        CER_HIP_CHECK(hipGetLastError());
        return CER_OK;
    }
    int rc = cer_col_sum(dy, C, nullptr, 0, nullptr, nullptr, db, R, C, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return cer_col_sum(dy, C, x, C, save_mean, save_invstd, dw, R, C, workspace, workspace_bytes, stream);
}
