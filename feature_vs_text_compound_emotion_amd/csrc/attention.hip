// attention.hip -- softmax(Q K^T * scale + mask) V on the fp32 matrix cores, flash style.
//
// Used by the BERT text encoder (12 heads x d=64, key-padding mask; the arithmetic the reference
// obtains from transformers' BertSelfAttention, call site abaw5_pre_processing/base/speech.py:603-606)
// and by the JMT/MT heads (nn.MultiheadAttention(128, 1 head), reference models/model.py:716-750,
// 967-972 -- including the final self-attention over L*B tokens).
//
// One wave owns 32 queries; a block of 4 waves shares the K/V tiles (32 keys) staged in LDS.
// Orientation is chosen so that NO cross-lane traffic is needed between the two products:
//   S^T = K Q^T   : D[i = key][j = query]  -> a lane holds 16 keys of ITS query (the partner lane
//                                             l^32 holds the other 16), so the softmax statistics
//                                             are 16 register ops + one shuffle;
//   O^T += V^T P^T: B[k = key][j = query] is exactly register r of S^T's accumulator when the
//                   A operand V^T[d][key] is read with the same (register, lane-half) -> key map.
// Everything is exact fp32 (v_mfma_f32_32x32x2_f32) with full-precision expf.
//
// SPLIT variants (round 3): when (Sq / 128) * H * B blocks cannot cover the chip -- the JMT / MT final stage is 1024 tokens x
// 6 stacks x 1 head = 48 blocks on 256 CUs -- the four waves of a block share the SAME 32 owner rows and split the streamed
// range between them (tile t goes to wave t % 4, each wave stages its own tiles in its own LDS region), so the grid has
// four times the blocks; the waves' partial results are merged in LDS in the fixed order wave 0..3 (forward: the usual
// flash rescale by exp(m_w - m); backward: plain sums, P is recomputed from the saved LSE) -- deterministic, no atomics,
// no second launch.
#include <math.h>

#include "cer_internal.h"

namespace cer {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
    const float *q, *k, *v;
    const int *key_mask;  // [B][Sk] 1 = attend, or NULL
    float *out;
    float *lse;  // optional [B][H][Sq]: log-sum-exp of the scaled, masked scores (saved for the backward)
    long long q_sb, q_ss, q_sh;  // element strides: batch, token, head (d is contiguous)
    long long k_sb, k_ss, k_sh;
    long long v_sb, v_ss, v_sh;
    long long o_sb, o_ss, o_sh;
    int B, H, Sq, Sk;
    float scale;
};

template <int D, bool SPLIT>
__global__ __launch_bounds__(256) void attention_fwd_kernel(AttnArgs p) {
    constexpr int KP = D + 4;   // K tile pitch: conflict-free ds_read_b128 over 16 keys
    constexpr int NG = D / 8;   // 8-wide k groups of the QK^T reduction
    constexpr int NT = D / 32;  // 32-row tiles of O^T
    constexpr int TILE = 32 * KP + 32 * D;      // floats of one K + V tile pair
    extern __shared__ __attribute__((aligned(16))) float attn_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *Ks = attn_smem + (SPLIT ? wave * TILE : 0), *Vs = Ks + 32 * KP;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int query = SPLIT ? blockIdx.x * 32 + l31 : blockIdx.x * 128 + wave * 32 + l31;
    const bool qok = query < p.Sq;

    // Q fragment (B operand of K Q^T), pre-scaled: lane half `half` holds dk = 8g + 4*half + e
    float4 qf[NG];
    {
        const float *qp = p.q + b * p.q_sb + (long long)(qok ? query : 0) * p.q_ss + h * p.q_sh + 4 * half;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 t = qok ? *reinterpret_cast<const float4 *>(qp + 8 * g) : make_float4(0, 0, 0, 0);
            qf[g] = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);
        }
    }
    f32x16 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const float *kbase = p.k + b * p.k_sb + h * p.k_sh;
    const float *vbase = p.v + b * p.v_sb + h * p.v_sh;
    const int *mbase = p.key_mask ? p.key_mask + (size_t)b * p.Sk : nullptr;

    // SPLIT: wave w takes tiles w, w + 4, ... (uniform trip count; tiles past the end are fully masked)
    const int kstep = SPLIT ? 128 : 32;
    // K / V tiles travel global -> registers -> LDS; the NEXT tile's loads are issued before the current tile's MFMAs, so
    // their latency (one wave per SIMD: nothing else would cover it) runs under ~128 matrix instructions
    constexpr int NLD = 32 * (D / 4) / (SPLIT ? 64 : 256);      // float4 per lane, tile and operand
    float4 pk[NLD], pv[NLD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int i = (SPLIT ? lane : tid) + n * (SPLIT ? 64 : 256);
            const int key = i / (D / 4), c4 = (i - key * (D / 4)) * 4;
            pk[n] = pv[n] = make_float4(0, 0, 0, 0);
            if (k0 + key < p.Sk) {
                pk[n] = *reinterpret_cast<const float4 *>(kbase + (long long)(k0 + key) * p.k_ss + c4);
                pv[n] = *reinterpret_cast<const float4 *>(vbase + (long long)(k0 + key) * p.v_ss + c4);
            }
        }
    };
    fetch(SPLIT ? 32 * wave : 0);
    for (int kb = 0; kb < p.Sk; kb += kstep) {
        const int k0 = SPLIT ? kb + 32 * wave : kb;
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int i = (SPLIT ? lane : tid) + n * (SPLIT ? 64 : 256);
            const int key = i / (D / 4), c4 = (i - key * (D / 4)) * 4;
            *reinterpret_cast<float4 *>(&Ks[key * KP + c4]) = pk[n];
            *reinterpret_cast<float4 *>(&Vs[key * D + c4]) = pv[n];
        }
        __syncthreads();
        if (kb + kstep < p.Sk) fetch(k0 + kstep);

        // S^T tile = K Q^T
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float *kr = &Ks[l31 * KP + 4 * half];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 a = *reinterpret_cast<const float4 *>(kr + 8 * g);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[g].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[g].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[g].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[g].w, s, 0, 0, 0);
        }
        // mask + online softmax; register r <-> key (r&3) + 8*(r>>2) + 4*half
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool ok = key < p.Sk && (!mbase || mbase[key] != 0);
            s[r] = ok ? s[r] : -INFINITY;
            mt = fmaxf(mt, s[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = (m_new == -INFINITY) ? 1.f : expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = (m_new == -INFINITY) ? 0.f : expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // O^T += V^T P^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * D + 32 * t + l31], s[r], o[t], 0, 0, 0);
        }
    }
    if constexpr (SPLIT) {
        // merge the four waves' (m, l, O) in the fixed order 0..3; partials live where the tiles were: [wave][j][lane]
        __syncthreads();
        float *part = attn_smem + wave * TILE;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(t * 16 + r) * 64 + lane] = o[t][r];
        part[NT * 16 * 64 + lane] = m_run;
        part[(NT * 16 + 1) * 64 + lane] = l_run;
        __syncthreads();
        if (wave != 0) return;
        float m = m_run;
#pragma unroll
        for (int w = 1; w < 4; ++w) m = fmaxf(m, attn_smem[w * TILE + NT * 16 * 64 + lane]);
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float *pw = attn_smem + w * TILE;
            const float mw = pw[NT * 16 * 64 + lane];
            const float sc = (mw == -INFINITY) ? 0.f : expf(mw - m);
            l += pw[(NT * 16 + 1) * 64 + lane] * sc;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] += pw[(t * 16 + r) * 64 + lane] * sc;
        }
        m_run = m;
        l_run = l;
    }
    if (!qok) return;
    if (p.lse && half == 0) p.lse[((size_t)b * p.H + h) * p.Sq + query] = l_run > 0.f ? m_run + logf(l_run) : INFINITY;
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float *op = p.out + b * p.o_sb + (long long)query * p.o_ss + h * p.o_sh;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int d = 32 * t + 8 * qd + 4 * half;
            *reinterpret_cast<float4 *>(op + d) = make_float4(o[t][4 * qd] * inv, o[t][4 * qd + 1] * inv,
                                                               o[t][4 * qd + 2] * inv, o[t][4 * qd + 3] * inv);
        }
}

// ------------------------------------------------------------------ backward
// One templated kernel, two roles (see the header comment for the orientation trick):
//   DQ mode : a wave OWNS 32 queries (Q, dO in registers; LSE, Delta per lane) and streams key
//             tiles (K, V in LDS):      dQ^T += K^T dS^T
//   DKV mode: a wave OWNS 32 keys (K, V in registers) and streams query tiles (Q, dO in LDS; LSE,
//             Delta per register row):  dV^T += dO^T P,  dK^T += Q^T dS
// with S = (owner1 . streamed1) * scale, P = exp(S - LSE[query]), dP = owner2 . streamed2,
// dS = P * (dP - Delta[query]) * scale.  Both modes recompute S/P, so no atomics are needed and the
// result is deterministic.  Delta = rowsum(dO * O) comes from attn_delta_kernel.
struct AttnBwdArgs {
    const float *q, *k, *v, *dout, *lse, *delta;
    const int *key_mask;
    float *dq, *dk, *dv;
    long long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh;  // o_* = dout strides
    long long dq_sb, dq_ss, dq_sh, dk_sb, dk_ss, dk_sh, dv_sb, dv_ss, dv_sh;
    int B, H, Sq, Sk;
    float scale;
};

// delta[b][h][q] = sum_d dO * O   (one wave per (b, h, q) row)
__global__ void attn_delta_kernel(const float *__restrict__ out, const float *__restrict__ dout, float *__restrict__ delta,
                                  long long a_sb, long long a_ss, long long a_sh, long long o_sb, long long o_ss,
                                  long long o_sh, int B, int H, int Sq, int D) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * H * Sq) return;
    const int q = row % Sq, h = (row / Sq) % H, b = row / (Sq * H);
    const float *op = out + b * a_sb + (long long)q * a_ss + h * a_sh;
    const float *gp = dout + b * o_sb + (long long)q * o_ss + h * o_sh;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += op[d] * gp[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) delta[row] = s;
}

template <int D, bool DKV, bool SPLIT>
__global__ __launch_bounds__(256, 1) void attention_bwd_kernel(AttnBwdArgs p) {
    constexpr int KP = D + 4, NG = D / 8, NT = D / 32;
    constexpr int TILE = 2 * 32 * KP + 64;                      // floats of one (T1, T2, sl, sd) set
    extern __shared__ __attribute__((aligned(16))) float attn_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *T1 = attn_smem + (SPLIT ? wave * TILE : 0);          // streamed tile 1 (K or Q), pitch KP
    float *T2 = T1 + 32 * KP;                                   // streamed tile 2 (V or dO)
    float *sl = T2 + 32 * KP, *sd = sl + 32;                    // DKV: LSE / Delta of the streamed queries
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int So = DKV ? p.Sk : p.Sq, Ss = DKV ? p.Sq : p.Sk;  // owner / streamed lengths
    const int orow = SPLIT ? blockIdx.x * 32 + l31 : blockIdx.x * 128 + wave * 32 + l31;
    const bool ook = orow < So;
    const float *o1 = DKV ? p.k + b * p.k_sb + h * p.k_sh : p.q + b * p.q_sb + h * p.q_sh;
    const float *o2 = DKV ? p.v + b * p.v_sb + h * p.v_sh : p.dout + b * p.o_sb + h * p.o_sh;
    const long long o1s = DKV ? p.k_ss : p.q_ss, o2s = DKV ? p.v_ss : p.o_ss;
    const float *s1 = DKV ? p.q + b * p.q_sb + h * p.q_sh : p.k + b * p.k_sb + h * p.k_sh;
    const float *s2 = DKV ? p.dout + b * p.o_sb + h * p.o_sh : p.v + b * p.v_sb + h * p.v_sh;
    const long long s1s = DKV ? p.q_ss : p.k_ss, s2s = DKV ? p.o_ss : p.v_ss;
    const float *lse = p.lse + ((size_t)b * p.H + h) * p.Sq, *delta = p.delta + ((size_t)b * p.H + h) * p.Sq;
    const int *mbase = p.key_mask ? p.key_mask + (size_t)b * p.Sk : nullptr;

    float4 r1[NG], r2[NG];  // owner fragments: lane half `half` holds d = 8g + 4*half + e
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        r1[g] = ook ? *reinterpret_cast<const float4 *>(o1 + (long long)orow * o1s + 8 * g + 4 * half) : make_float4(0, 0, 0, 0);
        r2[g] = ook ? *reinterpret_cast<const float4 *>(o2 + (long long)orow * o2s + 8 * g + 4 * half) : make_float4(0, 0, 0, 0);
    }
    float my_lse = INFINITY, my_delta = 0.f;
    bool my_keyok = true;
    if (!DKV) {
        if (ook) { my_lse = lse[orow]; my_delta = delta[orow]; }
    } else {
        my_keyok = ook && (!mbase || mbase[orow] != 0);
    }
    f32x16 accA[NT], accB[DKV ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accA[t][r] = 0.f;
            if constexpr (DKV) accB[t][r] = 0.f;
        }

    const int sstep = SPLIT ? 128 : 32;
    constexpr int NLD = 32 * (D / 4) / (SPLIT ? 64 : 256);      // float4 per lane, tile and operand (prefetched one tile ahead)
    float4 pa[NLD], pc[NLD];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int i = (SPLIT ? lane : tid) + n * (SPLIT ? 64 : 256);
            const int row = i / (D / 4), c4 = (i - row * (D / 4)) * 4;
            pa[n] = pc[n] = make_float4(0, 0, 0, 0);
            if (t0 + row < Ss) {
                pa[n] = *reinterpret_cast<const float4 *>(s1 + (long long)(t0 + row) * s1s + c4);
                pc[n] = *reinterpret_cast<const float4 *>(s2 + (long long)(t0 + row) * s2s + c4);
            }
        }
    };
    // (split variant at d = 128: owner fragments 128 + accumulators 128-192 registers leave no room for 128 more of prefetch
    // -- the compiler spilled 47 registers -- so there the tile is fetched where it is used)
    constexpr bool PF = !(SPLIT && D == 128);
    if constexpr (PF) fetch(SPLIT ? 32 * wave : 0);
    for (int tb = 0; tb < Ss; tb += sstep) {
        const int t0 = SPLIT ? tb + 32 * wave : tb;
        __syncthreads();
        if constexpr (PF) {
#pragma unroll
            for (int n = 0; n < NLD; ++n) {
                const int i = (SPLIT ? lane : tid) + n * (SPLIT ? 64 : 256);
                const int row = i / (D / 4), c4 = (i - row * (D / 4)) * 4;
                *reinterpret_cast<float4 *>(&T1[row * KP + c4]) = pa[n];
                *reinterpret_cast<float4 *>(&T2[row * KP + c4]) = pc[n];
            }
        } else {
            for (int i = SPLIT ? lane : tid; i < 32 * (D / 4); i += SPLIT ? 64 : 256) {
                const int row = i / (D / 4), c4 = (i - row * (D / 4)) * 4;
                float4 a = make_float4(0, 0, 0, 0), c = make_float4(0, 0, 0, 0);
                if (t0 + row < Ss) {
                    a = *reinterpret_cast<const float4 *>(s1 + (long long)(t0 + row) * s1s + c4);
                    c = *reinterpret_cast<const float4 *>(s2 + (long long)(t0 + row) * s2s + c4);
                }
                *reinterpret_cast<float4 *>(&T1[row * KP + c4]) = a;
                *reinterpret_cast<float4 *>(&T2[row * KP + c4]) = c;
            }
        }
        if (DKV && (SPLIT ? lane : tid) < 32) {
            const int j = SPLIT ? lane : tid;
            const bool ok = t0 + j < p.Sq;
            sl[j] = ok ? lse[t0 + j] : INFINITY;
            sd[j] = ok ? delta[t0 + j] : 0.f;
        }
        __syncthreads();
        if constexpr (PF) {
            if (tb + sstep < Ss) fetch(t0 + sstep);
        }
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const float *a1 = &T1[l31 * KP + 4 * half], *a2 = &T2[l31 * KP + 4 * half];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 x = *reinterpret_cast<const float4 *>(a1 + 8 * g);
            const float4 y = *reinterpret_cast<const float4 *>(a2 + 8 * g);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, r1[g].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, r1[g].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, r1[g].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, r1[g].w, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(y.x, r2[g].x, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(y.y, r2[g].y, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(y.z, r2[g].z, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(y.w, r2[g].w, dp, 0, 0, 0);
        }
        // register r <-> streamed row (r&3) + 8*(r>>2) + 4*half; lane <-> owner row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int srow = (r & 3) + 8 * (r >> 2) + 4 * half;
            float l, dl;
            bool ok;
            if (DKV) {  // streamed = query, owner = key
                l = sl[srow]; dl = sd[srow];
                ok = my_keyok && (t0 + srow < p.Sq);
            } else {    // streamed = key, owner = query
                l = my_lse; dl = my_delta;
                const int key = t0 + srow;
                ok = ook && key < p.Sk && (!mbase || mbase[key] != 0);
            }
            const float pr = ok ? expf(s[r] * p.scale - l) : 0.f;
            s[r] = pr;                                   // P
            dp[r] = pr * (dp[r] - dl) * p.scale;         // dS (already times the score scale)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int srow = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                accA[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(T1[srow * KP + 32 * t + l31], dp[r], accA[t], 0, 0, 0);
                if constexpr (DKV) accB[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(T2[srow * KP + 32 * t + l31], s[r], accB[t], 0, 0, 0);
            }
        }
    }
    if constexpr (SPLIT) {
        // sum the four waves' partial gradients in the fixed order 0..3 (partials stored where the tiles were)
        __syncthreads();
        float *pa = attn_smem + wave * TILE, *pb = pa + 32 * KP;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pa[(t * 16 + r) * 64 + lane] = accA[t][r];
                if constexpr (DKV) pb[(t * 16 + r) * 64 + lane] = accB[t][r];
            }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = 0.f, c = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    a += attn_smem[w * TILE + (t * 16 + r) * 64 + lane];
                    if constexpr (DKV) c += attn_smem[w * TILE + 32 * KP + (t * 16 + r) * 64 + lane];
                }
                accA[t][r] = a;
                if constexpr (DKV) accB[t][r] = c;
            }
    }
    if (!ook) return;
    // accA = (DQ: dQ^T | DKV: dK^T), accB = dV^T; layout [d][owner row]
    float *oa = DKV ? p.dk + b * p.dk_sb + (long long)orow * p.dk_ss + h * p.dk_sh
                    : p.dq + b * p.dq_sb + (long long)orow * p.dq_ss + h * p.dq_sh;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int d = 32 * t + 8 * qd + 4 * half;
            *reinterpret_cast<float4 *>(oa + d) = make_float4(accA[t][4 * qd], accA[t][4 * qd + 1], accA[t][4 * qd + 2], accA[t][4 * qd + 3]);
        }
    if constexpr (DKV) {
        float *ob = p.dv + b * p.dv_sb + (long long)orow * p.dv_ss + h * p.dv_sh;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int d = 32 * t + 8 * qd + 4 * half;
                *reinterpret_cast<float4 *>(ob + d) = make_float4(accB[t][4 * qd], accB[t][4 * qd + 1], accB[t][4 * qd + 2], accB[t][4 * qd + 3]);
            }
    }
}

}  // namespace cer

using namespace cer;

// the four waves of a block share their owner rows and split the streamed range when the plain grid ((rows / 128) * H * B
// blocks) would leave more than half of the 256 CUs idle and there are enough streamed tiles to split
static bool attn_use_split(int owner_rows, int streamed_rows, int H, int B) {
    const long long blocks = (long long)((owner_rows + 127) / 128) * H * B;
    return blocks < 128 && streamed_rows >= 128;
}

template <int D, bool SPLIT>
static int launch_attn_fwd(const AttnArgs &a, hipStream_t st) {
    constexpr int TILE = 32 * (D + 4) + 32 * D;
    // SPLIT: four private tile pairs; they are reused for the partial results (D / 2 floats per lane + m + l)
    const size_t lds = (size_t)(SPLIT ? 4 * TILE : TILE) * sizeof(float);
    static_assert(!SPLIT || TILE >= (D / 2 + 2) * 64, "partials must fit the tile region");
    auto k = attention_fwd_kernel<D, SPLIT>;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(SPLIT ? (a.Sq + 31) / 32 : (a.Sq + 127) / 128, a.H, a.B);
    CER_LAUNCH(k, grid, dim3(256), lds, st, a);
    return CER_OK;
}

extern "C" int cer_attention_fwd(const float *q, const float *k, const float *v, const int *key_mask, float *out,
                                 float *lse, int B, int H, int Sq, int Sk, int D, const long long *q_strides,
                                 const long long *k_strides, const long long *v_strides, const long long *o_strides,
                                 float scale, void *stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || !q_strides || !k_strides || !v_strides ||
        !o_strides)
        return cer_set_error(CER_ERR_INVALID_ARG, "attention_fwd: bad argument");
    for (int i = 0; i < 3; ++i)
        if ((q_strides[i] & 3) || (k_strides[i] & 3) || (v_strides[i] & 3) || (o_strides[i] & 3))
            return cer_set_error(CER_ERR_INVALID_ARG, "attention_fwd: strides must be multiples of 4 floats");
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.key_mask = key_mask; a.out = out; a.lse = lse;
    a.q_sb = q_strides[0]; a.q_ss = q_strides[1]; a.q_sh = q_strides[2];
    a.k_sb = k_strides[0]; a.k_ss = k_strides[1]; a.k_sh = k_strides[2];
    a.v_sb = v_strides[0]; a.v_ss = v_strides[1]; a.v_sh = v_strides[2];
    a.o_sb = o_strides[0]; a.o_ss = o_strides[1]; a.o_sh = o_strides[2];
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale;
    if (D != 32 && D != 64 && D != 128) return cer_set_error(CER_ERR_UNSUPPORTED, "attention_fwd: head dim must be 32, 64 or 128");
    const bool split = attn_use_split(Sq, Sk, H, B);
    int rc;
    if (D == 32) rc = split ? launch_attn_fwd<32, true>(a, (hipStream_t)stream) : launch_attn_fwd<32, false>(a, (hipStream_t)stream);
    else if (D == 64) rc = split ? launch_attn_fwd<64, true>(a, (hipStream_t)stream) : launch_attn_fwd<64, false>(a, (hipStream_t)stream);
    else rc = split ? launch_attn_fwd<128, true>(a, (hipStream_t)stream) : launch_attn_fwd<128, false>(a, (hipStream_t)stream);
    if (rc != CER_OK) return rc;
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

template <int D, bool DKV, bool SPLIT>
static int launch_attn_bwd_one(const AttnBwdArgs &a, hipStream_t st) {
    constexpr int TILE = 2 * 32 * (D + 4) + 64;
    static_assert(!SPLIT || 32 * (D + 4) >= (D / 2) * 64, "a partial gradient must fit one tile");
    const size_t lds = (size_t)(SPLIT ? 4 * TILE : TILE) * sizeof(float);
    auto k = attention_bwd_kernel<D, DKV, SPLIT>;
    if (lds > 64 * 1024) CER_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int owner = DKV ? a.Sk : a.Sq;
    CER_LAUNCH(k, dim3(SPLIT ? (owner + 31) / 32 : (owner + 127) / 128, a.H, a.B), dim3(256), lds, st, a);
    return CER_OK;
}

template <int D>
static int launch_attn_bwd(const AttnBwdArgs &a, hipStream_t st) {
    int rc = attn_use_split(a.Sq, a.Sk, a.H, a.B) ? launch_attn_bwd_one<D, false, true>(a, st) : launch_attn_bwd_one<D, false, false>(a, st);
    if (rc != CER_OK) return rc;
    return attn_use_split(a.Sk, a.Sq, a.H, a.B) ? launch_attn_bwd_one<D, true, true>(a, st) : launch_attn_bwd_one<D, true, false>(a, st);
}

extern "C" int cer_attention_bwd(const float *q, const float *k, const float *v, const float *out, const float *dout,
                                 const float *lse, const int *key_mask, float *delta, float *dq, float *dk, float *dv,
                                 int B, int H, int Sq, int Sk, int D, const long long *q_strides,
                                 const long long *k_strides, const long long *v_strides, const long long *o_strides,
                                 const long long *do_strides, const long long *dq_strides, const long long *dk_strides,
                                 const long long *dv_strides, float scale, void *stream) {
    if (!q || !k || !v || !out || !dout || !lse || !delta || !dq || !dk || !dv || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 ||
        !q_strides || !k_strides || !v_strides || !o_strides || !do_strides || !dq_strides || !dk_strides || !dv_strides)
        return cer_set_error(CER_ERR_INVALID_ARG, "attention_bwd: bad argument");
    const long long *all[] = {q_strides, k_strides, v_strides, o_strides, do_strides, dq_strides, dk_strides, dv_strides};
    for (const long long *st : all)
        for (int i = 0; i < 3; ++i)
            if (st[i] & 3) return cer_set_error(CER_ERR_INVALID_ARG, "attention_bwd: strides must be multiples of 4 floats");
    if (D != 32 && D != 64 && D != 128) return cer_set_error(CER_ERR_UNSUPPORTED, "attention_bwd: head dim must be 32, 64 or 128");
    hipStream_t st = (hipStream_t)stream;
    CER_LAUNCH(attn_delta_kernel, dim3((B * H * Sq + 3) / 4), dim3(256), 0, st, out, dout, delta, o_strides[0], o_strides[1],
               o_strides[2], do_strides[0], do_strides[1], do_strides[2], B, H, Sq, D);
    AttnBwdArgs a{};
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.lse = lse; a.delta = delta; a.key_mask = key_mask;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.q_sb = q_strides[0]; a.q_ss = q_strides[1]; a.q_sh = q_strides[2];
    a.k_sb = k_strides[0]; a.k_ss = k_strides[1]; a.k_sh = k_strides[2];
    a.v_sb = v_strides[0]; a.v_ss = v_strides[1]; a.v_sh = v_strides[2];
    a.o_sb = do_strides[0]; a.o_ss = do_strides[1]; a.o_sh = do_strides[2];
    a.dq_sb = dq_strides[0]; a.dq_ss = dq_strides[1]; a.dq_sh = dq_strides[2];
    a.dk_sb = dk_strides[0]; a.dk_ss = dk_strides[1]; a.dk_sh = dk_strides[2];
    a.dv_sb = dv_strides[0]; a.dv_ss = dv_strides[1]; a.dv_sh = dv_strides[2];
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale;
    const int rc = D == 32 ? launch_attn_bwd<32>(a, st) : D == 64 ? launch_attn_bwd<64>(a, st) : launch_attn_bwd<128>(a, st);
    if (rc != CER_OK) return rc;
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
