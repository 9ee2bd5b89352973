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
#include <math.h>

#include "cer_internal.h"

namespace cer {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
    const float *q, *k, *v;
    const int *key_mask;  // [B][Sk] 1 = attend, or NULL
    float *out;
    long long q_sb, q_ss, q_sh;  // element strides: batch, token, head (d is contiguous)
    long long k_sb, k_ss, k_sh;
    long long v_sb, v_ss, v_sh;
    long long o_sb, o_ss, o_sh;
    int B, H, Sq, Sk;
    float scale;
};

template <int D>
__global__ __launch_bounds__(256) void attention_fwd_kernel(AttnArgs p) {
    constexpr int KP = D + 4;   // K tile pitch: conflict-free ds_read_b128 over 16 keys
    constexpr int NG = D / 8;   // 8-wide k groups of the QK^T reduction
    constexpr int NT = D / 32;  // 32-row tiles of O^T
    __shared__ __attribute__((aligned(16))) float Ks[32 * KP];
    __shared__ __attribute__((aligned(16))) float Vs[32 * D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int query = blockIdx.x * 128 + wave * 32 + l31;
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

    for (int k0 = 0; k0 < p.Sk; k0 += 32) {
        __syncthreads();  // previous tile fully consumed
        // stage K and V tiles: 32 keys x D floats each, float4 per thread per pass
        for (int i = tid; i < 32 * (D / 4); i += 256) {
            const int key = i / (D / 4), c4 = (i - key * (D / 4)) * 4;
            float4 kv = make_float4(0, 0, 0, 0), vv = make_float4(0, 0, 0, 0);
            if (k0 + key < p.Sk) {
                kv = *reinterpret_cast<const float4 *>(kbase + (long long)(k0 + key) * p.k_ss + c4);
                vv = *reinterpret_cast<const float4 *>(vbase + (long long)(k0 + key) * p.v_ss + c4);
            }
            *reinterpret_cast<float4 *>(&Ks[key * KP + c4]) = kv;
            *reinterpret_cast<float4 *>(&Vs[key * D + c4]) = vv;
        }
        __syncthreads();

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
    if (!qok) return;
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

}  // namespace cer

using namespace cer;

extern "C" int cer_attention_fwd(const float *q, const float *k, const float *v, const int *key_mask, float *out,
                                 int B, int H, int Sq, int Sk, int D, const long long *q_strides,
                                 const long long *k_strides, const long long *v_strides, const long long *o_strides,
                                 float scale, void *stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || !q_strides || !k_strides || !v_strides ||
        !o_strides)
        return cer_set_error(CER_ERR_INVALID_ARG, "attention_fwd: bad argument");
    for (int i = 0; i < 3; ++i)
        if ((q_strides[i] & 3) || (k_strides[i] & 3) || (v_strides[i] & 3) || (o_strides[i] & 3))
            return cer_set_error(CER_ERR_INVALID_ARG, "attention_fwd: strides must be multiples of 4 floats");
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.key_mask = key_mask; a.out = out;
    a.q_sb = q_strides[0]; a.q_ss = q_strides[1]; a.q_sh = q_strides[2];
    a.k_sb = k_strides[0]; a.k_ss = k_strides[1]; a.k_sh = k_strides[2];
    a.v_sb = v_strides[0]; a.v_ss = v_strides[1]; a.v_sh = v_strides[2];
    a.o_sb = o_strides[0]; a.o_ss = o_strides[1]; a.o_sh = o_strides[2];
    a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale;
    dim3 grid((Sq + 127) / 128, H, B);
    if (D == 64) CER_LAUNCH(attention_fwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (D == 128) CER_LAUNCH(attention_fwd_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (D == 32) CER_LAUNCH(attention_fwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else return cer_set_error(CER_ERR_UNSUPPORTED, "attention_fwd: head dim must be 32, 64 or 128");
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
