// encoders.hip -- front ends of the audio and text encoders.
//
//  * log-mel: int16 PCM -> /32768 -> edge pad -> 400-sample periodic-Hann frames (hop 160) ->
//    |512-point DFT| -> 257x64 HTK mel matrix -> log(x + 0.01)
//    (reference abaw5_pre_processing/base/vggish/mel_features.py:75-114,207-236,
//     vggish_input.py:84-98).  The reference computes this in float64 numpy; the kernel keeps
//    float64 for the DFT and the mel product (MI355X has a full-rate fp64 vector pipe and the
//    whole front end is ~11 MFLOP per clip), and rounds to fp32 only at the output, exactly
//    where the reference casts for the VGGish input.
//  * example framing: gather 96-frame windows at host-computed start rows (the reference's
//    fractional hop uses Python's round-half-to-even, vggish_input.py:77-81 / my_frame).
//  * BERT embeddings: word + position + token-type gather fused with LayerNorm(eps 1e-12).
#include <math.h>

#include "cer_internal.h"

namespace cer {

constexpr int LM_WIN = 400, LM_HOP = 160, LM_FFT = 512, LM_BINS = 257, LM_MEL = 64;

// One block per STFT frame.  Direct DFT with an LDS twiddle table: 257 bins x 400 taps.
__global__ __launch_bounds__(256) void logmel_kernel(const int16_t *__restrict__ pcm, int num_samples,
                                                     int frames_per_clip, const double *__restrict__ mel,
                                                     float log_offset, float *__restrict__ out) {
    __shared__ double xs[LM_WIN];
    __shared__ double cs[LM_FFT], sn[LM_FFT];
    __shared__ double mag[LM_BINS + 3];
    const int clip = blockIdx.y, frame = blockIdx.x, tid = threadIdx.x;
    const int16_t *src = pcm + (size_t)clip * num_samples;
    for (int n = tid; n < LM_WIN; n += 256) {
        int idx = frame * LM_HOP + n;
        idx = idx < num_samples ? idx : num_samples - 1;  // np.pad(..., 'edge')
        const double w = 0.5 - 0.5 * cospi(2.0 * (double)n / (double)LM_WIN);
        xs[n] = ((double)src[idx] / 32768.0) * w;
    }
    for (int j = tid; j < LM_FFT; j += 256) {
        double s, c;
        sincospi(2.0 * (double)j / (double)LM_FFT, &s, &c);
        cs[j] = c;
        sn[j] = s;
    }
    __syncthreads();
    for (int k = tid; k < LM_BINS; k += 256) {
        double re = 0.0, im = 0.0;
        for (int n = 0; n < LM_WIN; ++n) {
            const int j = (k * n) & (LM_FFT - 1);
            re += xs[n] * cs[j];
            im -= xs[n] * sn[j];
        }
        mag[k] = sqrt(re * re + im * im);
    }
    __syncthreads();
    if (tid < LM_MEL) {
        double acc = 0.0;
        for (int k = 0; k < LM_BINS; ++k) acc += mag[k] * mel[k * LM_MEL + tid];
        out[((size_t)clip * frames_per_clip + frame) * LM_MEL + tid] = (float)log(acc + (double)log_offset);
    }
}

// examples[c][e][f][:] = logmel[c][starts[e] + f][:]
__global__ void frame_examples_kernel(const float4 *__restrict__ logmel, const int *__restrict__ starts,
                                      float4 *__restrict__ out, int clips, int frames_per_clip, int n_examples,
                                      int win) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)clips * n_examples * win * (LM_MEL / 4);
    if (idx >= total) return;
    const int c4 = (int)(idx % (LM_MEL / 4));
    size_t t = idx / (LM_MEL / 4);
    const int f = (int)(t % win); t /= win;
    const int e = (int)(t % n_examples);
    const int c = (int)(t / n_examples);
    out[idx] = logmel[((size_t)c * frames_per_clip + starts[e] + f) * (LM_MEL / 4) + c4];
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per token: e = word[id] + pos[p] + type[tt]; y = LN(e) * gamma + beta.
__global__ void bert_embed_ln_kernel(const long long *__restrict__ ids, const float *__restrict__ word,
                                     const float *__restrict__ pos, const float *__restrict__ type,
                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                     float *__restrict__ y, int tokens, int S, int Hd, int vocab, float eps) {
    const int tok = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tok >= tokens) return;
    long long id = ids[tok];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float *w = word + (size_t)id * Hd, *pp = pos + (size_t)(tok % S) * Hd;
    float s = 0.f;
    for (int c = lane; c < Hd; c += 64) s += w[c] + pp[c] + type[c];
    const float mu = wave_sum_f(s) / (float)Hd;
    s = 0.f;
    for (int c = lane; c < Hd; c += 64) { const float d = w[c] + pp[c] + type[c] - mu; s += d * d; }
    const float rstd = rsqrtf(wave_sum_f(s) / (float)Hd + eps);
    for (int c = lane; c < Hd; c += 64) y[(size_t)tok * Hd + c] = (w[c] + pp[c] + type[c] - mu) * rstd * gamma[c] + beta[c];
}

__global__ void add_inplace_kernel(float4 *__restrict__ y, const float4 *__restrict__ x, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 a = y[i], b = x[i];
    y[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

}  // namespace cer

using namespace cer;

extern "C" int cer_logmel_num_frames(int num_samples, int pad_samples) {
    const long long n = (long long)num_samples + pad_samples;
    return n < LM_WIN ? 0 : (int)(1 + (n - LM_WIN) / LM_HOP);
}

extern "C" int cer_logmel_fwd(const int16_t *pcm, int clips, int num_samples, int pad_samples, const double *mel_matrix,
                              float log_offset, float *logmel, void *stream) {
    const int frames = cer_logmel_num_frames(num_samples, pad_samples);
    if (!pcm || !mel_matrix || !logmel || clips <= 0 || num_samples <= 0 || pad_samples < 0 || frames <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "logmel_fwd: bad argument (need at least 400 samples incl. padding)");
    CER_LAUNCH(logmel_kernel, dim3(frames, clips), dim3(256), 0, (hipStream_t)stream, pcm, num_samples, frames, mel_matrix,
               log_offset, logmel);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_frame_examples(const float *logmel, const int *starts, float *examples, int clips, int frames_per_clip,
                                  int n_examples, int win, void *stream) {
    if (!logmel || !starts || !examples || clips <= 0 || frames_per_clip <= 0 || n_examples <= 0 || win <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "frame_examples: bad argument");
    const size_t total = (size_t)clips * n_examples * win * (LM_MEL / 4);
    CER_LAUNCH(frame_examples_kernel, dim3(cer_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)logmel,
               starts, (float4 *)examples, clips, frames_per_clip, n_examples, win);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_bert_embed_ln(const long long *ids, const float *word, const float *pos, const float *type,
                                 const float *gamma, const float *beta, float *y, int B, int S, int Hd, int vocab,
                                 int max_pos, float eps, void *stream) {
    if (!ids || !word || !pos || !type || !gamma || !beta || !y || B <= 0 || S <= 0 || Hd <= 0 || vocab <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "bert_embed_ln: bad argument");
    if (S > max_pos) return cer_set_error(CER_ERR_INVALID_ARG, "bert_embed_ln: sequence longer than the position table");
    const int tokens = B * S;
    CER_LAUNCH(bert_embed_ln_kernel, dim3((tokens + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, word, pos, type, gamma,
               beta, y, tokens, S, Hd, vocab, eps);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_add_inplace(float *y, const float *x, size_t n, void *stream) {
    if (!y || !x || n == 0 || (n & 3)) return cer_set_error(CER_ERR_INVALID_ARG, "add_inplace: n must be a positive multiple of 4");
    CER_LAUNCH(add_inplace_kernel, dim3(cer_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (float4 *)y,
               (const float4 *)x, n / 4);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
