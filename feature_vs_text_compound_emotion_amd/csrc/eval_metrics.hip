// eval_metrics.hip -- the evaluation path's aggregation on the device.
//
// The reference's Trainer.inference (trainer.py:436-523) copies every video's logits to the host and does the rest in
// numpy / sklearn: window stitching (trainer.py:832-892: scatter-add the window outputs, divide by the overlap
// counts), frame -> video aggregation (metrics.py:43-151: majority vote / mean logits / mean probabilities, optional
// drop of C-EXPR-DB's last class 'Other') and the confusion counts behind F1 / accuracy (metrics.py:148-193).  Here the
// logits never leave the GPU: three kernels leave one [C, C] frame-level and three [C, C] video-level confusion count
// matrices, which is all the scores need -- one small copy per evaluation instead of one per video.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cer_internal.h"

namespace cer {

// out[f][c] = (sum over windows w covering frame f, in window order, of win[w][f - start_w][c]) / (number of such windows)
// -- the reference adds window after window into a zero tensor and divides at the end; same order, same rounding.
__global__ void window_stitch_kernel(const float *__restrict__ win, const int *__restrict__ starts, int nw, int Lw, int C,
                                     int total, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)total * C) return;
    const int f = (int)(i / C), c = (int)(i - (size_t)f * C);
    float s = 0.f;
    int cnt = 0;
    for (int w = 0; w < nw; ++w) {
        const int r = f - starts[w];
        if (r >= 0 && r < Lw) {
            s += win[((size_t)w * Lw + r) * C + c];
            ++cnt;
        }
    }
    out[i] = cnt > 0 ? s / (float)cnt : 0.f;
}

__device__ __forceinline__ int argmax_first(const float *z, int n) {
    int b = 0;
    float m = z[0];
    for (int c = 1; c < n; ++c)
        if (z[c] > m) { m = z[c]; b = c; }   // first maximum, like numpy.argmax
    return b;
}

constexpr int MAXC = 16;

// Frame level: one thread per frame; cm[label][pred] += 1 (integer atomics: order-independent, deterministic).
// ignore_class >= 0: the last logit column is dropped and frames labelled ignore_class are skipped (metrics.py:62-84).
__global__ void frame_confusion_kernel(const float *__restrict__ logits, const float *__restrict__ labels, int R, int C,
                                       int ignore_class, unsigned long long *__restrict__ cm, int *__restrict__ bad) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int nc = ignore_class >= 0 ? C - 1 : C;
    const float lf = labels[r];
    const int l = (int)lf;
    if (!(lf >= 0.f && lf < (float)C)) { atomicAdd(bad, 1); return; }
    if (ignore_class >= 0 && l == ignore_class) return;
    const int p = argmax_first(logits + (size_t)r * C, nc);
    atomicAdd(&cm[(size_t)l * C + p], 1ull);
}

// Video level: one block per video (rows [off[v], off[v+1]) of the concatenated logits).  The three decisions of
// metrics.py:118-139: majority vote over the frame predictions (ties: the class that reaches the winning count and
// was seen FIRST, like collections.Counter.most_common), argmax of the mean logits, argmax of the mean of
// softmax(logits) (no max subtraction, like the reference).  A video whose frames carry different labels is an error in
// the reference (assert len(unique) == 1): counted in `bad`.
__global__ __launch_bounds__(256) void video_confusion_kernel(const float *__restrict__ logits, const float *__restrict__ labels,
                                                              const int *__restrict__ off, int C, int ignore_class,
                                                              unsigned long long *__restrict__ cm3, int *__restrict__ vpred,
                                                              int *__restrict__ bad) {
    __shared__ int votes[MAXC], first[MAXC], mixed;
    __shared__ float slog[MAXC], sprob[MAXC];
    __shared__ float red[256];
    const int v = blockIdx.x, r0 = off[v], r1 = off[v + 1], tid = threadIdx.x;
    const int nc = ignore_class >= 0 ? C - 1 : C;
    if (tid < MAXC) { votes[tid] = 0; first[tid] = 0x7fffffff; slog[tid] = 0.f; sprob[tid] = 0.f; }
    if (tid == 0) mixed = 0;
    __syncthreads();
    if (r1 <= r0) {
        if (tid == 0) atomicAdd(bad, 1);
        return;
    }
    const float lab0 = labels[r0];
    for (int r = r0 + tid; r < r1; r += blockDim.x) {
        if (labels[r] != lab0) mixed = 1;
        const int p = argmax_first(logits + (size_t)r * C, nc);
        atomicAdd(&votes[p], 1);
        atomicMin(&first[p], r);
    }
    // column sums of logits and of softmax(logits): deterministic tree per class (fixed partition of the rows)
    for (int c = 0; c < nc; ++c) {
        float a = 0.f, b = 0.f;
        for (int r = r0 + tid; r < r1; r += blockDim.x) {
            const float *z = logits + (size_t)r * C;
            float den = 0.f;
            for (int k = 0; k < nc; ++k) den += expf(z[k]);
            a += z[c];
            b += expf(z[c]) / den;
        }
        red[tid] = a;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) slog[c] = red[0];
        __syncthreads();
        red[tid] = b;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) sprob[c] = red[0];
        __syncthreads();
    }
    if (tid != 0) return;
    const int l = (int)lab0;
    if (mixed || !(lab0 >= 0.f && lab0 < (float)C)) { atomicAdd(bad, 1); return; }
    int pv = 0;
    for (int c = 1; c < nc; ++c)
        if (votes[c] > votes[pv] || (votes[c] == votes[pv] && first[c] < first[pv])) pv = c;
    const int pl = argmax_first(slog, nc), pp = argmax_first(sprob, nc);
    if (vpred) { vpred[3 * v] = pv; vpred[3 * v + 1] = pl; vpred[3 * v + 2] = pp; }
    if (ignore_class >= 0 && l == ignore_class) return;
    atomicAdd(&cm3[(size_t)(0 * C + l) * C + pv], 1ull);
    atomicAdd(&cm3[(size_t)(1 * C + l) * C + pl], 1ull);
    atomicAdd(&cm3[(size_t)(2 * C + l) * C + pp], 1ull);
}

}  // namespace cer

using namespace cer;

extern "C" int cer_window_stitch(const float *win_out, const int *starts, int nw, int Lw, int C, int total, float *out,
                                 void *stream) {
    if (!win_out || !starts || !out || nw <= 0 || Lw <= 0 || C <= 0 || total <= 0)
        return cer_set_error(CER_ERR_INVALID_ARG, "window_stitch: bad argument");
    const size_t n = (size_t)total * C;
    CER_LAUNCH(window_stitch_kernel, dim3(cer_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, win_out, starts, nw, Lw, C, total, out);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}

extern "C" int cer_eval_accumulate(const float *logits, const float *labels, const int *video_offsets, int V, int R, int C,
                                   int ignore_class, unsigned long long *frame_cm, unsigned long long *video_cm, int *video_pred,
                                   int *bad, void *stream) {
    if (!logits || !labels || !video_offsets || !frame_cm || !video_cm || !bad || V <= 0 || R <= 0 || C < 2 || C > MAXC)
        return cer_set_error(CER_ERR_INVALID_ARG, "eval_accumulate: bad argument (2 <= C <= 16)");
    if (ignore_class >= C) return cer_set_error(CER_ERR_INVALID_ARG, "eval_accumulate: ignore_class out of range");
    CER_LAUNCH(frame_confusion_kernel, dim3(cer_blocks((size_t)R, 256)), dim3(256), 0, (hipStream_t)stream, logits, labels, R, C,
               ignore_class, frame_cm, bad);
    CER_LAUNCH(video_confusion_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, logits, labels, video_offsets, C, ignore_class,
               video_cm, video_pred, bad);
    CER_HIP_CHECK(hipGetLastError());
    return CER_OK;
}
