// cer_internal.h -- shared helpers for the libcer_hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cer_hip.h"

// Records a thread-local message and returns `code` (always negative).
int cer_set_error(int code, const char *fmt, ...);

#define CER_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return cer_set_error(CER_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,    \
                                 hipGetErrorString(_e));                                      \
    } while (0)

static inline unsigned cer_blocks(size_t n, unsigned per_block) {
    return (unsigned)((n + per_block - 1) / per_block);
}

// Launch with the thread's sticky HIP error cleared first, so that the hipGetLastError()
// check after the launch reports THIS launch and not an unrelated earlier failure (e.g. a
// device probe made before the process had initialised its GPU).
#define CER_LAUNCH(...)                 \
    do {                                \
        (void)hipGetLastError();        \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)
