// conv_b3.h -- types shared by the bf16x3 (split hi/lo operand) conv kernels.
#pragma once
#include "conv_common.h"

namespace cer {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // 16-byte staging unit (first-class vector:
                                                                 // HIP's uint4 struct arrays ended up in scratch)
typedef __attribute__((address_space(3))) void *lds_ptr_t;

__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }


typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swz16(int q) { return (0x78 >> (2 * q)) & 3; }  // F = {0, 2, 3, 1}

int conv_b3_patch_launch(int tile, const ConvArgs &a, hipStream_t st);  // conv_b3_patch.hip
bool conv_b3_patch_ok(const ConvArgs &a);
bool conv_b3_win_ok(const ConvArgs &a);
int conv_b3_s2d_launch(int tile, const ConvArgs &a, hipStream_t st);    // conv_b3_s2d.hip
bool conv_b3_s2d_ok(const ConvArgs &a);

}  // namespace cer
