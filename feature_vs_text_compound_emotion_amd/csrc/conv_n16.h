// conv_n16.h -- types and the MFMA wrapper shared by the narrow (single 16-bit plane) conv kernels.
#pragma once
#include "conv_common.h"

namespace cer {

typedef __bf16 n_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 n_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int n_u32x4 __attribute__((ext_vector_type(4)));
typedef float n_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *n_lds_ptr_t;

template <bool F16>
__device__ __forceinline__ n_f32x4 mfma_n16(const n_u32x4 a, const n_u32x4 b, const n_f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(n_f16x8, a), __builtin_bit_cast(n_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(n_bf16x8, a), __builtin_bit_cast(n_bf16x8, b), c, 0, 0, 0);
}

// 16x16-patch kernels: slot = chunk ^ PATCH_F[window column], 3 bits per column (tools/check_swizzle.py: patch_table_constant())
constexpr unsigned long long PATCH_F_TABLE = 0xd92dad912240ull;
__device__ __forceinline__ int patch_f(int wx) { return (int)((PATCH_F_TABLE >> (3 * wx)) & 7ull); }
// block -> (patch, cout tile) -> (image, patch row, patch column) through reciprocals of the launch constants (conv_common.h)
struct PatchGeo { FastDiv tiles_n, pxn, pyn; };

int conv_n16_patch_launch(int tile, const ConvArgs &a, hipStream_t st);  // conv_n16_patch.hip
int conv_n16_p64_launch(const ConvArgs &a, hipStream_t st);              // conv_n16_p64.hip: the persistent Cin == 64 patch kernel (tile 79)
bool conv_n16_p64_ok(const ConvArgs &a);
bool conv_n16_patch_ok(const ConvArgs &a, int tile);         // can the patch kernel `tile` take this conv?
bool conv_n16_win_ok(const ConvArgs &a);                     // can the 1-D window kernels (tiles 73 / 74)?
int conv_n16_s2d_launch(int tile, const ConvArgs &a, hipStream_t st);    // conv_n16_s2d.hip
bool conv_n16_s2d_ok(const ConvArgs &a);

}  // namespace cer
