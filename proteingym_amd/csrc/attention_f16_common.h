// Shared by attention_f16.hip (4-wave kernels, prep passes) and attention_f16_pp.hip (the two-role 8-wave kernel).
#pragma once
#include "common.h"

namespace pgmi {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int AKT = 32;                       // keys per tile
constexpr float kAttDefer = 4.0f;             // lag allowed before O is rescaled (base-2 units; see the rescale in the kernel): +3.5 % (T = 288) / +4.5 % (T = 1024) over 0
constexpr int K_CH = AKT * 8;                 // chunks per K plane
constexpr int V_CH = 64 * 4;                  // chunks per V^T plane
constexpr int A_STAGE = 2 * K_CH + 2 * V_CH;  // chunks per buffer (hi+lo planes of K and V^T) = 16 KB

__device__ __forceinline__ unsigned int pack_h2(_Float16 a, _Float16 b) {
    const h2 t = {a, b};
    return __builtin_bit_cast(unsigned int, t);
}
__device__ __forceinline__ f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}


}  // namespace pgmi
