// MFMA wrappers (gfx950).  16x16xK, one wave: lane l supplies row/col (l & 15) and the KL consecutive k values
// starting at (l >> 4) * KL of each operand; D[row = (l>>4)*4 + r][col = l & 15] lands in accumulator element r.
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int K = 32, KL = 8;
    using Frag = short8;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <> struct Mma<f16_t> {
    static constexpr int K = 32, KL = 8;
    using Frag = short8;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int K = 4, KL = 1;
    using Frag = float;
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};
