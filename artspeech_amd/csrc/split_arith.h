// Device helpers of the split matrix arithmetic (as_set_matrix_arith(1)): an fp32 number as three bfloat16 numbers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// x = hi + mid + lo exactly: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest even (8 + 8 + 8
// significand bits).  Two fp32 numbers -> three words of packed bfloat16 pairs (element a in the low half): v_cvt_pk_bf16_f32
// + v_pk_add_f32, 4.5 vector instructions per element.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    v.x -= __uint_as_float(hi << 16);
    v.y -= __uint_as_float(hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    v.x -= __uint_as_float(mid << 16);
    v.y -= __uint_as_float(mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
