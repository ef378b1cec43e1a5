// Internal (non-ABI) hand-over between the GEMM translation units.
#pragma once
#include "as_common.h"

// wgrad_f32.hip: weight-gradient shapes (both operands reduction-strided, long reduction).  1 = taken and launched,
// 0 = not a shape for this kernel (the caller continues with the general kernel), < 0 = error.
int as_wgrad_try(const as_gemm* g, hipStream_t st);
