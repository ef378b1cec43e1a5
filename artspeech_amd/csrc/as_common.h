// Shared helpers for the gfx950 kernels of libartspeech_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "artspeech_hip.h"

#define AS_WAVE 64

// Diagnostic switches (ablations, tuning aids, legacy kernels) exist only in the -DAS_DIAG build of the library
// (artspeech_amd/build.py --diag -> libartspeech_hip_diag.so, used by tools/).  In the product build every switch is a
// compile-time constant: no environment variable can change what the shipped library computes or which kernel it takes.
#ifdef AS_DIAG
#include <stdlib.h>
#define AS_DIAG_SET(name) (getenv(name) != nullptr)
#define AS_DIAG_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define AS_DIAG_STR(name) (getenv(name))
#else
#define AS_DIAG_SET(name) false
#define AS_DIAG_INT(name, dflt) (dflt)
#define AS_DIAG_STR(name) ((const char*)nullptr)
#endif

void as_set_error(const char* fmt, ...);

#define AS_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            as_set_error(__VA_ARGS__);         \
            return (code);                     \
        }                                      \
    } while (0)

// launch check: kernels are enqueued, never synchronised
#define AS_LAUNCH_CHECK(name)                                                  \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            as_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                   \
        }                                                                      \
    } while (0)

#define AS_TRY(expr)            \
    do {                        \
        int r__ = (expr);       \
        if (r__ != 0) return r__; \
    } while (0)

// optional per-phase timing (prof.hip); a no-op unless as_profile_enable(1)
struct AsProfScope {
    AsProfScope(const char* name, hipStream_t st);
    ~AsProfScope();
    const char* name_; hipStream_t st_; void* a_; void* b_;
};
bool as_profile_active();   // as_profile_enable(1) is in force
#define AS_PROF_CAT2(a, b) a##b
#define AS_PROF_CAT(a, b) AS_PROF_CAT2(a, b)
#define AS_PROF(name, st) AsProfScope AS_PROF_CAT(as_prof_scope_, __LINE__)(name, st)
// one named, timed launch sequence
#define AS_STEP(name, st, expr) \
    do {                        \
        AS_PROF(name, st);      \
        AS_TRY(expr);           \
    } while (0)

static inline int64_t as_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int as_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float as_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double as_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// ReLU that keeps a NaN a NaN, like torch.relu (fmaxf(NaN, 0) = 0 would hide an invalid activation: the transformer's test
// loop finds utterances with NaN predictions by exactly that propagation, transformer/evaluation.py:69-86)
__device__ __forceinline__ float as_relu(float v) { return v < 0.f ? 0.f : v; }
// v_exp_f32 + v_rcp_f32 (1 ulp each): |abs err| ~ 1e-7, far inside the 1e-4 parity budget, and 3x fewer
// instructions than an IEEE division on the recurrence's critical path
__device__ __forceinline__ float as_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh via one exp; no overflow (exp(+inf) -> inf -> rcp = 0 -> 1)
// (exp(2x) as ONE multiply by 2 log2(e) in front of v_exp_f32)
__device__ __forceinline__ float as_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.8853900817779268f));
}
