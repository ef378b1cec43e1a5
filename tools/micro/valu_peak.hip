// Issue cost of v_fma_f32 vs v_pk_fma_f32 on one CU-resident workgroup: 12 independent accumulation chains per lane,
// 1 or 2 waves per SIMD (256 / 512 threads).  Prints shader cycles per wave-instruction and FMAs per cycle per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, float a, float b) {
    unsigned long long t0 = 0, t1 = 0;
    float s = 0.f;
    if (PK) {
        f32x2 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = f32x2{threadIdx.x * 1e-3f + i, i * 0.5f};
        const f32x2 x = {a, b}, y = {b, a};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_elementwise_fma(acc[i], x, y);
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 12; ++i) s += acc[i].x + acc[i].y;
    } else {
        float acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = threadIdx.x * 1e-3f + i;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 12; ++i) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int threads : {256, 512})
        for (int pk = 0; pk < 2; ++pk) {
            for (int rep = 0; rep < 2; ++rep) {
                if (pk) hipLaunchKernelGGL(k<true>, dim3(64), dim3(threads), 0, 0, out, cyc, iters, 0.999f, 0.001f);
                else hipLaunchKernelGGL(k<false>, dim3(64), dim3(threads), 0, 0, out, cyc, iters, 0.999f, 0.001f);
                hipDeviceSynchronize();
            }
            unsigned long long h[64];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            const double c = (double)h[0], insts = 12.0 * iters;                 // per wave
            const int waves_per_simd = threads / 256;
            const double fma_per_inst = pk ? 128.0 : 64.0;
            printf("%-14s %d wave(s)/SIMD: %.2f cycles per wave-instruction, %.1f FMA/cycle/SIMD\n", pk ? "v_pk_fma_f32" : "v_fma_f32",
                   waves_per_simd, c / insts, waves_per_simd * insts * fma_per_inst / c);
        }
    return 0;
}
