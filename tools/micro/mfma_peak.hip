// Practical fp32-MFMA peak of the box: every SIMD issues v_mfma_f32_32x32x2_f32 back to back (4 independent accumulators
// per wave, WPS waves per SIMD), operands in registers, random data.  Prints TFLOP/s and the in-kernel clock
// (s_memtime ticks / s_memrealtime at 100 MHz).   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* clk, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f + blockIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    const int blocks = 256, iters = 20000;
    float* out; unsigned long long* clk;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = 2.0 * 32 * 32 * 2 * 4.0 * iters * 8 * blocks;
        printf("rep %d: %.3f ms  %.1f TFLOP/s   in-kernel clock %.0f MHz (block 0), %.0f MHz (block 128)\n", rep, ms, flops / ms / 1e9,
               100.0 * h[0] / h[1], 100.0 * h[256] / h[257]);
    }
    return 0;
}
