// Probe (not part of the library): an fp32 GEMM whose operands arrive as three bfloat16 planes each (x = hi + mid + lo, an
// exact split of the 24-bit significand) and whose products run on the bf16 matrix instruction of gfx950
// (v_mfma_f32_32x32x16_bf16, 16 x the rate of v_mfma_f32_32x32x2_f32) with fp32 accumulation:
//
//     a . b = sum over plane pairs (p, q) of a_p . b_q           9 pairs: every bit of both operands
//                                                               6 pairs: without lo.lo, lo.mid, mid.lo (each <= 2^-24 |a||b|)
//
// Shape: C[z][M][256] = A[z][M][K] . B[z][256][K]^T (a Linear layer of the ArticulatorPredictor heads, models.py:10-33), the
// planes stored k-tile-major -- [plane][z][K / 16][rows][16] -- so that the 32 bytes a row contributes to a k-tile are one
// sector and a tile's rows are contiguous.  One workgroup = 128 rows x 256 columns, 8 waves (2 x 4) of 64 x 64, a 4-slot
// LDS ring of 16-deep k-tiles (36 KB each: 3 planes x 384 rows x 32 B) filled by LDS-DMA with three tiles in flight; a
// k-tile's fragments are read into registers under the MFMAs of the k-tile before it.
// Built and timed by tools/bench_split_gemm.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int BM = 128, BN = 256, BK = 16, NT = 512, NST = 4;
constexpr int ROWS = BM + BN;
constexpr int PLANE_B = ROWS * BK * 2;   // bytes of one plane of one k-tile
constexpr int STAGE_B = 3 * PLANE_B;
constexpr int PER_WAVE = 5;              // LDS-DMA instructions per wave and k-tile

struct SplitK {
    const uint16_t* A; const uint16_t* B; float* C;
    int M, K, Z, tiles_m;
    long a_plane, b_plane;               // elements between planes
};

__device__ __forceinline__ void glds16(const void* src, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

template <int NPROD>
__global__ __launch_bounds__(NT, 1) void split_gemm_kernel(SplitK g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int z = blockIdx.x / g.tiles_m, m0 = (blockIdx.x - z * g.tiles_m) * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = g.K / BK;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem);

    // ---- DMA plan: a piece = 32 rows x 32 B = 1 KiB; lane -> row lane >> 1, LDS chunk lane & 1, which receives the global
    // chunk (lane & 1) ^ ((row >> 2) & 1) (so that eight lanes of a ds_read_b128 touch all 32 banks)
    const uint16_t* src[PER_WAVE];
    unsigned dst[PER_WAVE];
    long stride[PER_WAVE];
    {
        const int r = lane >> 1, ch = (lane & 1) ^ ((r >> 2) & 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {       // A: 12 pieces in 16 slots (slots 12-15 repeat pieces 0-3)
            int q = wave * 2 + j;
            if (q >= 12) q -= 12;
            const int p = q >> 2, sub = q & 3;
            const long row = min(m0 + sub * 32 + r, g.M - 1);
            src[j] = g.A + p * g.a_plane + ((long)z * nk * g.M + row) * BK + ch * 8;
            dst[j] = p * PLANE_B + sub * 1024;
            stride[j] = (long)g.M * BK;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {       // B: 24 pieces
            const int q = wave * 3 + j, p = q >> 3, sub = q & 7;
            src[2 + j] = g.B + p * g.b_plane + ((long)z * nk * BN + sub * 32 + r) * BK + ch * 8;
            dst[2 + j] = p * PLANE_B + BM * BK * 2 + sub * 1024;
            stride[2 + j] = (long)BN * BK;
        }
    }
    auto issue = [&](int kt) {
        const unsigned base = smem_base + (unsigned)((kt % NST) * STAGE_B);
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) glds16(src[j] + kt * stride[j], base + dst[j]);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ring: at the top of iteration kt the fragments of k-tile kt are in registers (read during iteration kt - 1, under its
    // MFMAs), k-tile kt + 1 is awaited, kt + 2 and kt + 3 are in flight and slot kt % NST is refilled with k-tile kt + 4
#pragma unroll
    for (int s = 0; s < NST; ++s)
        if (s < nk) issue(s);
    const int swz = (l31 >> 2) & 1;
    const int a_off = (wm * 64 + l31) * 32 + ((lh ^ swz) * 16);
    const int b_off = (BM + wn * 64 + l31) * 32 + ((lh ^ swz) * 16);
    struct Frag { bf16x8 a[3][2], b[3][2]; };   // [plane: 0 hi, 1 mid, 2 lo][32-row / 32-column block]
    auto read = [&](Frag& f, int kt) {
        const unsigned char* st = smem + (kt % NST) * STAGE_B;
#pragma unroll
        for (int p = 2; p >= 0; --p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f.a[p][i] = *reinterpret_cast<const bf16x8*>(st + p * PLANE_B + a_off + i * 1024);
                f.b[p][i] = *reinterpret_cast<const bf16x8*>(st + p * PLANE_B + b_off + i * 1024);
            }
    };
    auto landed = [&](int kt, int issued) {   // this wave's pieces of k-tile kt have landed (k-tiles up to `issued` are issued)
        const int younger = min(nk - 1, issued) - kt;
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto mma = [&](const Frag& f) {
        constexpr int order[9][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}, {2, 0}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};   // small products first
#pragma unroll
        for (int o = 9 - NPROD; o < 9; ++o)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[order[o][0]][i], f.b[order[o][1]][j], acc[i][j], 0, 0, 0);
    };
    auto step = [&](Frag& cur, Frag& nxt, int kt) {
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): k-tile kt is in `cur` (the builtin, so that hipcc's own waitcnt pass sees it)
        if (kt + 1 < nk) landed(kt + 1, kt + NST - 1);
        __builtin_amdgcn_s_barrier();                        // ... for every wave: slot kt % NST is free, k-tile kt + 1 is complete
        if (kt + NST < nk) issue(kt + NST);
        read(nxt, kt + 1);   // (behind the last k-tile: a slot's stale bytes, never used)
        __builtin_amdgcn_sched_barrier(0);
        mma(cur);
        __builtin_amdgcn_sched_barrier(0);
    };
    Frag f0, f1;
    landed(0, NST - 1);
    __builtin_amdgcn_s_barrier();
    read(f0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        step(f0, f1, kt);
        if (kt + 1 < nk) step(f1, f0, kt + 1);
    }

    float* c0 = g.C + (long)z * g.M * BN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < g.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j) c0[(long)row * BN + wn * 64 + j * 32 + l31] = acc[i][j][r];
            }
        }
}

template <int NPROD>
int launch(const SplitK& g, hipStream_t s) {
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<NPROD>), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_B) != hipSuccess) return 2;
        once = true;
    }
    hipLaunchKernelGGL(split_gemm_kernel<NPROD>, dim3(g.Z * g.tiles_m), dim3(NT), NST * STAGE_B, s, g);
    return (int)hipGetLastError();
}
}  // namespace

// A: [3][Z][K/16][M][16] bf16, B: [3][Z][K/16][256][16] bf16, C: [Z][M][256] fp32
extern "C" int split_gemm_probe(const void* A, const void* B, float* C, int M, int K, int Z, int nprod, void* stream) {
    if (K % BK != 0 || M < 1 || Z < 1) return 1;
    SplitK g;
    g.A = (const uint16_t*)A; g.B = (const uint16_t*)B; g.C = C;
    g.M = M; g.K = K; g.Z = Z; g.tiles_m = (M + BM - 1) / BM;
    g.a_plane = (long)Z * M * K; g.b_plane = (long)Z * BN * K;
    hipStream_t s = (hipStream_t)stream;
    if (nprod == 9) return launch<9>(g, s);
    if (nprod == 6) return launch<6>(g, s);
    if (nprod == 3) return launch<3>(g, s);
    return 1;
}
