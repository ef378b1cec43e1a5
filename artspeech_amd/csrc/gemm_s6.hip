// General "NT" GEMM in the split matrix arithmetic (as_set_matrix_arith(1)):  C[g][M][N] = act(A[g][M][K] . B[g][N][K]^T + bias)
// with BOTH operands fp32 in memory, reduction-contiguous (an nn.Linear forward: encoder_decoder/models.py:111-116 -- the GRU
// input projection, the trunk Linear -- and, with a transposed copy of the weights, its input gradient).  Each fp32 operand
// element is split exactly into three bfloat16 numbers and the product is rebuilt from six plane products on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (lin_f32.hip has the arithmetic's rationale and its error measurements).
//
// Shape of the kernel -- what the measurements of lin_s6_kernel asked for:
//   * that kernel's loop is bound by the bytes its CUs pull from L2 (a wave streams the weight planes of its own 32 columns:
//     9.3 B per SIMD-cycle of matrix work at 64 x 256 tiles): here BOTH operands go through LDS, staged once per workgroup,
//     on square 128 x 128 tiles -- 5.3 B per SIMD-cycle, and no pre-split copy of the weights (no plane-emit launch);
//   * 4 waves (2 x 2), a wave owns 64 x 64 = four accumulators: a fragment read from LDS feeds two matrix instructions per
//     plane pair (12 ds_read_b128 per 24 MFMAs and 16-deep k-step);
//   * every thread loads 4 + 4 consecutive k of two A rows and two B rows per 16-deep k-tile (global_load_dwordx4, two tiles
//     ahead in two register sets), splits them (4.5 vector instructions per element, each element once per workgroup) and
//     writes 3 x 8 bytes per load into the tile's plane images ([plane][128 rows][16 k] bf16 = 32-byte rows, the two 16-byte
//     halves XOR-swizzled by (row >> 3) & 1: the 16 lanes one LDS cycle of a ds_read_b128 serves then cover all 64 banks);
//   * 48 KB of LDS (two tiles) and <= 168 VGPRs: three workgroups per CU, each other's split / barrier / epilogue phases under
//     each other's matrix work; one barrier per k-tile.
#include "gemm_internal.h"
#include "split_arith.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TB = 128, BK = 16, NTH = 256;
constexpr int PLANE = TB * BK * 2;        // bytes of one plane of one operand tile: 128 rows x 16 bf16
constexpr int OPER = 3 * PLANE;           // one operand's three planes
constexpr int BUF = 2 * OPER;             // A then B

struct S6K {
    const float* A; long lda, a_batch;
    const float* B; long ldb, b_batch;
    float* C; long ldc, c_batch;
    const float* bias; long bias_batch;
    int M, N, K, act, tiles_m, tiles_n;
    // grouped batches (as_gemm.a_off ...: element offsets per batch member instead of the linear strides) and the bit image of
    // a ReLU epilogue (as_gemm.relu_bits: [batch][M][ceil(N / 32)] words, bit n % 32 of word n / 32 = result > 0)
    const long* a_off; const long* b_off; const long* c_off; const long* bias_off;
    unsigned* relu_bits; long relu_bits_batch; int ncb;
    // the general kernel's extended operands (gemm_f32.hip; input-gradient shapes): initial value of the accumulators, the bit
    // image of a ReLU backward (elements whose bit is clear are stored as 0), segmented reduction
    const float* res; long res_ld, res_batch; const long* res_off;
    const unsigned* mask_bits; long mask_batch;
    int k_seg, nseg; const long* a_seg_off; const long* b_seg_off;
};

typedef const __attribute__((address_space(1))) char* gptr;
typedef const __attribute__((address_space(1))) f32x4* gptr_f4;
__device__ __forceinline__ gptr uniform_ptr(const void* p) {   // see lin_f32.hip
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<gptr>(((uintptr_t)hi << 32) | lo);
}

// BNC: the B operand is column-contiguous in memory, B[k][n] (an input gradient dx = dz . W with W as the forward stores it):
// a thread then loads 8 consecutive k of ONE column (8 dword loads, each coalesced over the 64 columns of its wave) and writes
// 16 bytes per plane.  EXT: res / mask_bits / k_seg.
template <bool BNC, bool EXT>
__global__ __launch_bounds__(NTH, 3) void gemm_s6_kernel(S6K g) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * BUF];
    // block -> (batch, n-tile, m-tile): the m-tiles of one (batch, n-tile) are consecutive (they share the B panel in L2)
    int t = blockIdx.x;
    const int tm = t % g.tiles_m;
    t /= g.tiles_m;
    const int tn = t % g.tiles_n, bz = t / g.tiles_n;
    const int m0 = tm * TB, n0 = tn * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // loads: thread -> rows (tid >> 2) and (tid >> 2) + 64 of each operand tile, 16-byte chunk tid & 3 of the row's 16 k
    const int lrow = tid >> 2, lch = tid & 3;
    const bool seg = EXT && g.k_seg > 0;
    const long* a_seg = seg ? g.a_seg_off + (long)bz * g.nseg : nullptr;
    const long* b_seg = seg ? g.b_seg_off + (long)bz * g.nseg : nullptr;
    gptr Au = uniform_ptr(g.A + (seg ? a_seg[0] : g.a_off ? g.a_off[bz] : (long)bz * g.a_batch));
    gptr Bu = uniform_ptr(g.B + (seg ? b_seg[0] : g.b_off ? g.b_off[bz] : (long)bz * g.b_batch));
    // the NEXT segment's bases are fetched a segment ahead (scalar loads that have 16 k-tiles to arrive)
    long a_next = seg ? a_seg[min(1, g.nseg - 1)] : 0, b_next = seg ? b_seg[min(1, g.nseg - 1)] : 0;
    unsigned a_off[2], b_off[2];
    int wr[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = lrow + 64 * q;
        a_off[q] = (unsigned)((long)min(m0 + r, g.M - 1) * g.lda + lch * 4) * 4u;    // bytes (operands < 4 GB: host check)
        b_off[q] = (unsigned)((long)min(n0 + r, g.N - 1) * g.ldb + lch * 4) * 4u;
        wr[q] = r * 32 + (((lch >> 1) ^ ((r >> 3) & 1)) * 16) + (lch & 1) * 8;
    }
    // BNC: column n0 + (tid & 127), k half tid >> 7 (8 consecutive k); byte step between two k = 4 * ldb
    const int bn = tid & 127, bkh = tid >> 7;
    const unsigned ldb4 = (unsigned)g.ldb * 4u;
    const unsigned bnc_off = (unsigned)min(n0 + bn, g.N - 1) * 4u + (unsigned)(bkh * 8) * ldb4;
    const int bnc_wr = bn * 32 + ((bkh ^ ((bn >> 3) & 1)) * 16);
    const int nk = g.K / BK;
    const int kspan = seg ? g.k_seg : g.K;     // k range addressed from the current bases
    int ld_t = 0, ld_k = 0, ld_seg = 0;        // the next load: k-tile, k inside the segment, segment (all wave-uniform)
    struct Regs { f32x4 a[2], b[2]; };
    auto load = [&](Regs& x) {
        const unsigned ko = (unsigned)ld_k * 4u;
#pragma unroll
        for (int q = 0; q < 2; ++q) x.a[q] = *reinterpret_cast<gptr_f4>(Au + (a_off[q] + ko));
        if constexpr (!BNC) {
#pragma unroll
            for (int q = 0; q < 2; ++q) x.b[q] = *reinterpret_cast<gptr_f4>(Bu + (b_off[q] + ko));
        } else {
            const unsigned kb = bnc_off + (unsigned)ld_k * ldb4;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    x.b[q][e] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(Bu + (kb + (unsigned)(q * 4 + e) * ldb4));
        }
        // advance (behind the last tile the loads repeat it: unconditional loads keep hipcc's vmcnt counts exact)
        if constexpr (!EXT) {
            ld_t = min(ld_t + 1, nk - 1);
            ld_k = ld_t * BK;
        } else if (ld_t + 1 < nk) {
            ++ld_t;
            ld_k += BK;
            if (ld_k == kspan) {
                ld_k = 0;
                ++ld_seg;
                Au = uniform_ptr(g.A + a_next);
                Bu = uniform_ptr(g.B + b_next);
                const int nx = min(ld_seg + 1, g.nseg - 1);
                a_next = a_seg[nx];
                b_next = b_seg[nx];
            }
        }
    };
    auto store = [&](const Regs& x, int buf) {
        unsigned char* base = sm + buf * BUF;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 v = x.a[q];
            unsigned h0, m0_, l0, h1, m1, l1;
            split_pair(v.x, v.y, h0, m0_, l0);
            split_pair(v.z, v.w, h1, m1, l1);
            unsigned char* d = base + wr[q];
            *reinterpret_cast<u32x2*>(d) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(d + PLANE) = (u32x2){m0_, m1};
            *reinterpret_cast<u32x2*>(d + 2 * PLANE) = (u32x2){l0, l1};
        }
        if constexpr (!BNC) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 v = x.b[q];
                unsigned h0, m0_, l0, h1, m1, l1;
                split_pair(v.x, v.y, h0, m0_, l0);
                split_pair(v.z, v.w, h1, m1, l1);
                unsigned char* d = base + OPER + wr[q];
                *reinterpret_cast<u32x2*>(d) = (u32x2){h0, h1};
                *reinterpret_cast<u32x2*>(d + PLANE) = (u32x2){m0_, m1};
                *reinterpret_cast<u32x2*>(d + 2 * PLANE) = (u32x2){l0, l1};
            }
        } else {
            u32x4 h, m, l;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned h0, m0_, l0, h1, m1, l1;
                split_pair(x.b[q].x, x.b[q].y, h0, m0_, l0);
                split_pair(x.b[q].z, x.b[q].w, h1, m1, l1);
                h[2 * q] = h0; h[2 * q + 1] = h1;
                m[2 * q] = m0_; m[2 * q + 1] = m1;
                l[2 * q] = l0; l[2 * q + 1] = l1;
            }
            unsigned char* d = base + OPER + bnc_wr;
            *reinterpret_cast<u32x4*>(d) = h;
            *reinterpret_cast<u32x4*>(d + PLANE) = m;
            *reinterpret_cast<u32x4*>(d + 2 * PLANE) = l;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fsw = ((l31 >> 3) & 1) ^ lh;     // 16-byte half of this lane's 8 k inside its (swizzled) row
    const unsigned char* a_rd = sm + (wm * 64 + l31) * 32 + fsw * 16;
    const unsigned char* b_rd = sm + OPER + (wn * 64 + l31) * 32 + fsw * 16;

    Regs x[2];
    load(x[0]);
    load(x[1]);
    if (EXT && g.res) {   // the accumulators start from `res` (requested behind the first two tiles' loads)
        const float* R = g.res + (g.res_off ? g.res_off[bz] : (long)bz * g.res_batch);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < g.M && col < g.N) acc[i][j][r] = R[(long)row * g.res_ld + col];
                }
        }
    }
    store(x[0], 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // k-tile kt (U = kt & 1): images in buffer U; x[U ^ 1] holds tile kt + 1, x[U] is refilled with tile kt + 2
    auto tile = [&](auto Uc) {
        constexpr int U = decltype(Uc)::value;
        load(x[U]);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i][p] = *reinterpret_cast<const bf16x8*>(a_rd + U * BUF + p * PLANE + i * 32 * 32);
                fb[i][p] = *reinterpret_cast<const bf16x8*>(b_rd + U * BUF + p * PLANE + i * 32 * 32);
            }
        constexpr int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {0, 1, 2, 0, 1, 0};   // without mid.lo, lo.mid, lo.lo
#pragma unroll
        for (int o = 0; o < 6; ++o)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[o]], fb[j][PB[o]], acc[i][j], 0, 0, 0);
        store(x[U ^ 1], U ^ 1);   // (behind the last tile: a clamped repeat into the idle buffer)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        tile(IC2<0>{});
        tile(IC2<1>{});
    }
    if (kt < nk) tile(IC2<0>{});

    // ---- epilogue: D[i][j] block (i, j) of the wave: row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = n0 + wn * 64 + j * 32 + l31
    float* C = g.C + (g.c_off ? g.c_off[bz] : (long)bz * g.c_batch);
    const float* bias = g.bias ? g.bias + (g.bias_off ? g.bias_off[bz] : (long)bz * g.bias_batch) : nullptr;
    if (g.relu_bits == nullptr) {
        const unsigned* mask = (EXT && g.mask_bits) ? g.mask_bits + (long)bz * g.mask_batch : nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            if (col >= g.N) continue;
            const float bj = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < g.M) {
                        float v = acc[i][j][r] + bj;
                        if (g.act == 1) v = as_relu(v);
                        else if (g.act == 2) v = as_sigmoid(v);
                        if (EXT && mask && !((mask[(long)row * g.ncb + (col >> 5)] >> l31) & 1u)) v = 0.f;
                        C[(long)row * g.ldc + col] = v;
                    }
                }
        }
        return;
    }
    // ReLU epilogue that leaves its bit image (the backward's mask): a ballot per accumulator register holds the 32 columns of
    // two rows; the 32 row words of a block are gathered into one register (v_writelane_b32, lane = block row) and stored by
    // lanes 0..31.  (Two wait states between the compare and v_writelane_b32: hipcc does not look into inline assembly.)
    unsigned* bits = g.relu_bits + (long)bz * g.relu_bits_batch;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const bool cok = col < g.N;
        const float bj = (bias && cok) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned mine = 0;
            const int rb = m0 + wm * 64 + i * 32;
#define AS_S6_BITS(r)                                                                                                          \
            {                                                                                                                 \
                const float v = as_relu(acc[i][j][r] + bj);                                                                   \
                const unsigned long long b = __ballot(cok && v > 0.f);                                                        \
                asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(mine) : "s"((unsigned)b), "i"(((r) & 3) + 8 * ((r) >> 2))); \
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mine) : "s"((unsigned)(b >> 32)), "i"(((r) & 3) + 8 * ((r) >> 2) + 4)); \
                const int row = rb + ((r) & 3) + 8 * ((r) >> 2) + 4 * lh;                                                     \
                if (cok && row < g.M) C[(long)row * g.ldc + col] = v;                                                         \
            }
            AS_S6_BITS(0) AS_S6_BITS(1) AS_S6_BITS(2) AS_S6_BITS(3) AS_S6_BITS(4) AS_S6_BITS(5) AS_S6_BITS(6) AS_S6_BITS(7)
            AS_S6_BITS(8) AS_S6_BITS(9) AS_S6_BITS(10) AS_S6_BITS(11) AS_S6_BITS(12) AS_S6_BITS(13) AS_S6_BITS(14) AS_S6_BITS(15)
#undef AS_S6_BITS
            const int word = (n0 + wn * 64 + j * 32) >> 5;
            if (lane < 32 && rb + lane < g.M && word < g.ncb) bits[(long)(rb + lane) * g.ncb + word] = mine;
        }
    }
}

}  // namespace

// see gemm_internal.h
int as_gemm_s6_nt_ext(const as_gemm* g, hipStream_t st) {
    if (as_matrix_arith() != AS_ARITH_BF16X6) return 0;
    const bool bnc = g->b_k != 1;     // B[k][n], n contiguous (input-gradient orientation)
    if (g->a_k != 1 || (bnc && g->b_j != 1) || g->K < BK || g->K % BK || g->act < 0 || g->act > 2) return 0;
    if (g->k_tri || g->colsum || g->splitk_ws || g->accumulate || g->b_kT || g->b_kshift || (g->precision != 0 && g->precision != 3)) return 0;
    const bool ext = g->res || g->mask_bits || g->k_seg;
    if (g->relu_bits && (g->act != 1 || ext)) return 0;
    if (g->a_i % 4 || (reinterpret_cast<uintptr_t>(g->A) & 15) || (!g->a_off && !g->k_seg && g->a_batch % 4)) return 0;
    if (!bnc && (g->b_j % 4 || (reinterpret_cast<uintptr_t>(g->B) & 15) || (!g->b_off && !g->k_seg && g->b_batch % 4))) return 0;
    if (g->k_seg && (g->k_seg % BK || g->K % g->k_seg || !g->a_seg_off || !g->b_seg_off)) return 0;
    const long kspan = g->k_seg ? g->k_seg : g->K;
    // 32-bit byte offsets inside one batch member / segment
    if ((long)g->M * g->a_i >= (1L << 30) || (bnc ? kspan * g->b_k + g->N : (long)g->N * g->b_j) >= (1L << 30)) return 0;
    S6K k{};
    k.A = g->A; k.lda = g->a_i; k.a_batch = g->a_batch;
    k.B = g->B; k.ldb = bnc ? g->b_k : g->b_j; k.b_batch = g->b_batch;
    k.C = g->C; k.ldc = g->ldc; k.c_batch = g->c_batch;
    k.bias = g->bias; k.bias_batch = g->bias_batch;
    k.M = g->M; k.N = g->N; k.K = g->K; k.act = g->act; k.tiles_m = as_cdiv(g->M, TB); k.tiles_n = as_cdiv(g->N, TB);
    k.a_off = (const long*)g->a_off; k.b_off = (const long*)g->b_off; k.c_off = (const long*)g->c_off; k.bias_off = (const long*)g->bias_off;
    k.relu_bits = g->relu_bits; k.relu_bits_batch = g->relu_bits_batch; k.ncb = (g->N + 31) / 32;
    k.res = g->res; k.res_ld = g->res_ld; k.res_batch = g->res_batch; k.res_off = (const long*)g->res_off;
    k.mask_bits = g->mask_bits; k.mask_batch = g->mask_batch;
    k.k_seg = g->k_seg; k.nseg = g->k_seg ? g->K / g->k_seg : 1;
    k.a_seg_off = (const long*)g->a_seg_off; k.b_seg_off = (const long*)g->b_seg_off;
    const long blocks = (long)k.tiles_m * k.tiles_n * g->batch;
    if (blocks > (1L << 30)) return 0;
    const dim3 grid((unsigned)blocks), blk(NTH);
    if (ext) {
        if (bnc) hipLaunchKernelGGL((gemm_s6_kernel<true, true>), grid, blk, 0, st, k);
        else hipLaunchKernelGGL((gemm_s6_kernel<false, true>), grid, blk, 0, st, k);
    } else {
        if (bnc) hipLaunchKernelGGL((gemm_s6_kernel<true, false>), grid, blk, 0, st, k);
        else hipLaunchKernelGGL((gemm_s6_kernel<false, false>), grid, blk, 0, st, k);
    }
    AS_LAUNCH_CHECK("as_gemm_s6");
    return 1;
}

int as_gemm_s6_nt(const float* A, long lda, long a_batch, const float* B, long ldb, long b_batch, const float* bias, long bias_batch, float* C,
                  long ldc, long c_batch, int M, int N, int K, int batch, int act, hipStream_t st) {
    if (!A || !B || !C || M < 1 || N < 1 || batch < 1) return 0;
    as_gemm g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.a_i = lda; g.a_k = 1; g.b_j = ldb; g.b_k = 1; g.ldc = ldc;
    g.batch = batch; g.a_batch = a_batch; g.b_batch = b_batch; g.c_batch = c_batch; g.bias_batch = bias_batch; g.act = act;
    return as_gemm_s6_nt_ext(&g, st);
}
