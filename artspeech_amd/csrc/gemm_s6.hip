// General GEMM in the split matrix arithmetic (as_set_matrix_arith(1); as_gemm.precision = 3):  C[g] = epilogue(A[g] . B[g]) with
// BOTH operands fp32 in memory, each element split exactly into three bfloat16 numbers and the product rebuilt from six plane
// products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (lin_f32.hip has the arithmetic's rationale and error measurements).
// Three operand orientations, one kernel:
//   forward          C[M][N] = act(A[M][K] . B[N][K]^T + bias)     (an nn.Linear: encoder_decoder/models.py:111-116,
//                    transformer/models.py:47-100), optional ReLU bit image;
//   input gradient   C[M][N] = keep ? res + A[M][K] . B[K][N] : 0  (dx = dz . W with W as the forward stores it; the residual
//                    gradient as the accumulators' initial value, the ReLU backward from the bit image, the reduction optionally
//                    in segments with their own operand bases: as_gemm.res / mask_bits / k_seg);
//   weight gradient  C[M][N] = A[K][M]^T . B[K][N]                 (dW = dz^T . x, bias gradient = column sums of A fused).
//
// Shape of the kernel -- what the measurements of lin_s6_kernel asked for:
//   * that kernel's loop is bound by the bytes its CUs pull from L2 (a wave streams the weight planes of its own 32 columns:
//     9.3 B per SIMD-cycle of matrix work at 64 x 256 tiles): here BOTH operands go through LDS, staged once per workgroup,
//     on square 128 x 128 tiles -- 5.3 B per SIMD-cycle, and no pre-split copy of the weights (no plane-emit launch);
//   * 4 waves (2 x 2), a wave owns 64 x 64 = four accumulators: a fragment read from LDS feeds two matrix instructions per
//     plane pair;
//   * every thread loads two float4s per operand and 16-deep k-tile (two tiles ahead in two register sets), splits them (4.5
//     vector instructions per element, each element once per workgroup) and writes 3 x 8 bytes per load into the tile's plane
//     images.  A reduction-contiguous operand gives a [128 rows][16 k] image (32-byte rows, the two 16-byte halves XOR-swizzled
//     by (row >> 3) & 1) read with ds_read_b128; a row-contiguous one (X[k][row]) a [16 k][128 rows] image (256-byte rows,
//     16-byte chunks XOR-swizzled) from which gfx950's transposing read ds_read_b64_tr_b16 delivers the same fragment (8
//     consecutive k of the lane's row) in two reads -- the loads stay float4s along the contiguous dimension;
//   * 48 KB of LDS (two tiles) and <= 168 VGPRs: three workgroups per CU, each other's split / barrier / epilogue phases under
//     each other's matrix work; one barrier per k-tile;
//   * XCD-aware tile order: workgroup w runs on XCD w % 8 and each XCD has its own L2, so the tiles that read the same
//     activation panel (the n-tiles of 128 rows of A; all tiles of a weight-gradient batch member) are numbered onto one XCD,
//     next to each other -- the panel is fetched from HBM once (PMC: 2.9 -> 1.5 GB for 110 x [256 x 256 x 6400]).
//   Measured: 51200 x 256 x 256 forward 82.9 (fp32 kernel) -> 56.0 us, 51200 x 1024 x 1024 877 -> 565 us (190 TF/s-equivalent);
//   the weight-gradient orientation is bound by its operand loads (ablation: 1045 -> 614 us without them at three workgroups
//   per CU), 155 - 164 TF/s-equivalent.
#include "gemm_internal.h"
#include "split_arith.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

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
    float* colsum; long colsum_batch;      // A column-contiguous only: colsum[g][m] = sum_k A[g][k][m]
    int xcd_group, batch;                  // > 0: tiles per batch member, all on one XCD (see the kernel)
    int vec_epi;                           // float4-clean output side (N, ldc, res_ld multiples of 4, 16-byte aligned bases): row-major epilogue
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
template <bool ANC, bool BNC, bool EXT>
__global__ __launch_bounds__(NTH, 3) void gemm_s6_kernel(S6K g) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * BUF];
    // block -> (batch member, m-tile, n-tile), XCD-aware (workgroup w runs on XCD w % 8, each XCD has its own L2)
    int t = blockIdx.x;
    int tm, tn, bz;
    if (g.xcd_group > 0) {
        // long reductions over few tiles (weight gradients): the tiles of one batch member read the same two operand panels from
        // HBM.  Workgroup w runs on XCD w % 8 and every XCD has its own L2: all tiles of a member go to ONE XCD, next to each
        // other in its dispatch order (member = XCD + 8 * round), so that each panel is fetched once instead of once per tile
        // column / row (110 x [256 x 256 x 6400]: 2.9 -> 1.4 GB per launch).
        const int q = t >> 3, rnd = q / g.xcd_group;
        const int member = (t & 7) + 8 * rnd;
        if (member >= g.batch) return;
        t = q - rnd * g.xcd_group;
        tm = t % g.tiles_m;
        tn = t / g.tiles_m;
        bz = member;
    } else if (g.tiles_n > 1) {
        // the n-tiles of one 128-row panel of A (activations, streamed from HBM; B = a weight matrix that stays in L2) on one XCD,
        // next to each other: panel = XCD + 8 * round
        const int q = t >> 3, rnd = q / g.tiles_n;
        const int panel = (t & 7) + 8 * rnd;
        if (panel >= g.batch * g.tiles_m) return;
        tn = q - rnd * g.tiles_n;
        tm = panel % g.tiles_m;
        bz = panel / g.tiles_m;
    } else {
        tm = t % g.tiles_m;
        tn = 0;
        bz = t / g.tiles_m;
    }
    const int m0 = tm * TB, n0 = tn * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const bool seg = EXT && g.k_seg > 0;
    const long* a_seg = seg ? g.a_seg_off + (long)bz * g.nseg : nullptr;
    const long* b_seg = seg ? g.b_seg_off + (long)bz * g.nseg : nullptr;
    gptr Au = uniform_ptr(g.A + (seg ? a_seg[0] : g.a_off ? g.a_off[bz] : (long)bz * g.a_batch));
    gptr Bu = uniform_ptr(g.B + (seg ? b_seg[0] : g.b_off ? g.b_off[bz] : (long)bz * g.b_batch));
    // the NEXT segment's bases are fetched a segment ahead (scalar loads that have 16 k-tiles to arrive)
    long a_next = seg ? a_seg[min(1, g.nseg - 1)] : 0, b_next = seg ? b_seg[min(1, g.nseg - 1)] : 0;
    // An operand tile is 128 rows (m or n) x 16 k.  Reduction-contiguous operand: thread -> rows (tid >> 2) and (tid >> 2) + 64,
    // 16-byte chunk tid & 3 of the row's 16 k (two dwordx4 loads, 3 x 8-byte LDS writes each) into a [row][16 k] image that
    // ds_read_b128 reads.  Row-contiguous operand (X[k][row]): two dwordx4 loads of 4 consecutive rows at one k each (a wave
    // reads 2 x 512 consecutive bytes), the same 3 x 8-byte writes into a [k][128 rows] image, and the fragment (8 consecutive k
    // of the lane's row) comes out of two ds_read_b64_tr_b16, gfx950's transposing LDS read.  (The first form of this path
    // loaded 8 dwords of consecutive k per thread: one vector-memory instruction per 256 bytes made the texture addresser the
    // bound -- 110 x [256 x 256 x 6400] weight gradients: 692 us.)
    struct Lay { unsigned off[2]; int wr[2]; unsigned ld4; };
    auto layout = [&](bool nc, int r0, int rows, long ld) {
        Lay y;
        if (!nc) {
            const int lrow = tid >> 2, lch = tid & 3;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = lrow + 64 * q;
                y.off[q] = (unsigned)((long)min(r0 + r, rows - 1) * ld + lch * 4) * 4u;    // bytes (operands < 4 GB: host check)
                y.wr[q] = r * 32 + (((lch >> 1) ^ ((r >> 3) & 1)) * 16) + (lch & 1) * 8;
            }
            y.ld4 = 4u;                                     // byte step per k
        } else {
            // k-major image [16 k][128 rows] per plane (256-byte image rows, 16-byte chunks XOR-swizzled: image (b) of the
            // guide's T10): thread -> k = tid / 32 (+ 8), rows 4 (tid % 32) .. + 3
            const int c4 = tid & 31, kq = tid >> 5;
            y.ld4 = (unsigned)ld * 4u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int kk = kq + 8 * q;
                y.off[q] = (unsigned)min(r0 + 4 * c4, rows - 4) * 4u + (unsigned)kk * y.ld4;
                y.wr[q] = 256 * kk + 16 * ((c4 >> 1) ^ (((kk & 3) << 2) | ((kk >> 2) & 3))) + 8 * (c4 & 1);
            }
        }
        return y;
    };
    const Lay la = layout(ANC, m0, g.M, g.lda), lb = layout(BNC, n0, g.N, g.ldb);
    const int nk = g.K / BK;
    const int kspan = seg ? g.k_seg : g.K;     // k range addressed from the current bases
    int ld_t = 0, ld_k = 0, ld_seg = 0;        // the next load: k-tile, k inside the segment, segment (all wave-uniform)
    struct Regs { f32x4 a[2], b[2]; };
    auto load_op = [&](auto ncc, f32x4 (&v)[2], gptr base, const Lay& y) {
        const unsigned ko = (unsigned)ld_k * y.ld4;
        if constexpr (!decltype(ncc)::value) {
#pragma unroll
            for (int q = 0; q < 2; ++q) v[q] = *reinterpret_cast<gptr_f4>(base + (y.off[q] + ko));
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) v[q] = *reinterpret_cast<gptr_f4>(base + (y.off[q] + ko));
        }
    };
    auto load = [&](Regs& x) {
#if !defined(AS_S6G_ABL) || AS_S6G_ABL != 1     // ablation 1 (diagnostic builds): no global loads
        load_op(IC2<ANC>{}, x.a, Au, la);
        load_op(IC2<BNC>{}, x.b, Bu, lb);
#endif
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
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};   // ANC with g.colsum: this thread's share of the column sums of A (its 4 columns, its 2 of every 16 k)
    auto store_op = [&](const f32x4 (&v)[2], unsigned char* base, const Lay& y) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split_pair(v[q].x, v[q].y, h0, m0_, l0);
            split_pair(v[q].z, v[q].w, h1, m1, l1);
            unsigned char* d = base + y.wr[q];
            *reinterpret_cast<u32x2*>(d) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(d + PLANE) = (u32x2){m0_, m1};
            *reinterpret_cast<u32x2*>(d + 2 * PLANE) = (u32x2){l0, l1};
        }
    };
    int st_t = 0;       // k-tile the next store() holds (behind the last one the stores repeat it)
    auto store = [&](const Regs& x, int buf) {
        unsigned char* base = sm + buf * BUF;
        if constexpr (ANC) {
            if (g.colsum && st_t < nk) csum += x.a[0] + x.a[1];
            ++st_t;
        }
        store_op(x.a, base, la);
        store_op(x.b, base + OPER, lb);
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
    // k-major images: lane 4q + p of a 16-lane group addresses row (= k) 8 lh + 4 jj + q, image columns 4p .. 4p + 3 of the group's 16
    // (ds_read_b64_tr_b16 hands lane i of the group column i of the four rows)
    typedef __attribute__((address_space(3))) s16x4* lds_tr;
    unsigned tr_a[2][2], tr_b[2][2];
    {
        const int p = lane & 3, q = (lane >> 2) & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int row = 8 * lh + 4 * jj + q;
                const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
                const int ca = wm * 64 + i * 32 + g1 * 16 + 4 * p, cb = wn * 64 + i * 32 + g1 * 16 + 4 * p;
                tr_a[i][jj] = 256 * row + 16 * ((ca >> 3) ^ sw) + 8 * ((ca >> 2) & 1);
                tr_b[i][jj] = OPER + 256 * row + 16 * ((cb >> 3) ^ sw) + 8 * ((cb >> 2) & 1);
            }
    }
    auto frag_tr = [&](unsigned o0, unsigned o1, int imm) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(sm + o0 + imm));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(sm + o1 + imm));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    Regs x[2];
    load(x[0]);
    load(x[1]);
    if (EXT && g.res && !g.vec_epi) {   // scalar epilogue: the accumulators start from `res` (requested behind the first two tiles' loads)
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
#if defined(AS_S6G_ABL) && AS_S6G_ABL == 2      // ablation 2: plain 16-byte reads in place of the transposing ones (wrong numbers)
                fa[i][p] = *reinterpret_cast<const bf16x8*>(a_rd + U * BUF + p * PLANE + i * 32 * 32);
                fb[i][p] = *reinterpret_cast<const bf16x8*>(b_rd + U * BUF + p * PLANE + i * 32 * 32);
                continue;
#endif
                if constexpr (ANC) fa[i][p] = frag_tr(tr_a[i][0], tr_a[i][1], U * BUF + p * PLANE);
                else fa[i][p] = *reinterpret_cast<const bf16x8*>(a_rd + U * BUF + p * PLANE + i * 32 * 32);
                if constexpr (BNC) fb[i][p] = frag_tr(tr_b[i][0], tr_b[i][1], U * BUF + p * PLANE);
                else fb[i][p] = *reinterpret_cast<const bf16x8*>(b_rd + U * BUF + p * PLANE + i * 32 * 32);
            }
        constexpr int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {0, 1, 2, 0, 1, 0};   // without mid.lo, lo.mid, lo.lo
#pragma unroll
        for (int o = 0; o < 6; ++o)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[o]], fb[j][PB[o]], acc[i][j], 0, 0, 0);
#if !defined(AS_S6G_ABL) || AS_S6G_ABL != 3     // ablation 3: no split + LDS writes
        store(x[U ^ 1], U ^ 1);   // (behind the last tile: a clamped repeat into the idle buffer)
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        tile(IC2<0>{});
        tile(IC2<1>{});
    }
    if (kt < nk) tile(IC2<0>{});

    if constexpr (ANC) {   // column sums of A (a bias gradient): the two k halves of a column meet in LDS, in a fixed order
        if (g.colsum && tn == 0) {
            f32x4* cs = reinterpret_cast<f32x4*>(sm);        // [8 k groups][32 column quads]
            cs[tid] = csum;
            __syncthreads();
            if (tid < TB) {
                const float* c1 = reinterpret_cast<const float*>(sm);
                float t8 = 0.f;
#pragma unroll
                for (int kq = 0; kq < 8; ++kq) t8 += c1[kq * TB + tid];
                // (a clamped quad -- M % 128 != 0 -- repeats valid columns at the tile's end: those lanes hold other columns' sums)
                if (m0 + tid < g.M && (m0 + (tid & ~3) + 4 <= g.M)) g.colsum[(long)bz * g.colsum_batch + m0 + tid] = t8;
            }
        }
    }
    // ---- epilogue: D[i][j] block (i, j) of the wave: row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = n0 + wn * 64 + j * 32 + l31
    float* C = g.C + (g.c_off ? g.c_off[bz] : (long)bz * g.c_batch);
    const float* bias = g.bias ? g.bias + (g.bias_off ? g.bias_off[bz] : (long)bz * g.bias_batch) : nullptr;
    if (g.vec_epi) {
        // Row-major epilogue: the accumulators leave in two halves of 64 rows through LDS ([64][128 + 4] floats: the matrix
        // instruction's layout gives a lane one column of 16 rows -- 64 dword stores, and as many dword loads each for `res` and
        // the mask words), and every thread handles float4s of consecutive columns: 8 stores, 8 `res` loads, 8 mask words
        // per thread and half.  At K = 256 the old form issued three times the vector-memory instructions of the main loop
        // (the extended input-gradient launches ran at 77 TF/s-equivalent against 178 for the plain ones).
        // `res` is added here, behind the reduction: (sum_k) + res + bias -- the fp32 kernel starts its accumulators from it.
        constexpr int LD = TB + 4;
        float* es = reinterpret_cast<float*>(sm);
        const float* R = (EXT && g.res) ? g.res + (g.res_off ? g.res_off[bz] : (long)bz * g.res_batch) : nullptr;
        const unsigned* mask = (EXT && g.mask_bits) ? g.mask_bits + (long)bz * g.mask_batch : nullptr;
        unsigned* bits = g.relu_bits ? g.relu_bits + (long)bz * g.relu_bits_batch : nullptr;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            es[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LD + wn * 64 + j * 32 + l31] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int id = tid + NTH * u, rr = id >> 5, c4 = id & 31;
                const int row = m0 + 64 * half + rr, col = n0 + 4 * c4;
                const bool ok = row < g.M && col < g.N;      // (N % 4 == 0: a float4 is inside or outside as a whole)
                f32x4 v = *reinterpret_cast<const f32x4*>(es + rr * LD + 4 * c4);
                if (ok) {
                    if (R) v += *reinterpret_cast<const f32x4*>(R + (long)row * g.res_ld + col);
                    if (bias) v += *reinterpret_cast<const f32x4*>(bias + col);
                    if (g.act == 1) { v.x = as_relu(v.x); v.y = as_relu(v.y); v.z = as_relu(v.z); v.w = as_relu(v.w); }
                    else if (g.act == 2) { v.x = as_sigmoid(v.x); v.y = as_sigmoid(v.y); v.z = as_sigmoid(v.z); v.w = as_sigmoid(v.w); }
                    if (mask) {
                        const unsigned nib = mask[(long)row * g.ncb + (col >> 5)] >> (col & 31);
                        v.x = (nib & 1u) ? v.x : 0.f; v.y = (nib & 2u) ? v.y : 0.f; v.z = (nib & 4u) ? v.z : 0.f; v.w = (nib & 8u) ? v.w : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(C + (long)row * g.ldc + col) = v;
                }
                if (bits) {   // the ReLU's bit image: eight consecutive lanes hold the 32 columns of a word
                    unsigned nib = 0;
                    if (ok) nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
                    unsigned wbits = nib << (4 * (c4 & 7));
                    wbits |= __shfl_xor(wbits, 1);
                    wbits |= __shfl_xor(wbits, 2);
                    wbits |= __shfl_xor(wbits, 4);
                    const int word = col >> 5;
                    if ((c4 & 7) == 0 && row < g.M && word < g.ncb) bits[(long)row * g.ncb + word] = wbits;
                }
            }
            __syncthreads();
        }
        return;
    }
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
    const bool anc = g->a_k != 1;     // A[k][m], m contiguous (weight-gradient orientation)
    const bool bnc = g->b_k != 1;     // B[k][n], n contiguous (input- and weight-gradient orientation)
    if ((anc && (g->a_i != 1 || !bnc)) || (bnc && g->b_j != 1) || g->K < BK || g->K % BK || g->act < 0 || g->act > 2) return 0;
    if (g->k_tri || g->accumulate || g->b_kT || g->b_kshift || (g->precision != 0 && g->precision != 3)) return 0;
    if (g->colsum && !anc) return 0;
    const bool ext = g->res || g->mask_bits || g->k_seg;
    if (anc && (ext || g->relu_bits || g->bias)) return 0;
    if (g->relu_bits && (g->act != 1 || ext)) return 0;
    if (!anc && (g->a_i % 4 || (reinterpret_cast<uintptr_t>(g->A) & 15) || (!g->a_off && !g->k_seg && g->a_batch % 4))) return 0;
    if (!bnc && (g->b_j % 4 || (reinterpret_cast<uintptr_t>(g->B) & 15) || (!g->b_off && !g->k_seg && g->b_batch % 4))) return 0;
    // row-contiguous operands are loaded as float4s along the rows
    if (anc && (g->a_k % 4 || g->M % 4 || (reinterpret_cast<uintptr_t>(g->A) & 15) || (!g->a_off && g->a_batch % 4))) return 0;
    if (bnc && (g->b_k % 4 || g->N % 4 || (reinterpret_cast<uintptr_t>(g->B) & 15) || (!g->b_off && !g->k_seg && g->b_batch % 4))) return 0;
    if (g->k_seg && (g->k_seg % BK || g->K % g->k_seg || !g->a_seg_off || !g->b_seg_off)) return 0;
    const long kspan = g->k_seg ? g->k_seg : g->K;
    // 32-bit byte offsets inside one batch member / segment
    if ((anc ? kspan * g->a_k + g->M : (long)g->M * g->a_i) >= (1L << 30) || (bnc ? kspan * g->b_k + g->N : (long)g->N * g->b_j) >= (1L << 30)) return 0;
    S6K k{};
    k.A = g->A; k.lda = anc ? g->a_k : g->a_i; k.a_batch = g->a_batch;
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
    k.colsum = g->colsum; k.colsum_batch = g->colsum_batch;
    {
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        static const bool no_vec = AS_DIAG_SET("AS_S6_SCALAR_EPI");   // ablation: the one-column-per-lane epilogue
        k.vec_epi = !no_vec && g->N % 4 == 0 && g->ldc % 4 == 0 && al16(g->C) && (g->c_off || g->c_batch % 4 == 0) &&
                    (!g->bias || (al16(g->bias) && (g->bias_off || g->bias_batch % 4 == 0))) &&
                    (!g->res || (al16(g->res) && g->res_ld % 4 == 0 && (g->res_off || g->res_batch % 4 == 0)));
    }
    long blocks = (long)k.tiles_m * k.tiles_n * g->batch;
    if (blocks > (1L << 30)) return 0;
    k.batch = g->batch;
    if (anc) {
        // one workgroup walks the whole reduction of its tile: worth it once the tiles fill the chip (else wgrad_f32.hip's stream-K)
        if (blocks < 256 && !AS_DIAG_SET("AS_S6_TN_ANY")) return 0;
        k.xcd_group = k.tiles_m * k.tiles_n;
        blocks = (long)as_round_up(g->batch, 8) * k.xcd_group;
    } else if (k.tiles_n > 1) {
        blocks = (long)as_round_up((long)g->batch * k.tiles_m, 8) * k.tiles_n;
    }
    const dim3 grid((unsigned)blocks), blk(NTH);
    if (anc) hipLaunchKernelGGL((gemm_s6_kernel<true, true, false>), grid, blk, 0, st, k);
    else if (ext) {
        if (bnc) hipLaunchKernelGGL((gemm_s6_kernel<false, true, true>), grid, blk, 0, st, k);
        else hipLaunchKernelGGL((gemm_s6_kernel<false, false, true>), grid, blk, 0, st, k);
    } else {
        if (bnc) hipLaunchKernelGGL((gemm_s6_kernel<false, true, false>), grid, blk, 0, st, k);
        else hipLaunchKernelGGL((gemm_s6_kernel<false, false, false>), grid, blk, 0, st, k);
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
