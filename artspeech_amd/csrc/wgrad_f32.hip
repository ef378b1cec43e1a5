// Weight-gradient GEMM  C[g][M][N] = sum_k A[g][k][m] * B[g][k][n]  (both operands reduction-strided: the rows of A and B
// are the frames, dW = dY^T X) on the fp32 matrix core.  This is the shape behind every nn.Linear / GRU weight gradient of
// the path (encoder_decoder/models.py:7-33, 111, 113-116 through autograd): a small output (at most a few hundred columns
// per batch member) under a reduction over all B*T frames.
//
// Why a kernel of its own (gemm_f32.hip's general kernel serves it with 64x64 tiles + split-K):
//   * with 64x64 (or 128x128) output tiles every k-panel of A and B is fetched by 4 (2) sibling workgroups that sit on
//     different XCDs, i.e. different L2s -- the launch moves 2-4x its operand bytes through the fabric.  Here ONE workgroup
//     owns a 128 x 256 (or 128 x 128) output tile, so for the head layers (N <= 256) a B panel is fetched by the M-tiles of
//     one XCD only (the block index -> work map puts the M-tiles of a (batch, k-slab, n-tile) on one XCD, back to back);
//   * 8 waves (2 per SIMD), each with a 64 x 64 (64 x 32) sub-tile = 4 (2) MFMAs per 2 + 2 (2 + 1) operand reads, the reads
//     of k-step s + 1 issued ahead of the MFMAs of k-step s;
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4) in a ring of three 32-deep k-tiles: two k-tiles (96 KB at
//     BN = 256) are in flight per CU while the third is multiplied; a counted s_waitcnt vmcnt + ONE raw barrier per k-tile.
//     The image is a straight copy of the global rows ([k][m]), which already is what an MFMA operand wants (lane i reads
//     column i of row k: conflict-free ds_read_b32).  (First version: register staging, one 16-deep k-tile in flight -- it
//     ran at the latency of its own loads: loads-only ablation 766 us, MFMA-only 874 us, both 1030 us on the grouped
//     transformer shape.  A second register set spilled.)
//   * the fused bias gradient (column sums of A) costs 8 LDS reads per thread and k-tile, combined in a fixed order;
//   * deterministic split-K over slabs with a wide reduce kernel (fixed summation order: results do not depend on the
//     order in which workgroups are scheduled); the split comes from a small cost model over whole rounds of workgroups
//     and slab traffic, for the number of CUs the caller expects to be free (as_gemm.cu_budget).
//   * SEVERAL problems in one launch (as_wgrad_multi): the weight gradients of the three head layers and of the trunk Linear
//     are 46 tiles of 128 x 256 under K = 6400 in all.  Launched one by one, each fills 176 of 256 CUs with one long
//     workgroup per CU and pays its own prologue, tail and reduce launch; as ONE grid of k-chunks of every problem the
//     dispatcher keeps every free CU busy until the work is gone, and one reduce launch sums all slabs.  A problem may ask
//     for the column sums of its B operand (bias gradient of the transposed formulation) and for a transposed result.
#include <cstdlib>

#include "gemm_internal.h"
#include "split_arith.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, NT = 512, NBUF = 3;
constexpr int KALIGN = 32;   // reduction chunks are multiples of both k-tile depths

// source of the shifted B operand's out-of-sequence rows (b_kT > 0): one row of zeros the DMA can read
__device__ __attribute__((aligned(16))) float g_zero_row[256];

constexpr int MAXP = 6;   // problems per launch

struct WgradK {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    long a_batch, b_batch, c_batch;
    const long* a_off; const long* b_off; const long* c_off;
    int batch, tiles_m, tiles_n, splitk, kchunk;
    long ncombos;
    int per_xcd;
    int b_kshift, b_kT, b_kshift_batch;
    int accumulate, c_vec;
    float* slab;     // [splitk][batch][M][N]
    float* cs_slab;  // [splitk][batch][M]
    float* colsum; long colsum_batch;
    // multi-problem launches: first work item / first reduce thread of this problem; optional column sums of B
    // (colsum_b [batch][N], slabs [splitk][batch][N]); c_trans: the result is stored transposed, C[n * ldc + m]
    long item0, red0;
    float* colsum_b; long colsum_b_batch; float* csb_slab;
    int c_trans;
    long tile0;   // stream-K launches: first output tile of this problem in the launch's tile numbering
#ifdef AS_DIAG
    int abl;  // diagnostic ablation (AS_WGRAD_ABL=1): no global loads after the first two k-tiles (matrix work only)
#else
    static constexpr int abl = 0;
#endif
};

// LDS-DMA helper: one wave-instruction copies 64 x 16 B from per-lane global addresses to 1 KiB of LDS starting at the
// wave-uniform `dst` (global_load_lds_dwordx4: no VGPR destination, counted by vmcnt).  Inline asm: with the builtin hipcc
// may put an s_waitcnt vmcnt(0) in front of the next ds_read (it did in the BK = 16 instantiation), which drains the ring.
__device__ __forceinline__ void glds16(const float* src, float* dst) {
    unsigned keep;
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds)
                 : "memory");
}

// BK = 32: one workgroup per CU (144 KB of LDS at BN = 256, two k-tiles = 96 KB in flight).  BK = 16: 72 KB, two workgroups
// per CU -- the partner's MFMAs cover this one's barriers, prologue and epilogue.
// Stream-K (streamk = 1): the launch's work is the flat sequence of k-tiles (32 frames) of all its output tiles, tile after
// tile; workgroup w takes units [w U, (w + 1) U).  A tile that one workgroup covers from its first k-tile to its last is
// written as usual; a tile cut by a workgroup boundary leaves one PIECE per workgroup in the slab (slot 0: the piece a
// workgroup starts with in the middle of a tile, slot 1: a piece that starts a tile and stops short of its end) and
// wgrad_reduce_sk_kernel adds a tile's pieces in k order (deterministic).  Every CU works for the same time whatever the
// tile count is (220 tiles of a 110-block transformer group leave 36 of 256 CUs idle for the whole launch otherwise), and
// the slab holds at most two pieces per workgroup instead of a split factor times the whole result.
constexpr int PIECE_FLOATS = BM * 256 + BM + 256;   // tile + column sums of the A and of the B operand
struct WgradMulti {
    int n, per_xcd;
    long total_items, total_red;
    int streamk, nkt;           // nkt = K / 32, the same for every problem of a stream-K launch
    long unit_per_wg, total_units, total_tiles;
    float* pieces;              // [workgroups][2][PIECE_FLOATS]
    WgradK p[MAXP];
};

// SK: the stream-K instantiation (mm.streamk launches); the plain one compiles to a single pass of the segment loop.
// S6 (BK = 32): the products on the bfloat16 matrix instruction (as_set_matrix_arith(1): three planes per operand, six plane
//   products per fp32 product, fp32 accumulation; see lin_f32.hip).  The fp32 image in LDS stays as it is ([k][m]: what the
//   DMA delivers); a lane's MFMA fragment -- 8 consecutive frames of one column -- is read down the image (8 ds_read_b32,
//   conflict-free) and split in registers (36 vector instructions).  Every wave splits the fragments it multiplies: the A
//   block of a wave is split by the four waves that share its rows, the B block by two -- redundant vector work, but no
//   second LDS image (the ring already takes 144 KB) and no extra barrier.  The split of k-step s + 1 is interleaved with
//   the matrix instructions of k-step s (about seven vector instructions fit under one 32-cycle MFMA), and the barrier
//   that publishes k-tile t + 1 sits in the MIDDLE of tile t, so that the first fragments of t + 1 are split under the second
//   half of t's matrix work.  Per 32-frame k-tile a wave issues 48 MFMAs of 32 cycles instead of 64 of 64.
template <int BN, int BK, bool SK = false, bool S6 = false>
__global__ __launch_bounds__(NT, BK == 16 ? 4 : 2) void wgrad_f32_kernel(WgradMulti mm) {
    static_assert(!S6 || BK == 32, "split arithmetic: 32-deep k-tiles");
    constexpr int WN = BN / 4, TN = WN / 32, TM = 2;   // 2 x 4 waves; a wave owns 64 rows x WN columns
    constexpr int TILE = BK * (BM + BN);                // floats per ring slot: A image [BK][BM] then B image [BK][BN]
    constexpr int PA = BK * BM / 256 / 8;               // 1-KiB DMA pieces of A per wave and k-tile (2)
    constexpr int PB = BK * BN / 256 / 8;               // ... of B (4 at BN = 256, 2 at BN = 128)
    constexpr int RB = 256 / BN;                        // B rows per piece (1 or 2)
    constexpr int CSR = BK / 4;                         // rows of a k-tile each of the four colsum row groups adds
    // ONE shared array (a second __shared__ object beside an LDS-DMA target makes hipcc drain the DMAs before every read)
    __shared__ __attribute__((aligned(16))) float smem[NBUF * TILE];

    // block -> work: blocks b and b + 8 share an XCD (round-robin dispatch; speed only).  Work items are ordered
    // (batch, k-slab, n-tile, m-tile) and every XCD takes one contiguous eighth of them: the M-tiles of one
    // (batch, k-slab, n-tile) combination, which read the same B panel, sit on one XCD (two at a seam), and the XCDs get
    // equal shares whatever the counts are.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= mm.per_xcd) return;
    const long wg = (long)xcd * mm.per_xcd + slot;
    if (wg >= mm.total_items) return;
    // stream-K: this workgroup's k-tile units [u, u_end) of the flat (tile, k-tile) sequence, one SEGMENT (the part inside one
    // output tile) per pass of the loop below; otherwise one pass for the work item `wg`
    long u = SK ? wg * mm.unit_per_wg : 0;
    const long u_end = SK ? min(u + mm.unit_per_wg, mm.total_units) : 1;
  for (bool first_seg = true; u < u_end; first_seg = false) {
    long item = wg;
    int kt0 = 0, kt1 = 0;
    if (SK) {
        item = u / mm.nkt;
        kt0 = (int)(u - item * mm.nkt);
        kt1 = (int)min((long)mm.nkt, kt0 + (u_end - u));
        u += kt1 - kt0;
    } else {
        u = u_end;
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i)
        if (i < mm.n && item >= (SK ? mm.p[i].tile0 : mm.p[i].item0)) pi = i;
    const WgradK& g = mm.p[pi];
    item -= SK ? g.tile0 : g.item0;
    const int tm = (int)(item % g.tiles_m);
    const long combo = item / g.tiles_m;
    const int tn = (int)(combo % g.tiles_n);
    const long rest = combo / g.tiles_n;
    const int ks = (int)(rest % g.splitk), bz = (int)(rest / g.splitk);   // (stream-K problems have splitk == 1)
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = SK ? kt0 * BK : ks * g.kchunk;
    const int kend = SK ? kt1 * BK : min(g.K, kbeg + g.kchunk);   // both multiples of BK (host contract)
    // a piece: part of a tile's reduction only -> the slab, summed by wgrad_reduce_sk_kernel
    const bool piece = SK && !(kt0 == 0 && kt1 == mm.nkt);
    float* const piece_out = SK ? mm.pieces + (wg * 2 + (kt0 > 0 ? 0 : 1)) * PIECE_FLOATS : nullptr;
    if (SK && !first_seg) __syncthreads();   // the previous segment's epilogue may still be reading the LDS the DMAs below overwrite
    const float* __restrict__ A = g.A + (g.a_off ? g.a_off[bz] : (long)bz * g.a_batch);
    const float* __restrict__ B = g.B + (g.b_off ? g.b_off[bz] : (long)bz * g.b_batch);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const bool do_cs = g.colsum != nullptr && tn == 0;
    const bool do_csb = g.colsum_b != nullptr && tm == 0;

    // DMA sources.  A piece = 2 rows x 128 floats: lanes 0-31 row r, lanes 32-63 row r + 1.  Columns beyond M / N are
    // redirected to the last valid float4 of the row: they only feed output rows / columns that are never stored.
    const int a_col = min(m0 + l31 * 4, g.M - 4);
    const float* a_src = A + (long)(kbeg + wave * PA * 2 + lh) * g.lda + a_col;           // + j * 2 rows, + kt * BK rows
    const int b_lane_col = RB == 1 ? lane * 4 : l31 * 4;
    const int b_col = min(n0 + b_lane_col, g.N - 4);
    const float* b_src = B + (long)(kbeg + wave * PB * RB + (RB == 2 ? lh : 0)) * g.ldb + b_col;  // + j * RB rows
    float* const a_dst = smem + wave * PA * 256;             // + j * 256, + slot * TILE
    float* const b_dst = smem + BK * BM + wave * PB * 256;
    // Time-shifted B (b_kT > 0; dW_hh = dgh^T . h_prev, models.py:111 through autograd): reduction row k is frame t = k % kT
    // of its sequence and reads row k + shift, or zeros when t + shift leaves [0, kT) (h_{-1} = 0).  The DMA takes per-lane
    // source addresses, so such a row simply comes from g_zero_row.  Each piece keeps its frame index and advances it by BK
    // per k-tile (no division in the loop).
    const int kshift = g.b_kT > 0 ? g.b_kshift + bz * g.b_kshift_batch : 0;
    int b_t[PB];
    if (g.b_kT > 0) {
#pragma unroll
        for (int j = 0; j < PB; ++j) b_t[j] = (kbeg + wave * PB * RB + (RB == 2 ? lh : 0) + j * RB) % g.b_kT;
    }
    const float* const zero_src = g_zero_row + (RB == 1 ? lane * 4 : l31 * 4);
    auto issue = [&](int kt) {  // the PA + PB DMA pieces of this wave for k-tile kt
        float* base = smem + (kt % NBUF) * TILE;
        const long koff = (long)kt * BK;
#pragma unroll
        for (int j = 0; j < PA; ++j) glds16(a_src + (koff + j * 2) * g.lda, base + (a_dst - smem) + j * 256);
        if (g.b_kT > 0) {
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int t = b_t[j] + kshift;
                const bool ok = t >= 0 && t < g.b_kT;
                glds16(ok ? b_src + (koff + j * RB + kshift) * g.ldb : zero_src, base + (b_dst - smem) + j * 256);
                b_t[j] += BK;
                while (b_t[j] >= g.b_kT) b_t[j] -= g.b_kT;
            }
        } else {
#pragma unroll
            for (int j = 0; j < PB; ++j) glds16(b_src + (koff + j * RB) * g.ldb, base + (b_dst - smem) + j * 256);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float cs = 0.f;  // bias gradient: this thread sums column tid & 127 over rows (tid >> 7) * 8 .. + 7 of every k-tile
    float csb = 0.f; // column sums of B: column tid % BN over rows (tid / BN) * CSRB .. of every k-tile
    constexpr int CSRB = BK / (NT / BN);

    const int nk = (kend - kbeg) / BK;
    if constexpr (S6) {
        struct Fr { bf16x8 p[3]; };
        // fragment of column `col` (pointer to its element in row 0 of the image), rows 16 s + 8 lh .. + 7
        auto frag = [&](const float* col, int ld, int s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = col[(16 * s + 8 * lh + e) * ld];
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], h[q], m[q], l[q]);
            Fr f;
            f.p[0] = __builtin_bit_cast(bf16x8, (u32x4w){h[0], h[1], h[2], h[3]});
            f.p[1] = __builtin_bit_cast(bf16x8, (u32x4w){m[0], m[1], m[2], m[3]});
            f.p[2] = __builtin_bit_cast(bf16x8, (u32x4w){l[0], l[1], l[2], l[3]});
            return f;
        };
        auto load_frags = [&](Fr (&fa)[TM], Fr (&fb)[TN], int kt, int s) {
            const float* tile = smem + (kt % NBUF) * TILE;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = frag(tile + wm * 64 + i * 32 + l31, BM, s);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = frag(tile + BK * BM + wn * WN + j * 32 + l31, BN, s);
        };
        auto mma = [&](const Fr (&fa)[TM], const Fr (&fb)[TN]) {
            constexpr int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {0, 1, 2, 0, 1, 0};   // without mid.lo, lo.mid, lo.lo
#pragma unroll
            for (int o = 0; o < 6; ++o)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].p[PA[o]], fb[j].p[PB[o]], acc[i][j], 0, 0, 0);
        };
        // one MFMA, then the vector / LDS instructions that fit under it: the scheduler interleaves the next k-step's split
        auto interleave = [&]() {
#pragma unroll
            for (int q = 0; q < 6 * TM * TN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 LDS reads
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
            }
        };
        issue(0);
        if (nk > 1) issue(1);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        Fr fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        load_frags(fa0, fb0, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            // slot (kt + 2) % 3 was last read (k-step 1 of tile kt - 1) before the barrier in the middle of tile kt - 1
            if (kt + 2 < nk && g.abl != 1) issue(kt + 2);
            const float* tile = smem + (kt % NBUF) * TILE;
            if (do_cs) {
                const float* c_s = tile + (tid >> 7) * CSR * BM + (tid & 127);
#pragma unroll
                for (int r = 0; r < CSR; ++r) cs += c_s[r * BM];
            }
            if (do_csb) {
                const float* c_s = tile + BK * BM + (tid / BN) * CSRB * BN + (tid % BN);
#pragma unroll
                for (int r = 0; r < CSRB; ++r) csb += c_s[r * BN];
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(fa0, fb0);
            load_frags(fa1, fb1, kt, 1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            // retire tile kt + 1 (leave kt + 2 in flight) and publish it; this wave's reads of tile kt are all issued
            if (kt + 2 < nk && g.abl != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mma(fa1, fb1);
            if (kt + 1 < nk) {
                load_frags(fa0, fb0, kt + 1, 0);
                interleave();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the bias-gradient exchange below reuses the ring's memory
    } else {
    // Ring of NBUF = 3 slots: while k-tile t is multiplied, t + 1 and t + 2 are in flight (2 x 48 KB per CU at BN = 256).
    // A counted vmcnt retires tile t + 1 at the END of iteration t (it was issued at the start of iteration t - 1: two
    // iterations of matrix work to land), then a raw barrier publishes it; __syncthreads() would drain every DMA.
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        // slot (kt + 2) % 3 was read in iteration kt - 1; every wave passed the barrier that ended it
        if (kt + 2 < nk && g.abl != 1) issue(kt + 2);
        const float* tile = smem + (kt % NBUF) * TILE;
        const float* a_s = tile + wm * 64 + l31;
        const float* b_s = tile + BK * BM + wn * WN + l31;
        if (do_cs) {
            const float* c_s = tile + (tid >> 7) * CSR * BM + (tid & 127);
#pragma unroll
            for (int r = 0; r < CSR; ++r) cs += c_s[r * BM];
        }
        if (do_csb) {
            const float* c_s = tile + BK * BM + (tid / BN) * CSRB * BN + (tid % BN);
#pragma unroll
            for (int r = 0; r < CSRB; ++r) csb += c_s[r * BN];
        }
        // operands of k-step kk + 1 are read before the MFMAs of k-step kk are issued (their LDS latency hides behind them)
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[0][i] = a_s[lh * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[0][j] = b_s[lh * BN + j * 32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int c = (kk >> 1) & 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) av[c ^ 1][i] = a_s[(kk + 2 + lh) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[c ^ 1][j] = b_s[(kk + 2 + lh) * BN + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (hipcc otherwise sinks them to their use)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i], bv[c][j], acc[i][j], 0, 0, 0);
        }
        // retire tile kt + 1 (leave kt + 2 in flight), then publish it
        if (kt + 2 < nk && g.abl != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of slot kt % 3 are done
        __builtin_amdgcn_s_barrier();
    }
    }   // !S6

    // ---- bias gradient: the four row groups of a column meet in LDS, summed in a fixed order
    if (do_cs) {
        smem[tid] = cs;   // every fragment read and every DMA is behind the loop's last barrier
        __syncthreads();
        if (tid < BM && (piece || m0 + tid < g.M)) {
            const float s = ((smem[tid] + smem[BM + tid]) + smem[2 * BM + tid]) + smem[3 * BM + tid];
            if (piece) piece_out[BM * 256 + tid] = s;
            else if (g.splitk > 1) g.cs_slab[((long)ks * g.batch + bz) * g.M + m0 + tid] = s;
            else g.colsum[(long)bz * g.colsum_batch + m0 + tid] = s;
        }
    }
    if (do_csb) {
        __syncthreads();   // (the A-side exchange above may still be reading)
        smem[tid] = csb;
        __syncthreads();
        if (tid < BN && (piece || n0 + tid < g.N)) {
            float s = smem[tid];
#pragma unroll
            for (int q = 1; q < NT / BN; ++q) s += smem[q * BN + tid];
            if (piece) piece_out[BM * 256 + BM + tid] = s;
            else if (g.splitk > 1) g.csb_slab[((long)ks * g.batch + bz) * g.N + n0 + tid] = s;
            else g.colsum_b[(long)bz * g.colsum_b_batch + n0 + tid] = s;
        }
    }

    // ---- epilogue: D[i][j], j = lane & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (piece) {   // the whole [BM][BN] tile, dense (rows / columns beyond M / N hold finite don't-cares the reduce never reads)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    piece_out[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wn * WN + j * 32 + l31] = acc[i][j][r];
        continue;
    }
    float* out;
    long ld;
    if (g.splitk > 1) {
        out = g.slab + ((long)ks * g.batch + bz) * g.M * g.N;
        ld = g.N;
    } else {
        out = g.C + (g.c_off ? g.c_off[bz] : (long)bz * g.c_batch);
        ld = g.ldc;
    }
    const bool acc_c = g.splitk == 1 && g.accumulate;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
        if (col >= g.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= g.M) continue;
                float* c = (g.splitk == 1 && g.c_trans) ? out + (long)col * ld + row : out + (long)row * ld + col;
                *c = acc_c ? *c + acc[i][j][r] : acc[i][j][r];
            }
    }
  }   // segments
}

// Stream-K: adds the pieces of every output tile that a workgroup boundary cut, in k order, and stores the tile like the
// main kernel would have (transposed / accumulating / column sums).  One thread per float4 of a tile + one per column sum.
template <int BN>
__global__ __launch_bounds__(256) void wgrad_reduce_sk_kernel(WgradMulti mm) {
    constexpr int PER_TILE = BM * BN / 4 + BM + BN;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long tile = idx / PER_TILE;
    if (tile >= mm.total_tiles) return;
    const int e = (int)(idx - tile * PER_TILE);
    const long ub = tile * mm.nkt, ue = ub + mm.nkt;
    const long w_first = ub / mm.unit_per_wg, w_last = (ue - 1) / mm.unit_per_wg;
    if (w_first == w_last) return;   // one workgroup had the whole tile and stored it
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i)
        if (i < mm.n && tile >= mm.p[i].tile0) pi = i;
    const WgradK& g = mm.p[pi];
    const long item = tile - g.tile0;
    const int tm = (int)(item % g.tiles_m);
    const long combo = item / g.tiles_m;
    const int tn = (int)(combo % g.tiles_n);
    const long bz = combo / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    // piece of workgroup w for this tile: slot 0 if w starts inside the tile, slot 1 if the tile starts inside w
    auto piece_of = [&](long w) { return mm.pieces + (w * 2 + (w * mm.unit_per_wg > ub ? 0 : 1)) * PIECE_FLOATS; };
    if (e < BM * BN / 4) {
        const int row = e / (BN / 4), c4 = e - row * (BN / 4);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long w = w_first; w <= w_last; ++w) {
            const float4 v = *reinterpret_cast<const float4*>(piece_of(w) + row * BN + c4 * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int gr = m0 + row, gc = n0 + c4 * 4;
        if (gr >= g.M || gc >= g.N) return;      // (N is a multiple of 4: a float4 is wholly inside or outside)
        float* cb = g.C + (g.c_off ? g.c_off[bz] : bz * g.c_batch);
        const float v4[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* c = g.c_trans ? cb + (long)(gc + q) * g.ldc + gr : cb + (long)gr * g.ldc + gc + q;
            *c = g.accumulate ? *c + v4[q] : v4[q];
        }
        return;
    }
    const int ce = e - BM * BN / 4;
    if (ce < BM) {                                // column sums of A (bias gradient): the tiles of the first column block hold them
        if (g.colsum == nullptr || tn != 0 || m0 + ce >= g.M) return;
        float s = 0.f;
        for (long w = w_first; w <= w_last; ++w) s += piece_of(w)[BM * 256 + ce];
        g.colsum[bz * g.colsum_batch + m0 + ce] = s;
        return;
    }
    const int cn = ce - BM;
    if (g.colsum_b == nullptr || tm != 0 || n0 + cn >= g.N) return;
    float s = 0.f;
    for (long w = w_first; w <= w_last; ++w) s += piece_of(w)[BM * 256 + BM + cn];
    g.colsum_b[bz * g.colsum_b_batch + n0 + cn] = s;
}

// C (+)= sum over the k-slabs in slab order, every problem of the launch in one grid: per problem first the float4 groups
// of the result, then the bias-gradient slabs (A side, then B side).  c_trans: C[n * ldc + m] (scalar stores).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradMulti mm) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= mm.total_red) return;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i)
        if (i < mm.n && idx >= mm.p[i].red0) pi = i;
    const WgradK& g = mm.p[pi];
    idx -= g.red0;
    if (g.splitk <= 1) return;   // that problem wrote its result directly
    const long per4 = (long)g.M * g.N / 4;
    const long total4 = (long)g.batch * per4;
    if (idx < total4) {
        const float4* s4 = reinterpret_cast<const float4*>(g.slab) + idx;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int k = 0; k < g.splitk; ++k) {
            const float4 v = s4[(long)k * total4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const long bz = idx / per4, rem = (idx - bz * per4) * 4;
        const long row = rem / g.N;
        const int col = (int)(rem - row * g.N);
        float* cb = g.C + (g.c_off ? g.c_off[bz] : bz * g.c_batch);
        if (g.c_trans) {
            float* c = cb + (long)col * g.ldc + row;
            if (g.accumulate) { s.x += c[0]; s.y += c[g.ldc]; s.z += c[2 * g.ldc]; s.w += c[3 * g.ldc]; }
            c[0] = s.x; c[g.ldc] = s.y; c[2 * g.ldc] = s.z; c[3 * g.ldc] = s.w;
            return;
        }
        float* c = cb + row * g.ldc + col;
        if (g.c_vec) {
            float4* c4 = reinterpret_cast<float4*>(c);
            if (g.accumulate) {
                const float4 o = *c4;
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            *c4 = s;
        } else {
            if (g.accumulate) { s.x += c[0]; s.y += c[1]; s.z += c[2]; s.w += c[3]; }
            c[0] = s.x; c[1] = s.y; c[2] = s.z; c[3] = s.w;
        }
        return;
    }
    long e = idx - total4;
    const long n_cs = g.colsum ? (long)g.batch * g.M : 0;
    if (e < n_cs) {
        float s = 0.f;
        for (int k = 0; k < g.splitk; ++k) s += g.cs_slab[(long)k * g.batch * g.M + e];
        const long bz = e / g.M;
        g.colsum[bz * g.colsum_batch + (e - bz * g.M)] = s;
        return;
    }
    e -= n_cs;
    if (g.colsum_b == nullptr || e >= (long)g.batch * g.N) return;
    float s = 0.f;
    for (int k = 0; k < g.splitk; ++k) s += g.csb_slab[(long)k * g.batch * g.N + e];
    const long bz = e / g.N;
    g.colsum_b[bz * g.colsum_b_batch + (e - bz * g.N)] = s;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// shape / alignment contract of one problem; fills the descriptor's operand fields.  false = not a case for this kernel.
bool describe(const as_gemm* g, WgradK& k) {
    if (!(g->a_i == 1 && g->b_j == 1) || g->K < 256 || g->K % KALIGN || g->act != 0 || g->bias) return false;
    if (g->b_kT > 0 && (g->a_off || g->b_off || g->c_off)) return false;   // shifted operand: linear batch strides only
    if (g->M % 4 || g->N % 4 || g->a_k % 4 || g->b_k % 4 || !aligned16(g->A) || !aligned16(g->B)) return false;
    const bool grouped = g->a_off || g->b_off || g->c_off;
    if (!grouped && (g->a_batch % 4 || g->b_batch % 4)) return false;
    if ((long)g->M * g->N % 4) return false;
    k = WgradK{};
    k.A = g->A; k.B = g->B; k.C = g->C;
    k.M = g->M; k.N = g->N; k.K = g->K;
    k.lda = g->a_k; k.ldb = g->b_k; k.ldc = g->ldc;
    k.a_batch = g->a_batch; k.b_batch = g->b_batch; k.c_batch = g->c_batch;
    k.a_off = (const long*)g->a_off; k.b_off = (const long*)g->b_off; k.c_off = (const long*)g->c_off;
    k.batch = g->batch;
    k.b_kshift = g->b_kshift; k.b_kT = g->b_kT; k.b_kshift_batch = g->b_kshift_batch;
    k.accumulate = g->accumulate;
    k.c_vec = aligned16(g->C) && g->ldc % 4 == 0 && (g->c_off ? 1 : g->c_batch % 4 == 0);
    k.colsum = g->colsum; k.colsum_batch = g->colsum_batch;
    return true;
}

template <int BNT>
int launch_multi(WgradMulti& mm, int bk, hipStream_t st, bool exact = false) {
    const dim3 grid((unsigned)(8 * mm.per_xcd));
    const bool s6 = !exact && as_matrix_arith() == AS_ARITH_BF16X6 && bk == 32;
    if (mm.streamk && s6) hipLaunchKernelGGL((wgrad_f32_kernel<BNT, 32, true, true>), grid, dim3(NT), 0, st, mm);
    else if (mm.streamk) hipLaunchKernelGGL((wgrad_f32_kernel<BNT, 32, true>), grid, dim3(NT), 0, st, mm);
    else if (bk == 16) hipLaunchKernelGGL((wgrad_f32_kernel<BNT, 16>), grid, dim3(NT), 0, st, mm);
    else if (s6) hipLaunchKernelGGL((wgrad_f32_kernel<BNT, 32, false, true>), grid, dim3(NT), 0, st, mm);
    else hipLaunchKernelGGL((wgrad_f32_kernel<BNT, 32>), grid, dim3(NT), 0, st, mm);
    AS_LAUNCH_CHECK("as_gemm_f32(wgrad)");
    if (mm.streamk) {
        constexpr long PER_TILE = BM * BNT / 4 + BM + BNT;
        const long threads = mm.total_tiles * PER_TILE;
        hipLaunchKernelGGL((wgrad_reduce_sk_kernel<BNT>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, mm);
        AS_LAUNCH_CHECK("as_gemm_f32(wgrad stream-K reduce)");
        return 0;
    }
    bool any = false;
    for (int i = 0; i < mm.n; ++i) any = any || mm.p[i].splitk > 1;
    if (any) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((mm.total_red + 255) / 256)), dim3(256), 0, st, mm);
        AS_LAUNCH_CHECK("as_gemm_f32(wgrad reduce)");
    }
    return 0;
}

}  // namespace

// Takes the GEMM if it is a weight-gradient shape this kernel is built for (returns 1 and launches), else returns 0;
// negative = error.  Called by as_gemm_f32 ahead of its general tile selection.
int as_wgrad_try(const as_gemm* g, hipStream_t st) {
    static const bool off = AS_DIAG_SET("AS_NO_WGRAD");  // ablation: the general kernel
    if (off) return 0;
    WgradMulti mm{};
    WgradK& k = mm.p[0];
    if (!describe(g, k)) return 0;
#ifdef AS_DIAG
    static const int abl = AS_DIAG_INT("AS_WGRAD_ABL", 0);
    k.abl = abl;
#endif
    const int bn = g->N > 128 ? 256 : 128;
    k.tiles_m = as_cdiv(g->M, BM);
    k.tiles_n = as_cdiv(g->N, bn);
    const long tiles = (long)k.tiles_m * k.tiles_n * g->batch;
    static const bool all_shapes = AS_DIAG_SET("AS_WGRAD_ALL");  // tuning aid: also the shapes below
    // too little work to give every CU a 128-row tile over >= 256 frames: the general kernel's 64 x 64 tiles spread it better
    static const int min_work = AS_DIAG_INT("AS_WGRAD_MIN_WORK", 256);   // tiles x 256-deep k-chunks
    if (!all_shapes && tiles * (g->K / 256) < min_work) return 0;
    // split K so that the launch has about `target` workgroups (one per CU and round); cost model in DESIGN.md 5
    static const int target_env = AS_DIAG_INT("AS_WGRAD_TARGET", 0);
    static const int bk = AS_DIAG_INT("AS_WGRAD_BK", 32);
    const int cus = (g->cu_budget > 0 ? g->cu_budget : 256) * (bk == 16 ? 2 : 1);   // slots: two workgroups per CU at BK = 16
    long S = 1;
    const long per = (long)g->batch * g->M * g->N, per_cs = g->colsum ? (long)g->batch * g->M : 0;
    if (g->splitk_ws && tiles < cus) {
        if (target_env > 0) {
            S = target_env / tiles;
        } else {
            // time ~ rounds * k-steps per workgroup * c1 + slab traffic; c1 = us per k of one 128 x bn tile on one CU
            const double c1 = (bn == 256 ? 256.0 : 128.0) / 2400.0 * (bk == 16 ? 2 : 1), c2 = 8.0 / 4.0e6;  // write + read of a float at ~4 TB/s
            double best = 1e30;
            for (long s = 1; s <= 64 && s * 128 <= g->K; ++s) {
                const long chunk = as_round_up(as_cdiv(g->K, s), KALIGN);
                const long rounds = (tiles * s + cus - 1) / cus;
                const double cost = rounds * chunk * c1 + (s > 1 ? s * (per + per_cs) * c2 : 0.0);
                if (cost < best - 1e-9) best = cost, S = s;
            }
        }
        if (S > g->K / 128) S = g->K / 128;
        if (S * (per + per_cs) > g->splitk_ws_floats) S = g->splitk_ws_floats / (per + per_cs);
        if (S < 1) S = 1;
    }
    // Many tiles that do not fill whole rounds of the CUs: stream-K (see WgradMulti) instead of whole tiles per workgroup
    static const bool no_sk = AS_DIAG_SET("AS_WGRAD_NO_STREAMK");
    const long rounds = (tiles + cus - 1) / cus;
    if (!no_sk && bk == 32 && g->splitk_ws && tiles * 2 >= cus && tiles * 100 < rounds * cus * 95 && g->K / 32 >= 16 &&
        (long)cus * 2 * PIECE_FLOATS <= g->splitk_ws_floats) {
        k.kchunk = g->K; k.splitk = 1; k.tile0 = 0;
        k.ncombos = (long)g->batch * k.tiles_n;
        mm.n = 1;
        mm.streamk = 1;
        mm.nkt = g->K / 32;
        mm.total_tiles = tiles;
        mm.total_units = tiles * mm.nkt;
        mm.unit_per_wg = (mm.total_units + cus - 1) / cus;
        mm.total_items = (mm.total_units + mm.unit_per_wg - 1) / mm.unit_per_wg;   // workgroups
        mm.per_xcd = (int)((mm.total_items + 7) / 8);
        mm.pieces = g->splitk_ws;
        const int rc = bn == 256 ? launch_multi<256>(mm, bk, st) : launch_multi<128>(mm, bk, st);
        return rc == 0 ? 1 : rc;
    }
    k.kchunk = (int)as_round_up(as_cdiv(g->K, S), KALIGN);
    k.splitk = as_cdiv(g->K, k.kchunk);
    if (k.splitk > 1) {
        k.slab = g->splitk_ws;
        k.cs_slab = g->splitk_ws + (long)k.splitk * per;
    }
    k.ncombos = (long)g->batch * k.splitk * k.tiles_n;
    mm.n = 1;
    mm.total_items = k.ncombos * k.tiles_m;
    mm.per_xcd = (int)((mm.total_items + 7) / 8);
    mm.total_red = per / 4 + per_cs;
    const int rc = bn == 256 ? launch_multi<256>(mm, bk, st) : launch_multi<128>(mm, bk, st);
    return rc == 0 ? 1 : rc;
}

// Several weight-gradient problems of one reduction length as ONE launch + one reduce launch (gemm_internal.h).  Every
// problem runs on 128 x 256 tiles (N > 128 expected) and is split over K in chunks of the same length, chosen so that the
// grid is a few rounds of the CUs the caller expects (cu_budget, 0 = chip): many short workgroups instead of one long one
// per CU, so the dispatcher fills every free CU and other streams' kernels get CUs as workgroups retire.
// 1 = launched, 0 = not a case (nothing launched: the caller issues the problems one by one), < 0 = error.
int as_wgrad_multi(const as_wgrad_job* jobs, int n, float* slab, long slab_floats, int cu_budget, hipStream_t st, bool exact) {
    static const bool off = AS_DIAG_SET("AS_NO_WGRAD_MULTI");  // ablation: one launch per problem
    if (off || n < 1 || n > MAXP || !slab) return 0;
    WgradMulti mm{};
    long tiles = 0;
    const int K = jobs[0].g.K;
    for (int i = 0; i < n; ++i) {
        const as_gemm* g = &jobs[i].g;
        WgradK& k = mm.p[i];
        if (g->K != K || g->a_off || g->b_off || g->c_off || !describe(g, k)) return 0;
        if (jobs[i].colsum_b && g->b_kT > 0) return 0;
        k.tiles_m = as_cdiv(g->M, BM);
        k.tiles_n = as_cdiv(g->N, 256);
        k.colsum_b = jobs[i].colsum_b; k.colsum_b_batch = jobs[i].colsum_b_batch; k.c_trans = jobs[i].c_trans;
        tiles += (long)k.tiles_m * k.tiles_n * g->batch;
    }
    // chunk length: a multiple of 32 frames, at least 256; cost = rounds of `cus` workgroups x (chunk + fixed cost per
    // workgroup) + slab traffic, all in units of one k-tile of 32 frames (~4 us for a 128 x 256 tile)
    const int cus = cu_budget > 0 ? cu_budget : 256;
    static const int chunk_env = AS_DIAG_INT("AS_WGRAD_MULTI_CHUNK", 0);   // k-tiles per workgroup (tuning aid)
    const int nkt = K / 32;
    int best_chunk = nkt;
    double best = 1e30;
    for (int chunk = 8; chunk <= nkt; ++chunk) {
        const long S = as_cdiv(nkt, chunk);
        const long W = tiles * S;
        const double rounds = (double)((W + cus - 1) / cus);
        const double cost = rounds * (chunk + 1.0) + 0.016 * W;   // 128 KB of slab written + read per workgroup at ~4 TB/s
        if (cost < best - 1e-9) best = cost, best_chunk = chunk;
    }
    if (chunk_env > 0) best_chunk = chunk_env < nkt ? chunk_env : nkt;
    long off_f = 0, item0 = 0, red0 = 0;
    for (int i = 0; i < n; ++i) {
        WgradK& k = mm.p[i];
        k.kchunk = best_chunk * 32;
        k.splitk = as_cdiv(K, k.kchunk);
        const long per = (long)k.batch * k.M * k.N;
        const long per_cs = k.colsum ? (long)k.batch * k.M : 0, per_csb = k.colsum_b ? (long)k.batch * k.N : 0;
        if (k.splitk > 1) {
            k.slab = slab + off_f;
            k.cs_slab = k.slab + (long)k.splitk * per;
            k.csb_slab = k.cs_slab + (long)k.splitk * per_cs;
            off_f += as_round_up((long)k.splitk * (per + per_cs + per_csb), 64);
        }
        k.ncombos = (long)k.batch * k.splitk * k.tiles_n;
        k.item0 = item0;
        k.red0 = red0;
        item0 += k.ncombos * k.tiles_m;
        red0 += per / 4 + per_cs + per_csb;
    }
    if (off_f > slab_floats) return 0;
    mm.n = n;
    mm.total_items = item0;
    mm.total_red = red0;
    mm.per_xcd = (int)((item0 + 7) / 8);
    const int rc = launch_multi<256>(mm, 32, st, exact);
    return rc == 0 ? 1 : rc;
}
