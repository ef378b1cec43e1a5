// Linear layers of the ArticulatorPredictor heads (encoder_decoder/models.py:10-33) with their LayerNorms fused into the
// GEMM that produces (forward) or consumes (backward) the normalised activations:
//
//   forward   x_hat_next = LN(relu(x_hat . W'^T + b'))        C = A[M][K] . B[N][K]^T   (B reduction-contiguous, "NT")
//   backward  dz = relu' * LNbwd(dz_next . W')                 C = A[M][K] . B[K][N]     (B output-contiguous, "NN")
//
// (W', b' = the weights with the LayerNorm affine folded in; LN = affine-free normalisation over the 256 features of one
// head.)  Unfused, every such layer moves its [rows][A][256] activation through HBM three to five times (GEMM store,
// normalize load + store, and in the backward x_hat besides): 216-288 MB per layer at B*T = 6400 against 72-144 MB here.
//
// One workgroup owns BM = 64 frames x the 256 features of ONE head, so a row's LayerNorm statistics never leave the
// workgroup.  8 waves side by side along the features (32 columns x all 64 rows each: two accumulators).  60 KB of LDS and
// < 128 VGPRs: TWO workgroups per CU, so that one's prologue (DMA latency), barriers and epilogue (a pure memory phase:
// 64 KB of x_hat out, in the backward 64 KB in as well) run under the other's MFMAs.  (One 96/128-row workgroup per CU with
// a deeper ring was measured first: every CU reached its epilogue at the same moment, 35-43 us of a 130 us launch were
// epilogue with the matrix pipes idle.)
// Operands stream through a 3-slot LDS ring of 16-deep k-tiles filled by LDS-DMA (global_load_lds_dwordx4, two k-tiles in
// flight, counted vmcnt + raw barriers) like wgrad_f32.hip.  Reduction-contiguous operands keep their global layout in LDS
// ([row][16 k], no padding possible under DMA) with the four 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3 on the
// SOURCE address: a lane then takes four consecutive k of its row with ONE conflict-free ds_read_b128, which feed four
// MFMA k-steps (k-step j of an 8-group pairs k = j in lanes 0-31 with k = j + 4 in lanes 32-63, for both operands alike).
// Epilogue, 32 rows at a time: the accumulators go to LDS (the ring's memory), then one wave per row does exactly what
// normalize_fwd_kernel / normalize_bwd_kernel (rowops.hip) do, reading the GEMM result from LDS instead of HBM.
#include <cstdlib>

#include <type_traits>

#include "gemm_internal.h"
#include "split_arith.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BN = 256, BK = 16, NT = 512, NBUF = 3;
enum { EPI_PLAIN = 0, EPI_LNF = 1, EPI_LNB = 2 };

struct LinK {
    const float* A; long lda, a_batch;
    const float* B; long ldb, b_batch;
    const uint16_t* Bp; long bp_plane, bp_batch; int bp_rows;   // lin_s6_kernel: B as three bfloat16 planes (as_lin.Bp)
    int kchunk; long c_split;           // lin_s6_plain_kernel: blockIdx.y = k-chunk of kchunk, its partial result at C + y * c_split
    float* C; long ldc, c_batch;
    const float* bias; long bias_batch;
    int M, N, K, ka_valid, batch, act;
    int n_big, big_per_batch, big_per_batch_rows, small_per_batch;   // tiles of 64 rows first, then tiles of 32 rows (see launch())
    int tile_rows;
#ifdef AS_DIAG
    int stagger;
#else
    static constexpr int stagger = 0;
#endif
    unsigned long long* dbg; long dbg_max;   // diagnostic cycle stamps (as_lin_debug_stamps), normally null
#ifdef AS_DIAG
    int abl;  // diagnostic (AS_LIN_ABL): 1 = return before the epilogue, 2 = no DMA after the prologue
#else
    static constexpr int abl = 0;
#endif
    float eps;
    float* rstd;                      // EPI_LNF out: [M][batch]
    unsigned long long* bits;         // EPI_LNF out: [M][batch][4]: v > 0 per feature
    const float* xhat; long ldx, x_batch;   // EPI_LNB in: the normalised activations this layer's LayerNorm produced
    const float* rstd_in;             // EPI_LNB in: [M][batch]
    const unsigned long long* bits_in;
};

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at the wave-uniform byte
// address `lds_dst` (+ lane * 16).  Inline asm on purpose: with the builtin hipcc knows that the instruction writes LDS and
// puts an s_waitcnt vmcnt(0) in front of the next ds_read -- every k-tile then waits for the DMAs it has just issued (seen
// in this kernel's ISA; the ring's counted waits + barriers below are what orders a slot's reads behind its DMAs).
// M0 is compiler-reserved: saved and restored inside the one statement that uses it.
__device__ __forceinline__ void glds16(const float* src, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);  // derived from the wave index: uniform, but only the hardware knows
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {  // byte address inside the workgroup's LDS, wave-uniform
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p);
}

// Sum over the 64 lanes, result in every lane.  Four DPP steps give every lane its 16-lane row total (xor 1, xor 2, mirror of
// 8, mirror of 16: ~8 cycles each); the four row totals are read out with v_readlane and added as wave-uniform values.
// (as_wave_sum's six __shfl_xor steps are six dependent LDS-crossbar round trips: 12 of them per LayerNorm row made the
// epilogue longer than the GEMM's main loop.)
__device__ __forceinline__ float wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ void lds_barrier() {  // LDS hazards only: unlike __syncthreads() it leaves global stores in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// Sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane of the row: four DPP steps, no
// cross-row traffic, no v_readlane.
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// ---- the LayerNorm epilogues.  The accumulators of a block of 32 rows (per group of 8 column waves) go to LDS, then every
// wave normalises FOUR rows at a time: lane = (row rr = lane >> 4, slot jj = lane & 15) holds the 16 features
// 64 c + 4 jj + e (c, e < 4) of its row -- four ds_read_b128 that are conflict-free (the 16 lanes one LDS cycle serves read
// 256 consecutive bytes of two rows), row sums by four DPP steps inside the 16-lane row, stores as four
// global_store_dwordx4 of 256 consecutive bytes per row.  (Before: one wave per row, 64 lanes x 4 features, two 64-lane sums
// with v_readlane each and an IEEE 1 / sqrt per row, the four rows of a wave one after the other: 12 k cycles per 64-row
// tile, as long as the split-arithmetic main loop.)
//   NWC  column waves per row block (8).  Waves of row group wm = wave / NWC own tile rows wm * 64 + i * 32 + ...; the
//        staging area holds one 32-row block per row group: [groups * 32][BN] floats.  groups (1 | 2) = row groups that hold
//        rows of the tile; tm_eff = their row blocks (a workgroup may have more waves than that: they only keep the barriers).
//   EPI_LNF: x = relu(acc + bias); C = (x - mean) * rstd; rstd; bits (x > 0, one 32-bit piece per (row, column wave), taken
//            from the accumulators by ballot before they leave for LDS).
//   EPI_LNB: C = relu'(bits_in) * rstd_in * (acc - mean(acc) - xhat * mean(acc * xhat)).
template <int TM, int EPI, int NWC>
__device__ __forceinline__ void lin_ln_rows(const LinK& g, f32x16 (&acc)[TM], float* smem, int bz, int m0, int tm_eff, int groups, int wave,
                                            int lane, float bj) {
    const int l31 = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wn = wave_u % NWC, wm = wave_u / NWC;       // column wave, row group
    const int col = wn * 32 + l31;
    const int rr = lane >> 4, jj = lane & 15;
    constexpr float inv_d = 1.0f / BN;
    // staged row of this lane in the normalisation phase: 4 * wave + rr of the NG * 32 staged rows; its tile row
    const int srow = 4 * wave_u + rr;
    const int trow_base = (srow >> 5) * 64 + (srow & 31);      // + i * 32
    const bool stages = wm < groups, normalises = 4 * wave_u < 32 * groups;   // (wave-uniform)
    // backward: everything this wave reads from global memory is requested up front (vector memory operations retire in
    // order: a load issued behind the first block's stores would wait for them)
    f32x4 h[TM][4];
    float rs_in[TM];
    u32x4 mw[TM][2];
    if (EPI == EPI_LNB && normalises) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long row = min((long)m0 + trow_base + i * 32, (long)g.M - 1);
            const float* xr = g.xhat + (long)bz * g.x_batch + row * g.ldx + 4 * jj;
#pragma unroll
            for (int c = 0; c < 4; ++c) h[i][c] = *reinterpret_cast<const f32x4*>(xr + 64 * c);
            rs_in[i] = g.rstd_in[row * g.batch + bz];
            const u32x4* mp = reinterpret_cast<const u32x4*>(g.bits_in + (row * g.batch + bz) * 4);
            mw[i][0] = mp[0];
            mw[i][1] = mp[1];
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (i >= tm_eff) break;
        if (i > 0) lds_barrier();    // the previous block's rows are all read
        float* st = smem + wm * 32 * BN;
        if (!stages) {
        } else if (EPI == EPI_LNF) {
            unsigned mine = 0;        // lane L < 32: the mask piece (columns 32 wn .. 32 wn + 31) of block row L
            // (v_writelane_b32 with the lane as an immediate: one instruction per piece -- a compare + select per piece would be
            // four, and two scalar operands violate the constant-bus limit -- hence a macro over the literal register index)
#define AS_LN_STAGE(r)                                                                                                        \
            {                                                                                                                 \
                const float v = as_relu(acc[i][r] + bj);                                                                      \
                const unsigned long long b = __ballot(v > 0.f);                                                               \
                /* gfx950: a scalar register written by a vector compare needs two wait states before a vector instruction */  \
                /* reads it; hipcc's hazard pass does not look into inline assembly (seen: a stale mask piece, rarely)   */  \
                asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(mine) : "s"((unsigned)b), "i"(((r) & 3) + 8 * ((r) >> 2))); \
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mine) : "s"((unsigned)(b >> 32)), "i"(((r) & 3) + 8 * ((r) >> 2) + 4)); \
                st[(((r) & 3) + 8 * ((r) >> 2) + 4 * lh) * BN + col] = v;                                                     \
            }
            AS_LN_STAGE(0) AS_LN_STAGE(1) AS_LN_STAGE(2) AS_LN_STAGE(3) AS_LN_STAGE(4) AS_LN_STAGE(5) AS_LN_STAGE(6) AS_LN_STAGE(7)
            AS_LN_STAGE(8) AS_LN_STAGE(9) AS_LN_STAGE(10) AS_LN_STAGE(11) AS_LN_STAGE(12) AS_LN_STAGE(13) AS_LN_STAGE(14) AS_LN_STAGE(15)
#undef AS_LN_STAGE
            const long brow = (long)m0 + wm * 64 + i * 32 + lane;
            if (lane < 32 && brow < g.M) reinterpret_cast<unsigned*>(g.bits)[(brow * g.batch + bz) * 8 + wn] = mine;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * lh) * BN + col] = acc[i][r];
        }
        lds_barrier();
        if (!normalises) continue;
        const long row = (long)m0 + trow_base + i * 32;
        f32x4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const f32x4*>(smem + srow * BN + 64 * c + 4 * jj);
        float* o = g.C + (long)bz * g.c_batch + row * g.ldc + 4 * jj;
        if (EPI == EPI_LNF) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
            const float mean = row16_sum(s) * inv_d;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] -= mean;
                q += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
            }
            const float rs = 1.0f / sqrtf(row16_sum(q) * inv_d + g.eps);
            if (row < g.M) {
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(o + 64 * c) = v[c] * rs;
                if (jj == 0) g.rstd[row * g.batch + bz] = rs;
            }
        } else {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1 += (v[c].x + v[c].y) + (v[c].z + v[c].w);
                s2 += (v[c].x * h[i][c].x + v[c].y * h[i][c].y) + (v[c].z * h[i][c].z + v[c].w * h[i][c].w);
            }
            const float m1 = row16_sum(s1) * inv_d, m2 = row16_sum(s2) * inv_d;
            if (row < g.M) {
                const float rs = rs_in[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // mask word c (64 bits: features 64 c ..): this lane's four bits sit at 4 jj
                    const unsigned lo = c == 0 ? mw[i][0].x : c == 1 ? mw[i][0].z : c == 2 ? mw[i][1].x : mw[i][1].z;
                    const unsigned hi = c == 0 ? mw[i][0].y : c == 1 ? mw[i][0].w : c == 2 ? mw[i][1].y : mw[i][1].w;
                    const unsigned nib = (jj < 8 ? lo : hi) >> (4 * (jj & 7));
                    f32x4 d = (v[c] - m1 - h[i][c] * m2) * rs;
                    d.x = (nib & 1u) ? d.x : 0.f;
                    d.y = (nib & 2u) ? d.y : 0.f;
                    d.z = (nib & 4u) ? d.z : 0.f;
                    d.w = (nib & 8u) ? d.w : 0.f;
                    *reinterpret_cast<f32x4*>(o + 64 * c) = d;
                }
            }
        }
    }
}

// ---- epilogue of the 256-column kernels (8 waves, wave w owns columns 32 w .. 32 w + 31 of all BM rows).
// D[row][col]: col = wave * 32 + l31, row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh -- the accumulator layout of
// v_mfma_f32_32x32x2_f32 and of v_mfma_f32_32x32x16_bf16 alike.  `smem`: >= 32 * BN floats, free of readers (behind a barrier).
template <int TM, int EPI>
__device__ __forceinline__ void lin_epilogue(const LinK& g, f32x16 (&acc)[TM], float* smem, int bz, int m0, int tm_eff, int wave, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
    const int col = wave * 32 + l31;
    if (EPI == EPI_PLAIN) {
        if (col < g.N) {
            const float bj = g.bias ? g.bias[(long)bz * g.bias_batch + col] : 0.f;
            float* c0 = g.C + (long)bz * g.c_batch + col;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (i < tm_eff && row < g.M) {   // (a 32-row tile owns only its first row block)
                        float v = acc[i][r] + bj;
                        if (g.act == 1) v = as_relu(v);
                        else if (g.act == 2) v = as_sigmoid(v);
                        c0[(long)row * g.ldc] = v;
                    }
                }
        }
        return;
    }
    const float bj = (EPI == EPI_LNF && g.bias) ? g.bias[(long)bz * g.bias_batch + col] : 0.f;
    lin_ln_rows<TM, EPI, 8>(g, acc, smem, bz, m0, tm_eff, 1, wave, lane, bj);
}

// NB ring slots: 3 (two k-tiles in flight, 60 KB: two workgroups per CU) or 2 (one in flight, 40 KB: three per CU; diagnostic)
template <int BM, bool B_KC, int EPI, int NB = NBUF>
__global__ __launch_bounds__(NT, NB == 2 ? 6 : 4) void lin_f32_kernel(LinK g) {
    constexpr int TM = BM / 32;
    constexpr int TILE = BK * (BM + BN);
    constexpr int PA_TOTAL = BM / 16;    // 1-KiB DMA pieces of the A tile (16 rows x 16 k each)
    constexpr int RING = NB * TILE, EPIT = 32 * BN;
    constexpr int AHEADT = NB - 1;       // k-tiles in flight besides the one being multiplied
    __shared__ __attribute__((aligned(16))) float smem[RING > EPIT ? RING : EPIT];

    // tile list: the 64-row tiles of every head first, then 32-row tiles over the remaining rows of every head
    int bz, m0, tm_eff;
    if ((int)blockIdx.x < g.n_big) {
        bz = blockIdx.x / g.big_per_batch;
        m0 = (blockIdx.x - bz * g.big_per_batch) * BM;
        tm_eff = TM;
    } else {
        const int j = blockIdx.x - g.n_big;
        bz = j / g.small_per_batch;
        m0 = g.big_per_batch_rows + (j - bz * g.small_per_batch) * 32;
        tm_eff = 1;
    }
    const float* __restrict__ A = g.A + (long)bz * g.a_batch;
    const float* __restrict__ B = g.B + (long)bz * g.b_batch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- DMA sources (per lane) and destinations (per wave)
    // A: piece p = rows 16p .. 16p+15 of the tile; lane -> row 16p + (lane >> 2), LDS chunk slot lane & 3, which receives the
    // global chunk slot ^ ((row >> 2) & 3).  Every wave issues ONE A piece: waves 4-7 re-issue pieces 0-3 (identical bytes to
    // identical addresses) so that all waves count the same number of DMAs per k-tile.
    // Chunks at or beyond ka_valid (a reduction padded up to a multiple of 16 whose B rows there are zero) re-read chunk 0.
    const int pa = wave % PA_TOTAL;
    const int ra = pa * 16 + (lane >> 2);
    const int a_gc = ((lane & 3) ^ ((ra >> 2) & 3)) * 4;
    const float* a_src = A + (long)min(m0 + ra, g.M - 1) * g.lda + a_gc;
    // B: 16 pieces, two per wave.  B_KC: piece = 16 rows (output features) x 16 k, swizzled like A.  Else: piece = one k row
    // of 256 output features.
    const float* b_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wave * 2 + j;
        if (B_KC) {
            const int r = p * 16 + (lane >> 2);
            const int gc = (lane & 3) ^ ((r >> 2) & 3);
            b_src[j] = B + (long)min(r, g.N - 1) * g.ldb + gc * 4;
        } else {
            b_src[j] = B + (long)p * g.ldb + min(lane * 4, g.N - 4);
        }
    }
    const unsigned smem_base = lds_addr(smem);
    // piece 0 = this wave's A piece, pieces 1, 2 = its two B pieces of k-tile kt
    auto issue_piece = [&](int kt, int piece) {
        const unsigned base = smem_base + (unsigned)((kt % NB) * TILE) * 4u;
        const int k0 = kt * BK;
        if (piece == 0) {
            const bool ok = k0 + a_gc + 4 <= g.ka_valid;
            glds16(a_src + (ok ? k0 : -a_gc), base + (unsigned)(pa * 1024));
        } else {
            const int j = piece - 1;
            glds16(b_src[j] + (B_KC ? (long)k0 : (long)k0 * g.ldb), base + (unsigned)((BK * BM + (wave * 2 + j) * 256) * 4));
        }
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) issue_piece(kt, pc);
    };

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int nk = g.K / BK;
    const bool stamp = g.dbg != nullptr && tid == 0 && blockIdx.x < g.dbg_max;
    if (stamp) {
        unsigned long long* d = g.dbg + 8L * blockIdx.x;
        d[0] = __builtin_amdgcn_s_memtime();
        d[4] = __builtin_amdgcn_s_memrealtime();
        d[6] = 0;
        d[5] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20) << 32);
    }
    // Two workgroups share a CU and would run in lockstep (same start, same tile time): both in their prologue, both in
    // their main loop, both in their epilogue -- nothing overlaps.  The second wave of the first fill (blocks 256..511 under
    // round-robin dispatch: speed only) starts half a tile late; every later workgroup inherits the phase of the slot it takes.
    if (g.stagger > 0 && g.stagger < 10 && blockIdx.x >= 256 && blockIdx.x < 256 * NB) {
        // NB = 3 (two per CU): the second starts half a tile late; NB = 2 (three per CU): a third and two thirds of a tile
        const int steps = NB == 3 ? nk * g.stagger : (int)(blockIdx.x >> 8) * nk * g.stagger * 2 / 3;
        for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(16);  // 16 * 64 cycles = half a k-tile of MFMAs
    }
    // diagnostic (AS_LIN_STAGGER >= 10): the same delay, but for the workgroup that really is the SECOND tenant of its CU --
    // told by the wave slot its first wave got on its SIMD (HW_ID.wave_id >= 2: slots 0, 1 belong to the first tenant) -- instead
    // of by block index (the dispatcher need not place blocks b and b + 256 on one CU)
    if (g.stagger >= 10 && blockIdx.x < 512) {
        const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID[3:0] = wave slot on the SIMD
        __shared__ int second;
        if (tid == 0) second = slot >= 2;
        __syncthreads();
        if (second)
            for (int i = 0; i < nk * (g.stagger - 9); ++i) __builtin_amdgcn_s_sleep(16);
    }
    // diagnostic (AS_LIN_STAGGER < 0): de-phase the XCDs instead -- the first-fill workgroups of XCD x start x * |stagger| / 8
    // of a 16-k-tile main loop late, so that the epilogues' store bursts of the eight dies do not coincide chip-wide
    if (g.stagger < 0 && blockIdx.x < 512) {
        const int steps = (int)(blockIdx.x & 7) * (-g.stagger) * 2;          // x * |stagger| * 2 sleeps of 1024 cycles
        for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const int swz = (l31 >> 2) & 3;   // rows i * 32 + l31 and wave * 32 + l31 share it (32 = 0 mod 16)
    // two k-tiles in flight; the older one is retired with vmcnt(3).  (A 2-slot ring with three workgroups per CU was no
    // faster: 108-118 us against 108-114 for head GEMM 2.)
    issue(0);
    if (AHEADT > 1 && nk > 1) issue(1);
    if (AHEADT > 1 && nk > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (stamp) g.dbg[8L * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
    // Main loop, software-pipelined across the k-tile boundary.  A k-tile is two 8-deep fragment groups; the barrier that
    // ends a tile sits BETWEEN the two groups' MFMAs: when a wave reaches it, all its LDS reads of the tile are in registers
    // and eight MFMAs are still to be issued, and right behind it the first fragments of the NEXT tile are requested, so
    // the barrier's skew and the LDS round trip run under matrix work instead of in front of it (before: every tile began
    // with all eight waves waiting for their first fragments, ~10 % of a 2048-cycle tile with the pipe idle).
    struct Frag { float4 av[TM]; float bv[4]; };
    auto load = [&](Frag& f, int kt_, int cc) {
        const float* tile = smem + (kt_ % NB) * TILE;
        const float* a_s = tile + l31 * BK;
        const float* b_s = tile + BK * BM;
        const int slot = ((2 * cc + lh) ^ swz) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.av[i] = *reinterpret_cast<const float4*>(a_s + i * 32 * BK + slot);
        if (B_KC) {
            const float4 t = *reinterpret_cast<const float4*>(b_s + (wave * 32 + l31) * BK + slot);
            f.bv[0] = t.x; f.bv[1] = t.y; f.bv[2] = t.z; f.bv[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) f.bv[j] = b_s[(cc * 8 + 4 * lh + j) * BN + wave * 32 + l31];
        }
    };
    // dma_kt >= 0: the three DMA pieces of k-tile dma_kt are issued BETWEEN the MFMAs (one after each of the first three
    // k-steps): an LDS-DMA costs the issuing wave ~60-180 cycles of instruction issue, and as a block at the top of the tile
    // -- every wave of the workgroup at the same moment, right behind the barrier -- those cycles were matrix-pipe idle
    // time; behind an MFMA they run while the pipe works on it.
    auto mma = [&](const Frag& f, int dma_kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (i < tm_eff) {   // (a 32-row tile owns only its first row block)
                    const float a = j == 0 ? f.av[i].x : j == 1 ? f.av[i].y : j == 2 ? f.av[i].z : f.av[i].w;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f.bv[j], acc[i], 0, 0, 0);
                }
            }
            if (j < 3 && dma_kt >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(dma_kt, j);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    static_assert(BK == 16, "two fragment groups per k-tile");
    Frag f0, f1;
    load(f0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        load(f1, kt, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0, kt + AHEADT < nk ? kt + AHEADT : -1);
        __builtin_amdgcn_sched_barrier(0);
        if (AHEADT > 1 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of slot kt % NB are in registers
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) load(f0, kt + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(f1, -1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (stamp) g.dbg[8L * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    if (g.abl == 1) {
        if (acc[0][0] == 123.456f) g.C[0] = acc[0][0];  // keep the loop alive
        return;
    }

    lin_epilogue<TM, EPI>(g, acc, smem, bz, m0, tm_eff, wave, lane);
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g.dbg[8L * blockIdx.x + 3] = __builtin_amdgcn_s_memtime();
        g.dbg[8L * blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layers with the products on the bfloat16 matrix instruction (v_mfma_f32_32x32x16_bf16: 16 x the rate of
// v_mfma_f32_32x32x2_f32), fp32 in, fp32 out, fp32 accumulation.  An fp32 number is EXACTLY the sum of three bfloat16
// numbers (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 8 + 8 + 8 significand bits), so a product a.b is the
// sum of nine plane products with exact operands; the three smallest (lo.lo, lo.mid, mid.lo, each <= 2^-24 |a||b|) are
// dropped: six MFMAs per 16-deep k-step.  Against an fp64 product of the same fp32 operands this is MORE accurate than the
// fp32 matrix instruction (rms 2.4e-7 vs 2.9e-7 of rms C: the bf16 instruction adds 16 products before it rounds, the fp32
// one two; tools/bench_split_gemm.py, tests/test_gpu_parity.py::test_split_matrix_arithmetic_*).
//   * B (the folded weights) arrives as planes, emitted once per step by as_emit_planes (rowops.hip) k-step-major:
//     [plane][head][K / 16][rows][16] bf16, so that the fragment of a wave (32 columns x 8 k x 2 lane halves) is 1 KiB of
//     consecutive bytes.  A wave's columns are its own: B fragments go from L2 straight to registers (global_load_dwordx4,
//     two k-steps ahead in three name-rotated register sets), not through the LDS.
//   * A (activations / gradients) stays fp32 in HBM.  Each thread loads 4 consecutive k of one row per 32-deep k-tile (two
//     tiles ahead), splits them (v_cvt_pk_bf16_f32 + v_pk_add_f32: 4.5 vector instructions per element, once per workgroup)
//     and writes 3 x 8 bytes into the plane image of the tile in LDS ([plane][64 rows][32 k] bf16 = 64-byte rows, the four
//     16-byte chunks of a row XOR-swizzled by (row >> 2) & 3: the 16 lanes a ds_read_b128 serves per LDS cycle -- rows
//     {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} -- then touch each of the 64 banks once).  All eight
//     waves read their A fragments from there: one barrier per 32-deep k-tile, two plane images (24 KB).
//   * accumulator layout = that of the fp32 instruction: the LayerNorm epilogues above are shared.
// a pointer the compiler must treat as wave-uniform (SGPR pair): base of the scalar-base form of a global load, whose lane
// part is then a 32-bit byte offset.  (Without it hipcc re-associates base + lane offset into a loop-invariant 64-bit VECTOR
// address and adds the uniform per-step part to that: two address registers and a 64-bit vector add per load.)
typedef const __attribute__((address_space(1))) char* gptr;   // global address space (the integer round trip would lose it: flat loads)
__device__ __forceinline__ gptr uniform_ptr(const void* p) {
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<gptr>(((uintptr_t)hi << 32) | lo);
}
typedef const __attribute__((address_space(1))) f32x4* gptr_f4;
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;

// Workgroups go to the eight XCDs round-robin by block index, and each XCD has its own 4 MB L2.  With tiles numbered head
// by head, block b -> tile b puts every head's weights into every L2 (11 heads x 393 KB of planes do not fit one).  This map
// gives XCD x a CONTIGUOUS range of the tile list instead (two or three heads): block b = x + 8 i -> tile start(x) + i.
__device__ __forceinline__ int xcd_contiguous(int b, int total) {
    const int q = total >> 3, r = total & 7, x = b & 7, i = b >> 3;
    return x * q + min(x, r) + i;
}

constexpr int S6_BK = 32;                 // k-tile of the A image (two MFMA k-steps)
constexpr int S6_PLANE = 64 * S6_BK * 2;  // bytes of one plane of one k-tile: 64 rows x 32 bf16
constexpr int S6_BUF = 3 * S6_PLANE;

template <int N> struct IC { static constexpr int value = N; };

// The main loop shared by the 256-column kernel (NW = 8 waves) and the output layer's (NW = 4): 64 rows x 32 NW columns,
// wave w = columns 32 w .. 32 w + 31 of all 64 rows (two accumulators).  `sm`: 2 x S6_BUF bytes.
//   A / lda: first row of the tile's operand rows (row index clamped to rows_valid - 1), k >= ka_valid reads as zero
//   bp (wave-uniform) + b_lane bytes: this lane's B fragment of plane 0, k-step 0; plane stride bp_plane, k-step stride bp_step (elements)
//   (rows of A within 2^31 bytes of the tile's first: lda < 2^23 floats)
template <int NW, int TME>   // TME: row blocks of 32 the tile really has (a 32-row tile multiplies only the first)
__device__ __forceinline__ void s6_main_loop(f32x16 (&acc)[2], unsigned char* sm, const float* __restrict__ A, long lda, int rows_valid,
                                             int K, int ka_valid, const uint16_t* __restrict__ bp, unsigned b_lane, long bp_plane,
                                             long bp_step, int tid, int lane, bool late, unsigned long long* stamp1 = nullptr) {
    // compile-time ablations of a diagnostic build (python -m artspeech_amd.build --diag -DAS_S6_ABL=n): 1 = no epilogue,
    // 2 = no B loads inside the loop, 3 = no A loads / splits / LDS stores inside the loop, 4 = no MFMAs, 5 = every tile reads
    // the fragments of plane image 0.  (A run-time switch here costs the loop its register arrays: 16 x slower.)
#ifdef AS_S6_ABL
    constexpr int abl = AS_S6_ABL;
#else
    constexpr int abl = 0;
#endif
    constexpr int NTH = NW * 64;
    constexpr int AL = 512 / NTH;         // float4 loads per thread and k-tile (64 rows x 8 chunks of 4 k)
    const int l31 = lane & 31, lh = lane >> 5;
    const int a_chunk = tid & 7;
    // every global access = wave-uniform base pointer + 32-bit lane offset (the scalar-base form of global_load: no 64-bit
    // vector address arithmetic, no address register pairs)
    unsigned a_off[AL];
    int a_wr[AL];
#pragma unroll
    for (int q = 0; q < AL; ++q) {
        const int row = (tid >> 3) + q * (NTH / 8);
        a_off[q] = (unsigned)(min(row, rows_valid - 1) * (int)lda + a_chunk * 4) * 4u;   // bytes
        a_wr[q] = row * 64 + (((a_chunk >> 1) ^ ((row >> 2) & 3)) * 16) + (a_chunk & 1) * 8;
    }
    const int nk = K / S6_BK, nj = 2 * nk;
    const gptr Au = uniform_ptr(A);
    // every load below is unconditional (indices clamped to the last tile / k-step): straight-line code, so that hipcc's
    // s_waitcnt vmcnt counts are exact -- a load under a branch makes every later wait assume the branch was not taken,
    // i.e. wait for (nearly) everything in flight
    auto a_load = [&](f32x4 (&dst)[AL], int kt) {
        kt = min(kt, nk - 1);
        // chunks at or beyond ka_valid (a reduction padded up to the tile whose B rows there are zero planes) re-read chunk 0
        // of the row: an ADDRESS select -- a value select would put the load itself under a branch
        // (the offset stays non-negative: it is zero-extended by the scalar-base form)
        const unsigned ko = kt * S6_BK + a_chunk * 4 + 4 <= ka_valid ? (unsigned)(kt * S6_BK) * 4u : 0u - (unsigned)a_chunk * 16u;
#pragma unroll
        for (int q = 0; q < AL; ++q) dst[q] = *reinterpret_cast<gptr_f4>(Au + (a_off[q] + ko));
    };
    auto a_store = [&](const f32x4 (&src)[AL], int buf) {
#pragma unroll
        for (int q = 0; q < AL; ++q) {
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(src[q].x, src[q].y, h0, m0, l0);
            split_pair(src[q].z, src[q].w, h1, m1, l1);
            unsigned char* d = sm + buf * S6_BUF + a_wr[q];
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + S6_PLANE) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(d + 2 * S6_PLANE) = make_uint2(l0, l1);
        }
    };
    auto b_load = [&](u32x4 (&dst)[3], int j) {
        j = min(j, nj - 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<gptr_u4>(uniform_ptr(bp + p * bp_plane + j * bp_step) + b_lane);
    };
    const int sw = (l31 >> 2) & 3;        // rows i * 32 + l31 share it (32 = 0 mod 16)
    const unsigned char* a_rd = sm + l31 * 64;

    f32x4 aq[3][AL];
    u32x4 bq[3][3];
    a_load(aq[0], 0);
    a_load(aq[1], 1);
    b_load(bq[0], 0);
    b_load(bq[1], 1);
    a_store(aq[0], 0);
    lds_barrier();
    if (stamp1) *stamp1 = __builtin_amdgcn_s_memtime();
    // k-tile kt (kt % 3 == U): its plane image is in buffer kt & 1, its B fragments in sets (2 kt) % 3 and (2 kt + 1) % 3
#ifdef AS_S6_TRACE   // diagnostic build: per-tile cycle stamps of every wave of the stamped workgroups (tools/s6_trace.py)
    unsigned long long* tr = stamp1 ? stamp1 - 1 + 8L * 4096 + ((long)blockIdx.x * 8 + (tid >> 6)) * 64 : nullptr;
#define AS_TR(slot) if (tr && lane == 0) tr[(kt < 10 ? kt : 9) * 6 + (slot)] = __builtin_amdgcn_s_memtime();
#else
#define AS_TR(slot)
#endif
    auto tile = [&](auto Uc, int kt) {
        constexpr int U = decltype(Uc)::value;
        const unsigned char* img = a_rd + (abl == 5 ? 0 : (kt & 1) * S6_BUF);
        AS_TR(0)
        if (abl != 3) a_load(aq[(U + 2) % 3], kt + 2);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (abl != 2) b_load(bq[(2 * U + s + 2) % 3], 2 * kt + s + 2);
            __builtin_amdgcn_sched_barrier(0);   // the look-ahead loads stay HERE, two k-steps in front of their first use
            bf16x8 fa[TME][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < TME; ++i)
                    fa[i][p] = *reinterpret_cast<const bf16x8*>(img + p * S6_PLANE + i * 32 * 64 + (((2 * s + lh) ^ sw) * 16));
            const u32x4(&bs)[3] = bq[(2 * U + s) % 3];
            if (s == 0) { AS_TR(1) }
            bf16x8 fb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[p] = __builtin_bit_cast(bf16x8, bs[p]);
            // six of the nine plane products (without mid.lo, lo.mid, lo.lo).  Those with the hi plane of A first: the first six
            // MFMAs need two of the six LDS reads, the mid / lo fragments arrive under them.  (The order does not matter to the
            // result's error: the accumulator already holds the sum of the earlier k-steps.)
            constexpr int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int o = 0; o < 6; ++o)
#pragma unroll
                for (int i = 0; i < TME; ++i)
                    if (abl != 4) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[o]], fb[PB[o]], acc[i], 0, 0, 0);
                    else acc[i][o] += (float)fa[i][PA[o]][0] * (float)fb[PB[o]][0];
            __builtin_amdgcn_sched_barrier(0);
            // The split of the NEXT tile (vector + LDS-store work, no matrix work) sits at a different place in the two halves of
            // the workgroup: waves 0-3 behind the tile's second k-step, waves 4-7 between the two.  Waves w and w + 4 share a
            // SIMD and, running the same program between the same barriers, would otherwise do their vector work at the same
            // moment and their matrix work at the same moment -- the matrix pipe idles through the former.
            if (s == 0) { AS_TR(2) } else { AS_TR(3) }
            if (abl != 3 && late == (s == 0)) a_store(aq[(U + 1) % 3], (kt & 1) ^ 1);   // (behind the last tile: a clamped repeat)
        }
        AS_TR(4)
        lds_barrier();
        AS_TR(5)
    };
#undef AS_TR
    int kt = 0;
    for (; kt + 3 <= nk; kt += 3) {
        tile(IC<0>{}, kt);
        tile(IC<1>{}, kt + 1);
        tile(IC<2>{}, kt + 2);
    }
    if (kt < nk) tile(IC<0>{}, kt);
    if (kt + 1 < nk) tile(IC<1>{}, kt + 1);
}

template <int EPI>
__global__ __launch_bounds__(NT, 4) void lin_s6_kernel(LinK g) {
    constexpr int EPIT = 32 * BN;
    static_assert(EPIT * 4 >= 2 * S6_BUF, "the epilogue's staging area holds both plane images");
    __shared__ __attribute__((aligned(16))) float smem[EPIT];
    int bz, m0, tm_eff;
    const int tile = blockIdx.x;   // (an XCD-contiguous map of the tile list, xcd_contiguous(), was measured: 5-10 % slower)
    if (tile < g.n_big) {
        bz = tile / g.big_per_batch;
        m0 = (tile - bz * g.big_per_batch) * 64;
        tm_eff = 2;
    } else {
        const int j = tile - g.n_big;
        bz = j / g.small_per_batch;
        m0 = g.big_per_batch_rows + (j - bz * g.small_per_batch) * 32;
        tm_eff = 1;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const uint16_t* bp = g.Bp + (long)bz * g.bp_batch;
    const unsigned b_lane = (unsigned)((wave * 32 + (lane & 31)) * 16 + (lane >> 5) * 8) * 2u;   // bytes
    unsigned char* sm = reinterpret_cast<unsigned char*>(smem);
    const float* A = g.A + (long)bz * g.a_batch + (long)m0 * g.lda;
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
    const bool stamp = g.dbg != nullptr && tid == 0 && blockIdx.x < g.dbg_max;   // diagnostic cycle stamps (as_lin_debug_stamps)
    unsigned long long* d = stamp ? g.dbg + 8L * blockIdx.x : nullptr;
    if (stamp) {
        d[0] = __builtin_amdgcn_s_memtime();
        d[4] = __builtin_amdgcn_s_memrealtime();
        d[5] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20) << 32);
    }
    // diagnostic (AS_LIN_STAGGER = n > 0): the workgroups of the second fill (blocks 256..511 under round-robin dispatch, the
    // second tenant of each CU) start n x 1024 cycles late, so that the two tenants' epilogues (and the chip's store bursts)
    // do not coincide; n < 0: the same for blocks whose CU-slot parity ... (see lin_f32_kernel)
    if (g.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    if (tm_eff == 2) s6_main_loop<8, 2>(acc, sm, A, g.lda, g.M - m0, g.K, g.ka_valid, bp, b_lane, g.bp_plane, (long)g.bp_rows * 16, tid, lane, late, stamp ? d + 1 : nullptr);
    else s6_main_loop<8, 1>(acc, sm, A, g.lda, g.M - m0, g.K, g.ka_valid, bp, b_lane, g.bp_plane, (long)g.bp_rows * 16, tid, lane, late, stamp ? d + 1 : nullptr);
    if (stamp) d[2] = __builtin_amdgcn_s_memtime();
#if defined(AS_S6_ABL) && AS_S6_ABL == 1
    if (acc[0][0] == 123.456f) g.C[0] = acc[0][0] + acc[1][0];  // keep the loop alive
    return;
#endif
    lin_epilogue<2, EPI>(g, acc, smem, bz, m0, tm_eff, wave, lane);
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        d[3] = __builtin_amdgcn_s_memtime();
        d[6] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The wide form of lin_s6_kernel: ONE workgroup of 16 waves per CU owns 128 rows x 256 columns (two row groups of 8 column
// waves; wave = 64 rows x 32 columns as before) and the weight planes go through the LDS too, staged once per workgroup.
// Why: the 64-row kernel's main loop is bound by what the CUs pull from L2, not by the matrix pipe -- every wave loads its
// own B fragments, 24 KB per 64 rows and 16-deep k-step, 432 MB per launch of head Linear 2 against 72 MB of activations;
// removing the matrix instructions from that loop changed its time by 4 % (profiles/r04_s6_ablation.log).  Here a k-tile's
// planes (48 KB) are fetched once per 128 rows: a quarter of the bytes per row.  LDS: two A images (2 x 24 KB) + two B
// images (2 x 48 KB) = 144 KB; one barrier per 32-deep k-tile; A two tiles ahead in registers, B one tile ahead (load at the
// top of a tile, ds_write_b128 at its end: one register set).  The epilogue stages 64 rows at a time ([64][256] floats in the
// images' memory), every wave normalising four rows (lin_ln_rows).
// Tile list: 128-row tiles first (whole rounds of one workgroup per CU), then 32-row tiles over the remaining rows (row group
// 0 multiplies one row block, row group 1 only helps with the loads): the partly filled last round costs a third of a tile.
constexpr int SW_NT = 1024;
constexpr int SW_AIMG = 3 * 128 * 64;      // bytes of one A image: 3 planes x 128 rows x 32 bf16
constexpr int SW_BIMG = 3 * 256 * 64;      // bytes of one B image: 3 planes x 256 rows x 32 bf16
constexpr int SW_LDS = 2 * SW_AIMG + 2 * SW_BIMG;

template <int TME>   // row blocks of 32 this wave multiplies per k-step: 2 (128-row tile), 1 (32-row tile, row group 0), 0 (loads only)
__device__ __forceinline__ void s6w_main_loop(f32x16 (&acc)[2], unsigned char* sm, const float* __restrict__ A, long lda, int rows_valid, int K,
                                              int ka_valid, const uint16_t* __restrict__ bp, long bp_plane, int bp_rows, int tid, int lane,
                                              int wm, int wn, bool late) {
    const int l31 = lane & 31, lh = lane >> 5;
    // A: thread -> row tid >> 3 (0..127), 16-byte chunk tid & 7 of the row's 32 k
    const int a_row = tid >> 3, a_chunk = tid & 7;
    const unsigned a_off = (unsigned)(min(a_row, rows_valid - 1) * (int)lda + a_chunk * 4) * 4u;   // bytes
    const int a_wr = a_row * 64 + (((a_chunk >> 1) ^ ((a_row >> 2) & 3)) * 16) + (a_chunk & 1) * 8;
    // B: thread -> image row tid >> 2 (0..255), chunk tid & 3 = 8 consecutive k (k-step chunk >> 1, half chunk & 1)
    const int b_row = tid >> 2, b_chunk = tid & 3;
    const unsigned b_off = (unsigned)((((b_chunk >> 1) * bp_rows + b_row) * 16 + (b_chunk & 1) * 8) * 2);   // bytes
    const int b_wr = b_row * 64 + ((b_chunk ^ ((b_row >> 2) & 3)) * 16);
    const int nk = K / S6_BK;
    const gptr Au = uniform_ptr(A);
    const long bp_tile = 2L * bp_rows * 16;     // elements per 32-deep k-tile of one plane
    auto a_load = [&](f32x4& dst, int kt) {
        kt = min(kt, nk - 1);
        const unsigned ko = kt * S6_BK + a_chunk * 4 + 4 <= ka_valid ? (unsigned)(kt * S6_BK) * 4u : 0u - (unsigned)a_chunk * 16u;
        dst = *reinterpret_cast<gptr_f4>(Au + (a_off + ko));
    };
    auto a_store = [&](const f32x4& src, int buf) {
        unsigned h0, m0, l0, h1, m1, l1;
        split_pair(src.x, src.y, h0, m0, l0);
        split_pair(src.z, src.w, h1, m1, l1);
        unsigned char* d = sm + buf * SW_AIMG + a_wr;
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + 128 * 64) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(d + 2 * 128 * 64) = make_uint2(l0, l1);
    };
    auto b_load = [&](u32x4 (&dst)[3], int kt) {
        kt = min(kt, nk - 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<gptr_u4>(uniform_ptr(bp + p * bp_plane + kt * bp_tile) + b_off);
    };
    auto b_store = [&](const u32x4 (&src)[3], int buf) {
        unsigned char* d = sm + 2 * SW_AIMG + buf * SW_BIMG + b_wr;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(d + p * 256 * 64) = src[p];
    };
    const int sw = (l31 >> 2) & 3;
    const unsigned char* a_rd = sm + (wm * 64 + l31) * 64;
    const unsigned char* b_rd = sm + 2 * SW_AIMG + (wn * 32 + l31) * 64;

    f32x4 aq[2];
    u32x4 bq[3];
    a_load(aq[0], 0);
    b_load(bq, 0);
    a_load(aq[1], 1);
    a_store(aq[0], 0);
    b_store(bq, 0);
    lds_barrier();
    // k-tile kt: images in buffers kt & 1; registers aq[(kt + 1) & 1] hold A of tile kt + 1
    auto tile = [&](auto Uc, int kt) {
        constexpr int U = decltype(Uc)::value;   // kt & 1
        b_load(bq, kt + 1);
        f32x4 a_next;
        a_load(a_next, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ai = a_rd + U * SW_AIMG;
        const unsigned char* bi = b_rd + U * SW_BIMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (TME > 0) {
                bf16x8 fa[TME > 0 ? TME : 1][3], fb[3];
                const int ch = ((2 * s + lh) ^ sw) * 16;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int i = 0; i < TME; ++i) fa[i][p] = *reinterpret_cast<const bf16x8*>(ai + p * 128 * 64 + i * 32 * 64 + ch);
                    fb[p] = *reinterpret_cast<const bf16x8*>(bi + p * 256 * 64 + ch);
                }
                constexpr int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int o = 0; o < 6; ++o)
#pragma unroll
                    for (int i = 0; i < TME; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[o]], fb[PB[o]], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next tile's images are written between the two k-steps by half of the waves and behind the second by the other
            // half (SIMD partners then do their vector / LDS-store work at different moments, see s6_main_loop)
            if (late == (s == 0)) {
                a_store(aq[U ^ 1], U ^ 1);
                b_store(bq, U ^ 1);
            }
        }
        aq[U] = a_next;
        lds_barrier();
    };
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        tile(IC<0>{}, kt);
        tile(IC<1>{}, kt + 1);
    }
    if (kt < nk) tile(IC<0>{}, kt);
}

template <int EPI>
__global__ __launch_bounds__(SW_NT, 4) void lin_s6w_kernel(LinK g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
    int bz, m0, rows;
    const int tile = blockIdx.x;
    if (tile < g.n_big) {
        bz = tile / g.big_per_batch;
        m0 = (tile - bz * g.big_per_batch) * 128;
        rows = 128;
    } else {
        const int j = tile - g.n_big;
        bz = j / g.small_per_batch;
        m0 = g.big_per_batch_rows + (j - bz * g.small_per_batch) * 32;
        rows = 32;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_u >> 3, wn = wave_u & 7;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const uint16_t* bp = g.Bp + (long)bz * g.bp_batch;
    const float* A = g.A + (long)bz * g.a_batch + (long)m0 * g.lda;
    const bool late = (wave_u >> 2) & 1;
    if (rows == 128) s6w_main_loop<2>(acc, smw, A, g.lda, g.M - m0, g.K, g.ka_valid, bp, g.bp_plane, g.bp_rows, tid, lane, wm, wn, late);
    else if (wm == 0) s6w_main_loop<1>(acc, smw, A, g.lda, g.M - m0, g.K, g.ka_valid, bp, g.bp_plane, g.bp_rows, tid, lane, wm, wn, late);
    else s6w_main_loop<0>(acc, smw, A, g.lda, g.M - m0, g.K, g.ka_valid, bp, g.bp_plane, g.bp_rows, tid, lane, wm, wn, late);
    const int col = wn * 32 + (lane & 31);
    const float bj = (EPI == EPI_LNF && g.bias) ? g.bias[(long)bz * g.bias_batch + col] : 0.f;
    lin_ln_rows<2, EPI, 8>(g, acc, reinterpret_cast<float*>(smw), bz, m0, rows == 128 ? 2 : 1, rows == 128 ? 2 : 1, wave, lane, bj);
}

// A plain Linear on the same main loop: C = act(A . B^T + bias), 64 (or 32) rows x 32 NW columns per workgroup (NW = 8:
// N <= 256, NW = 4: N <= 128; four workgroups per CU at NW = 4), optionally split over K (gridDim.y chunks of kchunk, each
// writing its partial sums -- no bias, no activation -- to its own slab C + y * c_split: the consumer adds them in a fixed order).
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void lin_s6_plain_kernel(LinK g) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * S6_BUF];
    int bz, m0, tm_eff;
    const int tile = blockIdx.x;   // (an XCD-contiguous map of the tile list, xcd_contiguous(), was measured: 5-10 % slower)
    if (tile < g.n_big) {
        bz = tile / g.big_per_batch;
        m0 = (tile - bz * g.big_per_batch) * 64;
        tm_eff = 2;
    } else {
        const int j = tile - g.n_big;
        bz = j / g.small_per_batch;
        m0 = g.big_per_batch_rows + (j - bz * g.small_per_batch) * 32;
        tm_eff = 1;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ks = blockIdx.y, k0 = ks * g.kchunk;
    const int klen = min(g.kchunk, g.K - k0);
    const int kav = max(0, min(klen, g.ka_valid - k0));
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long bp_step = (long)g.bp_rows * 16;
    const uint16_t* bp = g.Bp + (long)bz * g.bp_batch + (k0 / 16) * bp_step;
    const unsigned b_lane = (unsigned)((wave * 32 + l31) * 16 + lh * 8) * 2u;   // bytes
    const float* A = g.A + (long)bz * g.a_batch + (long)m0 * g.lda + k0;
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
    if (tm_eff == 2) s6_main_loop<NW, 2>(acc, sm, A, g.lda, g.M - m0, klen, kav, bp, b_lane, g.bp_plane, bp_step, tid, lane, late);
    else s6_main_loop<NW, 1>(acc, sm, A, g.lda, g.M - m0, klen, kav, bp, b_lane, g.bp_plane, bp_step, tid, lane, late);
    const int col = wave * 32 + l31;
    if (col >= g.N) return;
    const bool whole = gridDim.y == 1;
    const float bj = (whole && g.bias) ? g.bias[(long)bz * g.bias_batch + col] : 0.f;
    float* c0 = g.C + (long)ks * g.c_split + (long)bz * g.c_batch + col;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (i < tm_eff && row < g.M) {
                float v = acc[i][r] + bj;
                if (whole && g.act == 1) v = as_relu(v);
                else if (whole && g.act == 2) v = as_sigmoid(v);
                c0[(long)row * g.ldc] = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Output layer of the heads: out = sigmoid(x_hat . W3'^T + b3') (models.py:28-31, 145), N = 2 x n_samples <= 128 columns per
// head, optionally with the training criterion fused into the epilogue (EuclideanDistance + padding mask + mean,
// phoneme_to_articulation/metrics.py:17-24, train_phoneme_to_articulation.py:86-90, and its gradient through the sigmoid).
// One workgroup = 64 frames x the (<= 128) outputs of ONE head: 2 x 4 waves, a wave owns 32 rows x 32 columns (one
// accumulator), so none of the eight waves multiplies padding beyond the N -> 128 round-up (the 256-column kernel above would
// leave four of its eight waves on zeros).  36 KB of LDS, < 64 VGPRs: four workgroups per CU.  Same LDS-DMA ring as above
// (16-deep k-tiles, XOR-swizzled 16-byte chunks, counted vmcnt, raw barriers); every wave issues one A piece (waves 4-7
// repeat pieces 0-3) and one B piece per k-tile.
// Fused criterion: the x coordinates (columns [0, N/2)) and y coordinates ([N/2, N)) of a point lie in different waves, so
// the sigmoid outputs of the tile meet in LDS (the ring's memory); a thread then owns (frame, point) pairs: distance to the
// target, its share of the masked sum, and d loss / d(pre-sigmoid) written where the backward expects d(out).  Partial sums
// leave per workgroup and are added in a fixed order by loss_final (deterministic).
constexpr int ON = 128;

struct LinOutK {
    const float* A; long lda, a_batch;       // x_hat [M][batch][K]
    const float* B; long ldb, b_batch; int b_rows;   // W3' [batch][b_rows >= N][K] (rows N .. b_rows-1 zero)
    const float* bias; long bias_batch;
    float* out; long ldo, o_batch;           // [M][batch][N]
    int M, N, K, batch;
    int n_big, big_per_batch, big_per_batch_rows, small_per_batch;
    // fused criterion (tgt == nullptr: plain output layer)
    const float* tgt; long tgt_T; const int* lengths; int T; float scale;
    float* dout; float* partial;
    const uint16_t* Bp; long bp_plane, bp_batch; int bp_rows;   // lin_out_s6_kernel: W3' as bfloat16 planes (as_lin_out.Bp)
    float* loss; int* counter;                                  // the last workgroup sums the partials (as_lin_out.loss)
};

// The criterion's workgroup sum goes out; with an arrival counter the workgroup that delivers the last one adds them all, thread
// i taking partials i, i + 256, ... and the 256 sums meeting as in loss_final_kernel (metrics.hip): the same value, bit for bit.
// Partials are written through to memory and read back at agent scope (another XCD's L2 does not see a plain store: gemm_f32.hip).
__device__ __forceinline__ void criterion_tail(const LinOutK& g, int tile, float sum, int tid, float* red, int nwaves) {
    if (g.counter == nullptr) {
        if (tid == 0) g.partial[tile] = sum;
        return;
    }
    __shared__ int s_last;
    if (tid == 0) {
        __hip_atomic_store(&g.partial[tile], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int last = atomicAdd(g.counter, 1) == (int)gridDim.x - 1;
        if (last) __hip_atomic_store(g.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    float s = 0.f;
    if (tid < 256)
        for (int i = tid; i < (int)gridDim.x; i += 256) s += __hip_atomic_load(&g.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = as_wave_sum(s);
    __syncthreads();                 // `red` held the workgroup's own wave sums until here
    if ((tid & 63) == 0 && tid < 256) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) g.loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * g.scale;
    (void)nwaves;
}

__global__ __launch_bounds__(NT, 4) void lin_out_kernel(LinOutK g) {
    constexpr int TILE = BK * (64 + ON);
    constexpr int RING = NBUF * TILE;            // 9216 floats = 36 KB; the epilogue's [64][ON] tile (32 KB) reuses it
    __shared__ __attribute__((aligned(16))) float smem[RING];
    int bz, m0, rows;
    if ((int)blockIdx.x < g.n_big) {
        bz = blockIdx.x / g.big_per_batch;
        m0 = (blockIdx.x - bz * g.big_per_batch) * 64;
        rows = 64;
    } else {
        const int j = blockIdx.x - g.n_big;
        bz = j / g.small_per_batch;
        m0 = g.big_per_batch_rows + (j - bz * g.small_per_batch) * 32;
        rows = 32;
    }
    const float* __restrict__ A = g.A + (long)bz * g.a_batch;
    const float* __restrict__ B = g.B + (long)bz * g.b_batch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    // DMA sources: A piece = wave & 3 (16 rows x 16 k), B piece = wave (16 output rows x 16 k); chunk swizzle as above
    const int pa = wave & 3;
    const int ra = pa * 16 + (lane >> 2);
    const int a_gc = ((lane & 3) ^ ((ra >> 2) & 3)) * 4;
    const float* a_src = A + (long)min(m0 + ra, g.M - 1) * g.lda + a_gc;
    const int rb = wave * 16 + (lane >> 2);
    const int b_gc = ((lane & 3) ^ ((rb >> 2) & 3)) * 4;
    const float* b_src = B + (long)min(rb, g.b_rows - 1) * g.ldb + b_gc;   // rows beyond the head's own only feed columns >= N
    const unsigned smem_base = lds_addr(smem);
    auto issue = [&](int kt) {
        const unsigned base = smem_base + (unsigned)((kt % NBUF) * TILE) * 4u;
        glds16(a_src + kt * BK, base + (unsigned)(pa * 1024));
        glds16(b_src + kt * BK, base + (unsigned)((BK * 64 + wave * 256) * 4));
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = g.K / BK;
    const int swz = (l31 >> 2) & 3;
    const bool active = wm == 0 || rows == 64;      // a 32-row tile has one row block: waves 4-7 only help with the DMAs
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) issue(kt + 2);
        const float* tile = smem + (kt % NBUF) * TILE;
        const float* a_s = tile + (wm * 32 + l31) * BK;
        const float* b_s = tile + BK * 64 + (wn * 32 + l31) * BK;
        if (active) {
#pragma unroll
            for (int cc = 0; cc < BK / 8; ++cc) {
                const int slot = ((2 * cc + lh) ^ swz) * 4;
                const float4 av = *reinterpret_cast<const float4*>(a_s + slot);
                const float4 bv = *reinterpret_cast<const float4*>(b_s + slot);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
        }
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // ---- epilogue.  D[row][col]: col = wn * 32 + l31, row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh
    const int col = wn * 32 + l31;
    const float bj = (g.bias && col < g.N) ? g.bias[(long)bz * g.bias_batch + col] : 0.f;
    if (g.tgt == nullptr) {
        if (active && col < g.N) {
            float* o0 = g.out + (long)bz * g.o_batch + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M) o0[(long)row * g.ldo] = as_sigmoid(acc[r] + bj);
            }
        }
        return;
    }
    // fused criterion: sigmoid outputs through LDS ([64][ON]; every ring read is behind the loop's last barrier)
    if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ON + col] = as_sigmoid(acc[r] + bj);
    }
    lds_barrier();
    const int Np = g.N >> 1;                                    // points per contour
    // one wave per frame (row) at a time, lanes over the points: frame index, utterance, validity and the target row are
    // wave-uniform (computed once per row on the scalar unit, no per-pair divisions); rows of a wave are independent, so
    // the next row's target loads are in flight while this row's stores go out
    float part = 0.f;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if (Np <= 64) {
        // a lane owns one point of each of the wave's (up to 8) frames.  ALL target coordinates are requested before the first
        // store goes out: vector memory operations retire in order, so a target load issued behind a frame's stores waits
        // for their acknowledgement -- frame after frame, that chain was most of this epilogue
        float tx[8], ty[8];
        bool in[8], valid[8];
        const int nl = min(lane, Np - 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = wave_u + 8 * k;
            const int frame = m0 + r;
            in[k] = r < rows && frame < g.M;
            const int fc = in[k] ? frame : m0;
            const int b = fc / g.T, t = fc - b * g.T;
            valid[k] = in[k] && t < g.lengths[b];
            const float* tg = g.tgt + (((long)b * g.tgt_T + t) * g.batch + bz) * g.N;
            tx[k] = tg[nl];
            ty[k] = tg[Np + nl];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (!in[k]) continue;                       // wave-uniform
            const int r = wave_u + 8 * k;
            const long frame = m0 + r;
            float* o = g.out + frame * g.ldo + (long)bz * g.o_batch;
            float* dz = g.dout + frame * g.ldo + (long)bz * g.o_batch;
            if (lane < Np) {
                const float ox = smem[r * ON + lane], oy = smem[r * ON + Np + lane];
                o[lane] = ox;
                o[Np + lane] = oy;
                float gx = 0.f, gy = 0.f;
                if (valid[k]) {
                    const float dx = ox - tx[k], dy = oy - ty[k];
                    const float d = sqrtf(dx * dx + dy * dy);
                    part += d;
                    const float gg = g.scale / d;                   // NaN at zero distance, as torch autograd
                    gx = dx * gg * ox * (1.f - ox);                  // through the sigmoid (same product order as the unfused kernels)
                    gy = dy * gg * oy * (1.f - oy);
                }
                dz[lane] = gx;
                dz[Np + lane] = gy;
            }
        }
    } else
    for (int r = wave_u; r < rows; r += 8) {
        const int frame = m0 + r;
        if (frame >= g.M) break;
        const int b = frame / g.T, t = frame - b * g.T;
        const bool valid = t < g.lengths[b];
        float* o = g.out + (long)frame * g.ldo + (long)bz * g.o_batch;
        float* dz = g.dout + (long)frame * g.ldo + (long)bz * g.o_batch;
        const float* tg = g.tgt + (((long)b * g.tgt_T + t) * g.batch + bz) * g.N;
        for (int n = lane; n < Np; n += 64) {
            const float ox = smem[r * ON + n], oy = smem[r * ON + Np + n];
            o[n] = ox;
            o[Np + n] = oy;
            float gx = 0.f, gy = 0.f;
            if (valid) {
                const float dx = ox - tg[n], dy = oy - tg[Np + n];
                const float d = sqrtf(dx * dx + dy * dy);
                part += d;
                const float gg = g.scale / d;                   // NaN at zero distance, as torch autograd
                gx = dx * gg * ox * (1.f - ox);                  // through the sigmoid (same product order as the unfused kernels)
                gy = dy * gg * oy * (1.f - oy);
            }
            dz[n] = gx;
            dz[Np + n] = gy;
        }
    }
    part = wave_sum(part);
    __shared__ float red[8];
    if (lane == 0) red[wave] = part;
    __syncthreads();
    float wg_sum = 0.f;
    if (tid == 0) wg_sum = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    criterion_tail(g, blockIdx.x, wg_sum, tid, red, 8);
}

// The output layer on the bf16 matrix instruction (see lin_s6_kernel): 4 waves, wave w = columns 32 w .. 32 w + 31 of all 64
// rows (two accumulators), A split once per workgroup into the plane image, W3' planes from L2 into registers.  32 KB of
// LDS, <= 128 VGPRs: four workgroups per CU.  Epilogue as lin_out_kernel's, with 16 instead of 8 frames per wave.
__global__ __launch_bounds__(256, 4) void lin_out_s6_kernel(LinOutK g) {
    __shared__ __attribute__((aligned(16))) float smem[64 * ON];
    static_assert(64 * ON * 4 >= 2 * S6_BUF, "the epilogue's tile holds both plane images");
    int bz, m0, rows;
    const int tile = blockIdx.x;   // (an XCD-contiguous map of the tile list, xcd_contiguous(), was measured: 5-10 % slower)
    if (tile < g.n_big) {
        bz = tile / g.big_per_batch;
        m0 = (tile - bz * g.big_per_batch) * 64;
        rows = 64;
    } else {
        const int j = tile - g.n_big;
        bz = j / g.small_per_batch;
        m0 = g.big_per_batch_rows + (j - bz * g.small_per_batch) * 32;
        rows = 32;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const uint16_t* bp = g.Bp + (long)bz * g.bp_batch;
    const unsigned b_lane = (unsigned)((wave * 32 + l31) * 16 + lh * 8) * 2u;   // bytes
    unsigned char* sm = reinterpret_cast<unsigned char*>(smem);
    const float* A = g.A + (long)bz * g.a_batch + (long)m0 * g.lda;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if (rows == 64) s6_main_loop<4, 2>(acc, sm, A, g.lda, g.M - m0, g.K, g.K, bp, b_lane, g.bp_plane, (long)g.bp_rows * 16, tid, lane, wave_u >= 2);
    else s6_main_loop<4, 1>(acc, sm, A, g.lda, g.M - m0, g.K, g.K, bp, b_lane, g.bp_plane, (long)g.bp_rows * 16, tid, lane, wave_u >= 2);
    // ---- epilogue.  D[row][col]: col = wave * 32 + l31, row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh
    const int col = wave * 32 + l31;
    const float bj = (g.bias && col < g.N) ? g.bias[(long)bz * g.bias_batch + col] : 0.f;
    const int nblk = rows / 32;
    if (g.tgt == nullptr) {
        if (col < g.N) {
            float* o0 = g.out + (long)bz * g.o_batch + col;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (i < nblk && row < g.M) o0[(long)row * g.ldo] = as_sigmoid(acc[i][r] + bj);
                }
        }
        return;
    }
    // fused criterion: sigmoid outputs through LDS ([64][ON]; every plane-image read is behind the loop's last barrier)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (i < nblk) smem[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ON + col] = as_sigmoid(acc[i][r] + bj);
    lds_barrier();
    const int Np = g.N >> 1;                                    // points per contour
    float part = 0.f;
    if (Np <= 64) {
        // one wave per frame at a time, lanes over the points; ALL target coordinates of the wave's 16 frames are requested
        // before the first store goes out (vector memory operations retire in order: a load issued behind a frame's stores
        // waits for their acknowledgement -- two passes of 8 frames cost the second pass exactly that, see lin_out_kernel)
        const int nl = min(lane, Np - 1);
        float tx[16], ty[16];
        bool in[16], valid[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int r = wave_u + 4 * k;
            const int frame = m0 + r;
            in[k] = r < rows && frame < g.M;
            const int fc = in[k] ? frame : m0;
            const int b = fc / g.T, t = fc - b * g.T;
            valid[k] = in[k] && t < g.lengths[b];
            const float* tg = g.tgt + (((long)b * g.tgt_T + t) * g.batch + bz) * g.N;
            tx[k] = tg[nl];
            ty[k] = tg[Np + nl];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (!in[k]) continue;                       // wave-uniform
            const int r = wave_u + 4 * k;
            const long frame = m0 + r;
            float* o = g.out + frame * g.ldo + (long)bz * g.o_batch;
            float* dz = g.dout + frame * g.ldo + (long)bz * g.o_batch;
            if (lane < Np) {
                const float ox = smem[r * ON + lane], oy = smem[r * ON + Np + lane];
                o[lane] = ox;
                o[Np + lane] = oy;
                float gx = 0.f, gy = 0.f;
                if (valid[k]) {
                    const float dx = ox - tx[k], dy = oy - ty[k];
                    const float d = sqrtf(dx * dx + dy * dy);
                    part += d;
                    const float gg = g.scale / d;                   // NaN at zero distance, as torch autograd
                    gx = dx * gg * ox * (1.f - ox);                  // through the sigmoid (same product order as the unfused kernels)
                    gy = dy * gg * oy * (1.f - oy);
                }
                dz[lane] = gx;
                dz[Np + lane] = gy;
            }
        }
    } else
    for (int r = wave_u; r < rows; r += 4) {
        const int frame = m0 + r;
        if (frame >= g.M) break;
        const int b = frame / g.T, t = frame - b * g.T;
        const bool valid = t < g.lengths[b];
        float* o = g.out + (long)frame * g.ldo + (long)bz * g.o_batch;
        float* dz = g.dout + (long)frame * g.ldo + (long)bz * g.o_batch;
        const float* tg = g.tgt + (((long)b * g.tgt_T + t) * g.batch + bz) * g.N;
        for (int n = lane; n < Np; n += 64) {
            const float ox = smem[r * ON + n], oy = smem[r * ON + Np + n];
            o[n] = ox;
            o[Np + n] = oy;
            float gx = 0.f, gy = 0.f;
            if (valid) {
                const float dx = ox - tg[n], dy = oy - tg[Np + n];
                const float d = sqrtf(dx * dx + dy * dy);
                part += d;
                const float gg = g.scale / d;
                gx = dx * gg * ox * (1.f - ox);
                gy = dy * gg * oy * (1.f - oy);
            }
            dz[n] = gx;
            dz[Np + n] = gy;
        }
    }
    part = wave_sum(part);
    __shared__ float red[4];
    if (lane == 0) red[wave] = part;
    __syncthreads();
    float wg_sum = 0.f;
    if (tid == 0) wg_sum = (red[0] + red[1]) + (red[2] + red[3]);   // (tile order, like lin_out_kernel: same final sum order)
    criterion_tail(g, tile, wg_sum, tid, red, 4);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tile list.  Two workgroups per CU = 512 slots; a launch of T equal tiles takes ceil(T / 512) rounds, and the head
// layers' 1100 tiles of 64 rows were 3 rounds of which the last held 76 tiles (measured: 31 us of a 118 us launch with
// 180 CUs idle).  So: as many 64-row tiles as fill whole rounds, dispatched first, then the remaining rows as 32-row
// tiles (half the matrix work each), which pack the last round about half as high.
template <bool B_KC, int EPI>
int launch(const LinK& k, hipStream_t st) {
    LinK kk = k;
    static const int nbuf = AS_DIAG_INT("AS_LIN_NBUF", 3);      // 2: two ring slots, three workgroups per CU (diagnostic)
    const int slots = nbuf == 2 ? 768 : 512;
    const long units = (long)as_cdiv(k.M, 64) * k.batch;      // work in 64-row tiles
    const long rounds = units / slots;
    static const bool all_big = AS_DIAG_SET("AS_LIN_ALLBIG");  // ablation: 64-row tiles only (+ a ragged end)
    int x = (int)(rounds * slots / k.batch);                    // 64-row tiles per head that fill whole rounds
    if (x > k.M / 64 || all_big || k.tile_rows == 64) x = k.M / 64;
    if (k.tile_rows == 32) x = 0;
    const int rest = k.M - x * 64;
    kk.big_per_batch = x > 0 ? x : 1;
    kk.big_per_batch_rows = x * 64;
    kk.n_big = x * k.batch;
    kk.small_per_batch = as_cdiv(rest, 32);
    const long total = (long)kk.n_big + (long)kk.small_per_batch * k.batch;
    if (kk.small_per_batch == 0) kk.small_per_batch = 1;
    if (kk.Bp && EPI != EPI_PLAIN && k.tile_rows == 0 && (long)as_cdiv(k.M, 128) * k.batch >= 256 && k.bp_rows == BN) {
        // at least one whole round of 128-row tiles: the wide kernel (one 16-wave workgroup per CU, B planes through LDS)
        // MEASURED SLOWER than the 64-row kernel (head Linear 2: 77.5 vs 63.4 us, dx2 86.5 vs 80.2 us, gpurun_out/r04a): all 16
        // waves of the CU meet at one barrier per k-tile, and a workgroup alone on its CU keeps the matrix pipe about half busy
        // (per-tile stamps, tools/s6_trace.py: ~1000 cycles until a tile's first fragments are in registers, ~900 at the barrier,
        // ~550 in the split, against 768 of matrix instructions per wave) -- two independent 8-wave workgroups fill each
        // other's gaps, one 16-wave workgroup does not.  Kept for the diagnostic build only (AS_LIN_WIDE=1).
        static const bool wide = AS_DIAG_SET("AS_LIN_WIDE");
        if (wide) {
            static const hipError_t attr =
                hipFuncSetAttribute(reinterpret_cast<const void*>(lin_s6w_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS);
            if (attr != hipSuccess) {
                as_set_error("as_lin_s6w: cannot reserve %d bytes of LDS: %s", SW_LDS, hipGetErrorString(attr));
                return (int)attr;
            }
            const long t128 = (long)(k.M / 128) * k.batch;          // whole 128-row tiles
            int xw = (int)((t128 / 256) * 256 / k.batch);            // ... per head that fill whole rounds of the 256 CUs
            if (xw > k.M / 128) xw = k.M / 128;
            const int restw = k.M - xw * 128;
            kk.big_per_batch = xw > 0 ? xw : 1;
            kk.big_per_batch_rows = xw * 128;
            kk.n_big = xw * k.batch;
            kk.small_per_batch = as_cdiv(restw, 32);
            const long totalw = (long)kk.n_big + (long)kk.small_per_batch * k.batch;
            if (kk.small_per_batch == 0) kk.small_per_batch = 1;
            hipLaunchKernelGGL((lin_s6w_kernel<EPI>), dim3((unsigned)totalw), dim3(SW_NT), SW_LDS, st, kk);
            AS_LAUNCH_CHECK("as_lin_s6w");
            return 0;
        }
    }
    if (kk.Bp) {   // the products on the bf16 matrix instruction, B as planes (lin_s6_kernel)
        hipLaunchKernelGGL((lin_s6_kernel<EPI>), dim3((unsigned)total), dim3(NT), 0, st, kk);
        AS_LAUNCH_CHECK("as_lin_s6");
        return 0;
    }
#ifdef AS_DIAG
    if (nbuf == 2) hipLaunchKernelGGL((lin_f32_kernel<64, B_KC, EPI, 2>), dim3((unsigned)total), dim3(NT), 0, st, kk);
    else
#endif
    hipLaunchKernelGGL((lin_f32_kernel<64, B_KC, EPI>), dim3((unsigned)total), dim3(NT), 0, st, kk);
    AS_LAUNCH_CHECK("as_lin_f32");
    return 0;
}

unsigned long long* g_dbg = nullptr;
long g_dbg_max = 0;

}  // namespace

extern "C" void as_lin_debug_stamps(uint64_t* buf, int64_t max_workgroups) {
    g_dbg = (unsigned long long*)buf;
    g_dbg_max = buf ? max_workgroups : 0;
}

// see gemm_internal.h.  Returns 1 if launched, 0 if the arguments are outside what the kernel is built for (the caller
// then takes the general GEMM + row kernels), < 0 on a launch error.
int as_lin_try(const as_lin* a, hipStream_t st) {
    static const bool off = AS_DIAG_SET("AS_NO_LIN");  // ablation: the round-1 path
    if (off) return 0;
    if (a->K % BK || a->K < BK || a->N > BN || a->N < 4 || a->M < 1 || a->batch < 1) return 0;
    if (!aligned16(a->A) || !aligned16(a->B) || a->lda % 4 || a->ldb % 4 || a->a_batch % 4 || a->b_batch % 4) return 0;
    if (!a->b_kc && a->N % 4) return 0;
    if (a->epi != EPI_PLAIN && (a->N != BN || !a->C)) return 0;
    if (a->epi == EPI_PLAIN && a->N <= BN / 2) return 0;  // half of the 8 feature-side waves would multiply padding
    if ((long)as_cdiv(a->M, 64) * a->batch > (1L << 30)) return 0;
    LinK k{};
    k.A = a->A; k.lda = a->lda; k.a_batch = a->a_batch;
    k.B = a->B; k.ldb = a->ldb; k.b_batch = a->b_batch;
    k.C = a->C; k.ldc = a->ldc; k.c_batch = a->c_batch;
    k.bias = a->bias; k.bias_batch = a->bias_batch;
    k.M = a->M; k.N = a->N; k.K = a->K; k.ka_valid = a->ka_valid > 0 ? a->ka_valid : a->K; k.batch = a->batch; k.act = a->act;
    k.tile_rows = a->tile_rows;
    k.eps = 1e-5f;
#ifdef AS_DIAG
    static const int abl = AS_DIAG_INT("AS_LIN_ABL", 0);
    k.abl = abl;
    // round 3: 1 (second workgroup of a CU half a tile late) is -3 % on one box and +3 % on the next: noise; < 0 = XCD de-phasing
    // (strictly slower); with AS_LIN_NBUF=2 (three workgroups per CU on a 2-slot ring) thirds of a tile: no change either
    static const int stagger = AS_DIAG_INT("AS_LIN_STAGGER", 0);
    k.stagger = stagger;
#endif
    k.dbg = g_dbg; k.dbg_max = g_dbg_max;
    k.rstd = a->rstd; k.bits = a->bits;
    k.xhat = a->xhat; k.ldx = a->ldx; k.x_batch = a->x_batch; k.rstd_in = a->rstd_in; k.bits_in = a->bits_in;
    if (k.ka_valid % 4) return 0;
    // B as bfloat16 planes (as_emit_planes) and the split arithmetic on: the bf16-MFMA kernel; else the exact fp32 one
    if (a->Bp && as_matrix_arith() == AS_ARITH_BF16X6 && a->K % S6_BK == 0 && a->bp_rows >= BN && (reinterpret_cast<uintptr_t>(a->Bp) & 15) == 0 &&
        a->bp_plane % 8 == 0 && a->bp_batch % 8 == 0 && a->lda < (1L << 23)) {
        k.Bp = a->Bp; k.bp_plane = a->bp_plane; k.bp_batch = a->bp_batch; k.bp_rows = a->bp_rows;
    }
    if (a->epi == EPI_LNF) {
        if (!a->rstd || !a->bits || !a->b_kc) return 0;
        return launch<true, EPI_LNF>(k, st) == 0 ? 1 : -1;
    }
    if (a->epi == EPI_LNB) {
        if (!a->xhat || !a->rstd_in || !a->bits_in || a->b_kc) return 0;
        return launch<false, EPI_LNB>(k, st) == 0 ? 1 : -1;
    }
    if (a->b_kc) return launch<true, EPI_PLAIN>(k, st) == 0 ? 1 : -1;
    return launch<false, EPI_PLAIN>(k, st) == 0 ? 1 : -1;
}

// see gemm_internal.h
int as_lin_plain_s6(const as_lin* a, int ksplit, long c_split, hipStream_t st) {
    static const bool off = AS_DIAG_SET("AS_NO_PLAIN_S6");   // diagnostic: callers fall back to the general fp32 kernel
    if (off || as_matrix_arith() != AS_ARITH_BF16X6 || !a->Bp) return 0;
    if (a->epi != EPI_PLAIN || a->K % S6_BK || a->K < S6_BK || a->N > BN || a->N < 1 || a->M < 1 || a->batch < 1 || !a->C) return 0;
    if (!aligned16(a->A) || a->lda % 4 || a->a_batch % 4 || a->lda >= (1L << 23)) return 0;
    if ((reinterpret_cast<uintptr_t>(a->Bp) & 15) || a->bp_plane % 8 || a->bp_batch % 8) return 0;
    const int nw = a->N <= 128 ? 4 : 8;
    if (a->bp_rows < nw * 32) return 0;
    if (ksplit < 1) ksplit = 1;
    if (ksplit > 1 && (a->bias || a->act)) return 0;
    LinK k{};
    k.A = a->A; k.lda = a->lda; k.a_batch = a->a_batch;
    k.Bp = a->Bp; k.bp_plane = a->bp_plane; k.bp_batch = a->bp_batch; k.bp_rows = a->bp_rows;
    k.C = a->C; k.ldc = a->ldc; k.c_batch = a->c_batch;
    k.bias = a->bias; k.bias_batch = a->bias_batch;
    k.M = a->M; k.N = a->N; k.K = a->K; k.ka_valid = a->ka_valid > 0 ? a->ka_valid : a->K; k.batch = a->batch; k.act = a->act;
    if (k.ka_valid % 4) return 0;
    k.kchunk = (int)as_round_up(as_cdiv(a->K, ksplit), S6_BK);
    ksplit = as_cdiv(a->K, k.kchunk);
    k.c_split = c_split;
    // tiles: 64 rows unless that leaves CUs idle (fewer workgroups than 1.5 x 256)
    int tile = a->tile_rows;
    if (tile != 32 && tile != 64) tile = (long)as_cdiv(a->M, 64) * a->batch * ksplit >= 384 ? 64 : 32;
    const int x = tile == 64 ? a->M / 64 : 0;
    const int rest = a->M - x * 64;
    k.big_per_batch = x > 0 ? x : 1;
    k.big_per_batch_rows = x * 64;
    k.n_big = x * a->batch;
    k.small_per_batch = as_cdiv(rest, 32);
    const long total = (long)k.n_big + (long)k.small_per_batch * a->batch;
    if (k.small_per_batch == 0) k.small_per_batch = 1;
    if (total > (1L << 30) || ksplit > 65535) return 0;
    if (nw == 4) hipLaunchKernelGGL((lin_s6_plain_kernel<4>), dim3((unsigned)total, ksplit), dim3(256), 0, st, k);
    else hipLaunchKernelGGL((lin_s6_plain_kernel<8>), dim3((unsigned)total, ksplit), dim3(512), 0, st, k);
    AS_LAUNCH_CHECK("as_lin_plain_s6");
    return 1;
}

// Output layer of the heads, optionally with the masked Euclidean criterion and its gradient fused in (see lin_out_kernel).
// 1 = launched (fused: *n_partials workgroup sums were written to `partial`), 0 = not a case, < 0 = error.
int as_lin_out_try(const as_lin_out* a, int* n_partials, hipStream_t st) {
    static const bool off = AS_DIAG_SET("AS_NO_LIN_OUT");  // ablation: the general GEMM (+ the separate criterion kernel)
    if (off) return 0;
    if (a->K % BK || a->K < BK || a->N > ON || a->N < 4 || a->N % 2 || a->M < 1 || a->batch < 1) return 0;
    if (!aligned16(a->A) || !aligned16(a->B) || a->lda % 4 || a->ldb % 4 || a->a_batch % 4 || a->b_batch % 4) return 0;
    if (a->b_rows < a->N) return 0;
    LinOutK k{};
    k.A = a->A; k.lda = a->lda; k.a_batch = a->a_batch;
    k.B = a->B; k.ldb = a->ldb; k.b_batch = a->b_batch; k.b_rows = a->b_rows;
    k.bias = a->bias; k.bias_batch = a->bias_batch;
    k.out = a->out; k.ldo = a->ldo; k.o_batch = a->o_batch;
    k.M = a->M; k.N = a->N; k.K = a->K; k.batch = a->batch;
    k.tgt = a->tgt; k.tgt_T = a->tgt_T; k.lengths = a->lengths; k.T = a->T; k.scale = a->scale; k.dout = a->dout; k.partial = a->partial;
    if (k.tgt && (!k.lengths || !k.dout || !k.partial || k.T < 1 || k.ldo != (long)k.batch * k.N || k.o_batch != k.N)) return 0;
    // tile list as for the 256-column kernel: 64-row tiles that fill whole rounds of the 4 x 256 resident slots, 32-row tiles
    // over the rest
    constexpr int slots = 1024;
    const long units = (long)as_cdiv(k.M, 64) * k.batch;
    const long rounds = units / slots;
    int x = (int)(rounds * slots / k.batch);
    if (x > k.M / 64) x = k.M / 64;
    if (rounds == 0) x = k.M / 64;                       // less than one round: plain 64-row tiles (+ a ragged end)
    const int rest = k.M - x * 64;
    k.big_per_batch = x > 0 ? x : 1;
    k.big_per_batch_rows = x * 64;
    k.n_big = x * k.batch;
    k.small_per_batch = as_cdiv(rest, 32);
    const long total = (long)k.n_big + (long)k.small_per_batch * k.batch;
    if (k.small_per_batch == 0) k.small_per_batch = 1;
    if (k.tgt && total > a->partial_capacity) return 0;
    static const bool no_tail = AS_DIAG_SET("AS_NO_LOSS_TAIL");   // ablation: the separate final-sum kernel
    if (k.tgt && a->loss && !no_tail) {
        k.counter = as_arrival_counter(st);
        k.loss = k.counter ? a->loss : nullptr;
    }
    const int left = k.counter ? 0 : (int)total;   // partials that still await as_loss_final
    static const bool no_s6 = AS_DIAG_SET("AS_NO_LIN_OUT_S6");   // diagnostic: the output layer on the fp32 instruction
    if (!no_s6 && a->Bp && as_matrix_arith() == AS_ARITH_BF16X6 && a->K % S6_BK == 0 && a->bp_rows >= ON && (reinterpret_cast<uintptr_t>(a->Bp) & 15) == 0 &&
        a->bp_plane % 8 == 0 && a->bp_batch % 8 == 0 && a->lda < (1L << 23)) {
        k.Bp = a->Bp; k.bp_plane = a->bp_plane; k.bp_batch = a->bp_batch; k.bp_rows = a->bp_rows;
        hipLaunchKernelGGL(lin_out_s6_kernel, dim3((unsigned)total), dim3(256), 0, st, k);
        AS_LAUNCH_CHECK("as_lin_out_s6");
        if (n_partials) *n_partials = left;
        return 1;
    }
    hipLaunchKernelGGL(lin_out_kernel, dim3((unsigned)total), dim3(NT), 0, st, k);
    AS_LAUNCH_CHECK("as_lin_out");
    if (n_partials) *n_partials = left;
    return 1;
}

// ---- include/artspeech_hip.h: as_linear_fwd
namespace {
// column blocks of the split-arithmetic plain kernel for an N-column Linear: block width (rows of the plane image) and count
inline bool linear_blocks(int N, int* bn, int* nb) {
    if (N <= 256) { *bn = N <= 128 ? 128 : 256; *nb = 1; return true; }
    if (N % 256 == 0) { *bn = 256; *nb = N / 256; return true; }
    return false;
}
}  // namespace

extern "C" int64_t as_linear_planes_floats(int32_t N, int32_t K) {
    int bn, nb;
    if (N < 1 || K < 1 || !linear_blocks(N, &bn, &nb)) return 0;
    return as_round_up(as_planes_floats(nb, bn, (int)as_round_up(K, 32)), 64);
}

extern "C" int as_linear_fwd(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* out, int64_t ldo,
                             int32_t M, int32_t N, int32_t K, int32_t act, float* planes_ws, void* stream) {
    AS_REQUIRE(A && W && out && M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldo >= N && act >= 0 && act <= 2, AS_ERR_BAD_ARG,
               "as_linear_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    int bn, nb;
    if (!planes_ws) {   // both operands split inside the kernel (gemm_s6.hip): no scratch needed
        const int took = as_gemm_s6_nt(A, lda, 0, W, ldw, 0, bias, 0, out, ldo, 0, M, N, K, 1, act, st);
        AS_REQUIRE(took >= 0, took, "as_linear_fwd: launch failed");
        if (took) return 0;
    }
    if (as_matrix_arith() == AS_ARITH_BF16X6 && planes_ws && K % S6_BK == 0 && lda % 4 == 0 && linear_blocks(N, &bn, &nb) &&
        (reinterpret_cast<uintptr_t>(planes_ws) & 15) == 0) {
        const int rows_last = N - (nb - 1) * bn;   // (nb > 1: whole blocks)
        as_planes_job j{W, ldw, 1, (long)bn * ldw, nb, nb > 1 ? bn : rows_last, K, bn, K, reinterpret_cast<uint16_t*>(planes_ws)};
        AS_TRY(as_emit_planes(&j, 1, st));
        as_lin l{};
        l.A = A; l.lda = lda;
        l.Bp = reinterpret_cast<const uint16_t*>(planes_ws); l.bp_rows = bn; l.bp_batch = as_planes_batch_stride(bn, K); l.bp_plane = nb * l.bp_batch;
        l.C = out; l.ldc = ldo; l.c_batch = bn;
        l.bias = bias; l.bias_batch = bn; l.act = act;
        l.M = M; l.N = nb > 1 ? bn : N; l.K = K; l.batch = nb;
        const int took = as_lin_plain_s6(&l, 1, 0, st);
        AS_REQUIRE(took >= 0, took, "as_linear_fwd: launch failed");
        if (took) return 0;
    }
    as_gemm g{};
    g.A = A; g.B = W; g.C = out; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.a_i = lda; g.a_k = 1; g.b_j = ldw; g.b_k = 1; g.ldc = ldo; g.batch = 1; g.act = act;
    return as_gemm_f32(&g, st);
}
