// Strided-batched fp32 GEMM on the CDNA4 f32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32, a
// k-ordered fmaf chain, 64 FLOP/clk/SIMD).  One kernel template serves the three shapes the training
// step needs -- forward linears (both operands reduction-contiguous), input gradients and weight
// gradients (reduction index strided) -- by choosing the LDS image per operand:
//   reduction-contiguous operand  -> image [i][BK+1]  (odd row stride: conflict-free ds_read_b32 column reads)
//   output-contiguous operand     -> image [k][BI]    (lanes 0-31 / 32-63 read two consecutive k rows)
// so that a lane's MFMA operand (A[i = lane&31][k = lane>>5]) is always one conflict-free ds_read_b32.
// 256 threads = 4 waves in a 2x2 arrangement; global->register->LDS staging is software pipelined
// (loads of tile t+1 in flight while the matrix core works on tile t), one barrier per K tile.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <hip/hip_ext.h>

#include "gemm_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;

struct GemmK {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long a_i, a_k, b_j, b_k, ldc;
    long a_batch, b_batch, c_batch, bias_batch;
    int act, accumulate, b_kshift, b_kT, b_kshift_batch;
    int a_vec, b_vec;
    // split-K: blockIdx.y owns reduction range [y*kchunk, (y+1)*kchunk); partial tiles go to a dense slab
    // [split][batch][M][N + has_colsum]; a second kernel sums the slabs in a fixed order (deterministic).
    int splitk, kchunk, batch;
    // split-K with splitk % 8 == 0: all output tiles of one reduction chunk go to ONE XCD (workgroup w runs on XCD w % 8, each
    // XCD has its own L2): the chunk's operand panels are then fetched from memory once and shared by its tiles, instead of
    // (nearly) once per XCD -- the three GRU-side weight gradients of the BiGRU step: 83 -> 32 MB fetched per launch (PMC)
    int xcd_chunks;
    // big launches of 128 x 128 tiles with a reduction-contiguous A (activations streamed from HBM, N > 128): the n-tiles of one
    // 128-row panel of A run on ONE XCD next to each other (panel = XCD + 8 * round), so the panel is fetched once, not once
    // per n-tile.  Only the order of the work list changes: every tile's sum is what it was, bit for bit.
    int xcd_panels;
    float* slab;
    // optional arrival counters, one per (batch, output tile), zero between launches: the workgroup that delivers the last
    // partial of a tile sums the slabs itself (fixed k order) and runs the ordinary epilogue -- no second kernel
    int* counters;
    // optional fused column sum of the (output-contiguous) A operand: colsum[i] = sum_k Aop[i][k]
    float* colsum; long colsum_batch;
    // grouped batches: per-batch element offsets (device arrays) override the linear batch strides
    const long* a_off; const long* b_off; const long* c_off; const long* bias_off;
    // optional extra operands (as_gemm.res / .mask_bits / .relu_bits): C = keep ? act(res + acc + bias) : 0
    const float* res; long res_ld, res_batch; const long* res_off;
    const unsigned* mask_bits; long mask_batch;
    unsigned* relu_bits; long relu_bits_batch;
    // optional segmented reduction (as_gemm.k_seg): per (batch, segment) element offsets of the A rows and the B panel
    int k_seg;
    int k_tri;   // as_gemm.k_tri: the A operand is exactly zero for k < i (1) or k > i (2): a tile's reduction range shrinks
                 // (kept in k_seg's padding: a larger kernel-argument block measurably slows the BiGRU step's 64 x 64 launches)
    const long* a_seg_off; const long* b_seg_off;
};

// device-scope accesses for data handed between workgroups of one launch (they may sit on different XCDs, whose L2s are not
// coherent with each other for ordinary accesses)
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int BI, bool KC> struct Img { static constexpr int size = KC ? BI * (BK + 1) : BK * BI; };

// global -> registers for one operand tile.  I = extent of the output index, K = extent of the reduction.
// FAST: 16-byte aligned operand whose contiguous extent is a multiple of 4, so every float4 is wholly
// inside or wholly outside: the loads are branch-free (clamped address + select), issued back to back
// and waited for once.  (A per-element "load or zero" branch makes hipcc wait vmcnt(0) per element.)
template <int BI, bool KC, bool FAST>
__device__ __forceinline__ void tile_load(float4 (&r)[BI / 32], const float* __restrict__ p, long s_i, long s_k,
                                          int i0, int k0, int I, int K, bool vec, int kshift, int kT, int tid) {
    constexpr int P = BI / 32;
    if constexpr (FAST && KC) {
        const int kq = tid & 7, row0 = tid >> 3;
        const int k = k0 + kq * 4;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            const int i = i0 + row0 + 32 * pp;
            const bool ok = i < I && k < K;
            const float4 v = *reinterpret_cast<const float4*>(p + (ok ? (long)i * s_i + k : 0L));
            r[pp] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else if constexpr (FAST && !KC) {
        constexpr int V4 = BI / 4;
        constexpr int RP = 256 / V4;
        const int iq = tid % V4, kr0 = tid / V4;
        const int i = i0 + iq * 4;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            const int k = k0 + kr0 + RP * pp;
            bool ok = k < K && i < I;
            if (kT > 0) {
                const int t = k % kT + kshift;
                ok = ok && t >= 0 && t < kT;
            }
            const float4 v = *reinterpret_cast<const float4*>(p + (ok ? ((long)k + (kT > 0 ? kshift : 0)) * s_k + i : 0L));
            r[pp] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else if constexpr (KC) {
        const int kq = tid & 7, row0 = tid >> 3;
        const int k = k0 + kq * 4;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            const int i = i0 + row0 + 32 * pp;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < I) {
                const float* q = p + (long)i * s_i + k;
                if (vec && k + 3 < K) {
                    v = *reinterpret_cast<const float4*>(q);
                } else {
                    if (k < K) v.x = q[0];
                    if (k + 1 < K) v.y = q[1];
                    if (k + 2 < K) v.z = q[2];
                    if (k + 3 < K) v.w = q[3];
                }
            }
            r[pp] = v;
        }
    } else {
        constexpr int V4 = BI / 4;          // float4 per k row
        constexpr int RP = 256 / V4;        // k rows per pass
        const int iq = tid % V4, kr0 = tid / V4;
        const int i = i0 + iq * 4;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            const int k = k0 + kr0 + RP * pp;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = k < K;
            long ks = k;
            if (kT > 0) {
                const int t = k % kT + kshift;
                ok = ok && t >= 0 && t < kT;
                ks = (long)k + kshift;
            }
            if (ok) {
                const float* q = p + ks * s_k + i;
                if (vec && i + 3 < I) {
                    v = *reinterpret_cast<const float4*>(q);
                } else {
                    if (i < I) v.x = q[0];
                    if (i + 1 < I) v.y = q[1];
                    if (i + 2 < I) v.z = q[2];
                    if (i + 3 < I) v.w = q[3];
                }
            }
            r[pp] = v;
        }
    }
}

template <int BI, bool KC>
__device__ __forceinline__ void tile_store(float* __restrict__ s, const float4 (&r)[BI / 32], int tid) {
    constexpr int P = BI / 32;
    if constexpr (KC) {
        const int kq = tid & 7, row0 = tid >> 3;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            float* d = s + (row0 + 32 * pp) * (BK + 1) + kq * 4;
            d[0] = r[pp].x; d[1] = r[pp].y; d[2] = r[pp].z; d[3] = r[pp].w;
        }
    } else {
        constexpr int V4 = BI / 4;
        constexpr int RP = 256 / V4;
        const int iq = tid % V4, kr0 = tid / V4;
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
            *reinterpret_cast<float4*>(s + (kr0 + RP * pp) * BI + iq * 4) = r[pp];
    }
}

template <int BI, bool KC>
__device__ __forceinline__ float frag(const float* __restrict__ s, int i, int k) {
    return KC ? s[i * (BK + 1) + k] : s[k * BI + i];
}

// EXT: the instantiation that knows the optional extra operands (res / mask_bits / relu_bits) and the segmented reduction; the plain one
// carries neither (its register budget at three workgroups per CU has no room for them).
template <int BM, int BN, bool A_KC, bool B_KC, bool FAST, bool EXT = false>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128 ? 3 : 4)) void gemm_f32_kernel(GemmK g) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    // ONE LDS image per operand: the next tile waits in registers while this one is consumed, so a
    // second LDS buffer would only halve the resident workgroups (34 KB -> 4 per CU, 4 waves per SIMD:
    // other workgroups' MFMAs fill this one's barrier / prologue / epilogue bubbles).
    __shared__ __attribute__((aligned(16))) float sA[1][Img<BM, A_KC>::size];
    __shared__ __attribute__((aligned(16))) float sB[1][Img<BN, B_KC>::size];
    __shared__ int s_last;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    // persistent workgroups: each walks output tiles (tile, k-split, batch) with the grid as stride.  The next tile's first
    // operand loads are issued BEFORE this tile's C stores, and the stores are fire-and-forget: they drain under the next
    // tile's MFMAs instead of holding the workgroup's slot until they land (measured before: 13 us of a 53 us tile at
    // K = 256, every workgroup of a round storing at the same moment).
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int per_split = tiles_m * tiles_n;
    const long per_batch = (long)per_split * g.splitk;
    const long total = per_batch * g.batch;
    struct Work {
        const float* A; const float* B; float* C;
        int m0, n0, bz, ks, kbeg, kend, tn_idx, kshift, xy;
    };
    auto decode = [&](long w) {
        Work x;
        x.bz = (int)(w / per_batch);
        const int wrem = (int)(w - x.bz * per_batch);
        x.ks = wrem / per_split;
        int xy = wrem - x.ks * per_split;
        if (!EXT && g.xcd_chunks) {   // chunk = XCD + 8 * round, tile = position inside the XCD's share of the work list
            const int q = wrem >> 3, rnd = q / per_split;
            x.ks = (wrem & 7) + 8 * rnd;
            xy = q - rnd * per_split;
        }
        if (g.xcd_panels) {
            const long panels = (long)g.batch * tiles_m, p8 = panels & ~7L, cut = p8 * tiles_n;
            long panel;
            int tn;
            if (w < cut) {
                const long q = w >> 3, rnd = q / tiles_n;
                tn = (int)(q - rnd * tiles_n);
                panel = (w & 7) + 8 * rnd;
            } else {      // the last (panels % 8) panels, plainly
                const long r = w - cut;
                panel = p8 + r / tiles_n;
                tn = (int)(r % tiles_n);
            }
            x.bz = (int)(panel / tiles_m);
            x.ks = 0;
            xy = tn * tiles_m + (int)(panel - (long)x.bz * tiles_m);
        }
        x.xy = xy;
        x.tn_idx = xy / tiles_m;
        if constexpr (EXT) {
            int tm = xy - x.tn_idx * tiles_m;
            // triangular A: an M-tile's reduction length depends on its position, and a workgroup that strides the work list by
            // a multiple of the tiles per batch member would always get the same position (the longest one sets the launch
            // time): rotate the M-tile by the round, so that every workgroup sees every length.  (A bijection on a batch
            // member's tiles as long as they all sit in one round: grid % tiles per batch member == 0, else no rotation.)
            if (g.k_tri != 0 && tiles_n == 1 && g.splitk == 1 && gridDim.x % tiles_m == 0) tm = (tm + (int)(w / gridDim.x)) % tiles_m;
            x.m0 = tm * BM;
        } else {
            x.m0 = (xy - x.tn_idx * tiles_m) * BM;
        }
        x.n0 = x.tn_idx * BN;
        x.A = g.A + (g.a_off ? g.a_off[x.bz] : (long)x.bz * g.a_batch);
        x.B = g.B + (g.b_off ? g.b_off[x.bz] : (long)x.bz * g.b_batch);
        x.C = g.C + (g.c_off ? g.c_off[x.bz] : (long)x.bz * g.c_batch);
        x.kbeg = x.ks * g.kchunk;
        x.kend = min(g.K, x.kbeg + g.kchunk);
        // triangular A (causal attention probabilities and their gradients): the k-tiles in which every row of this tile is zero
        // are not visited -- the skipped products are exact zeros
        // (extended instantiation only: in the plain one the few extra scalar instructions per tile cost the BiGRU step's short
        // 64 x 64 launches 8-13 %)
        if constexpr (EXT) {
            if (g.k_tri == 1) x.kbeg = max(x.kbeg, (x.m0 / BK) * BK);
            else if (g.k_tri == 2) x.kend = min(x.kend, ((x.m0 + BM + BK - 1) / BK) * BK);
        }
        x.kshift = g.b_kshift + x.bz * g.b_kshift_batch;
        return x;
    };
    long work = blockIdx.x;
    if (work >= total) return;
    Work w = decode(work);
    float4 ra[BM / 32], rb[BN / 32];
    // operand tiles of work item x at reduction index k0 -> registers.  Segmented reduction: the segment's own A rows / B
    // panel, k counted from the segment's start (k_seg is a multiple of BK: a k-tile never straddles two segments).
    auto load_tiles = [&](const Work& x, int k0) {
        const float* Ap = x.A;
        const float* Bp = x.B;
        int kl = k0, ke = x.kend;
        if (EXT && g.k_seg > 0) {
            const int seg = k0 / g.k_seg;
            const long si = (long)x.bz * (g.K / g.k_seg) + seg;
            Ap = g.A + g.a_seg_off[si];
            Bp = g.B + g.b_seg_off[si];
            kl = k0 - seg * g.k_seg;
            ke = g.k_seg;
        }
        tile_load<BM, A_KC, FAST>(ra, Ap, g.a_i, g.a_k, x.m0, kl, g.M, ke, g.a_vec, 0, 0, tid);
        tile_load<BN, B_KC, FAST>(rb, Bp, g.b_j, g.b_k, x.n0, kl, g.N, ke, g.b_vec, x.kshift, g.b_kT, tid);
    };
    load_tiles(w, w.kbeg);
    for (;;) {
        float* C = w.C;
        const int m0 = w.m0, n0 = w.n0, bz = w.bz, kbeg = w.kbeg, kend = w.kend;
        f32x16 acc[TM][TN];
        unsigned mw[TM][TN];   // EXT: the ReLU-mask words of this lane's rows (lane l31 & 15 holds the word of accumulator row r = l31 & 15)
        if (EXT && g.res != nullptr) {
            // the residual is the accumulators' initial value: 64 loads into the registers the products will be added to, in
            // flight while the operand tiles are staged through LDS -- no extra registers, no epilogue loads (an epilogue
            // that fetches it has to interleave loads with its stores, and vector memory operations retire in order:
            // measured 1281-1393 us against 945 for the 110 x [6400 x 256 x 256] launch, 834 without a residual).
            // 32-bit element offsets from a wave-uniform base (the host checks the extents): an address is one VGPR
            const float* res = g.res + (g.res_off ? g.res_off[bz] : (long)bz * g.res_batch);
            const unsigned res_ld = (unsigned)g.res_ld;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const unsigned colc = (unsigned)min(n0 + wn * WN + j * 32 + l31, g.N - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned row = (unsigned)min(m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
                        acc[i][j][r] = res[row * res_ld + colc];
                    }
                }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        const unsigned ncb = (unsigned)(g.N + 31) >> 5;   // mask / ReLU-bit words per row
        if (EXT && g.mask_bits != nullptr) {
            const unsigned* mb = g.mask_bits + (long)bz * g.mask_batch;
            const int r = l31 & 15;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const unsigned row = (unsigned)min(m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
                    const unsigned cb = min((unsigned)(n0 + wn * WN + j * 32) >> 5, ncb - 1);
                    mw[i][j] = mb[row * ncb + cb];
                }
        }
        const int nk = (kend - kbeg + BK - 1) / BK;
        const bool do_cs = !EXT && !A_KC && g.colsum != nullptr && w.tn_idx == 0 && tid < BM;
        float cs_acc = 0.f;
        __syncthreads();  // the previous tile's fragment reads are done
        tile_store<BM, A_KC>(sA[0], ra, tid);
        tile_store<BN, B_KC>(sB[0], rb, tid);
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tiles(w, kbeg + (kt + 1) * BK);
            const float* a_s = sA[0];
            const float* b_s = sB[0];
            if (do_cs) {  // image [k][BM]: consecutive threads read consecutive words (zero padded beyond M / kend)
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) cs_acc += a_s[kk * BM + tid];
            }
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = frag<BM, A_KC>(a_s, wm * WM + i * 32 + l31, kk + lh);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = frag<BN, B_KC>(b_s, wn * WN + j * 32 + l31, kk + lh);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nk) {
                __syncthreads();  // every wave is done reading the image
                tile_store<BM, A_KC>(sA[0], ra, tid);
                tile_store<BN, B_KC>(sB[0], rb, tid);
                __syncthreads();
            }
        }

        // the next tile's first operand tiles go in flight ahead of this tile's stores
        work += gridDim.x;
        const bool more = work < total;
        const int ks = w.ks, xy = w.xy;
        if (more) {
            w = decode(work);
            load_tiles(w, w.kbeg);
        }

        // epilogue: D[i][j], j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const bool whole = m0 + BM <= g.M;  // workgroup-uniform: whole rows, only a per-lane column predicate
        bool final_store = true;
        // 64x64 tiles only: in the larger tiles the summing loop costs the main loop its registers (scratch spills)
        const bool fix = !EXT && BM * BN <= 64 * 64 && g.counters != nullptr;
        if (!EXT && g.splitk > 1) {
            const int ncs = g.colsum ? 1 : 0;
            const long W = g.N + ncs;
            const long slab_k = (long)g.batch * g.M * W;  // one split's slab
            float* slab = g.slab + (long)ks * slab_k + (long)bz * g.M * W;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WN + j * 32 + l31;
                if (col >= g.N) continue;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (row < g.M) {
                            if (fix) st_agent(&slab[(long)row * W + col], acc[i][j][r]);
                            else slab[(long)row * W + col] = acc[i][j][r];
                        }
                    }
            }
            if (do_cs && m0 + tid < g.M) {
                if (fix) st_agent(&slab[(long)(m0 + tid) * W + g.N], cs_acc);
                else slab[(long)(m0 + tid) * W + g.N] = cs_acc;
            }
            final_store = false;
            if (fix) {
                // The partial went out as agent-scope (write-through) stores and is read back by agent-scope loads, which do
                // not hit in another XCD's L2; waiting for the stores' acknowledgement orders them before the arrival count.
                // A __threadfence() here would write back the whole L2 per workgroup: measured 2.4 vs 1.45 ms per step.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {   // ... before its arrival is counted
                    int* cnt = g.counters + (long)bz * per_split + xy;
                    const int last = atomicAdd(cnt, 1) == g.splitk - 1;
                    // every partial of the tile has arrived: nobody touches the counter again this launch
                    if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_last = last;
                }
                __syncthreads();
                if (s_last) {
                    const float* sl = g.slab + (long)bz * g.M * W;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                    cs_acc = 0.f;
                    for (int kk = 0; kk < g.splitk; ++kk, sl += slab_k) {  // k = 0, 1, ...: the order of splitk_reduce_kernel
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int col = min(n0 + wn * WN + j * 32 + l31, g.N - 1);
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int row = min(m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
                                    acc[i][j][r] += ld_agent(&sl[(long)row * W + col]);
                                }
                        }
                        if (do_cs) cs_acc += ld_agent(&sl[(long)min(m0 + tid, g.M - 1) * W + g.N]);
                    }
                    final_store = true;
                }
            }
        }
        if (final_store) {
            if (do_cs && m0 + tid < g.M) g.colsum[(long)bz * g.colsum_batch + m0 + tid] = cs_acc;
            const float* bias = g.bias ? g.bias + (g.bias_off ? g.bias_off[bz] : (long)bz * g.bias_batch) : nullptr;
            if (EXT) {   // (the one epilogue of the extended instantiation: bias, ReLU + its bit image, ReLU mask)
                const unsigned ldc = (unsigned)g.ldc;
                unsigned* rb_out = g.relu_bits ? g.relu_bits + (long)bz * g.relu_bits_batch : nullptr;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * WN + j * 32 + l31;
                    const bool col_ok = col < g.N;
                    const float bj = bias ? bias[min(col, g.N - 1)] : 0.f;
                    const unsigned cb = (unsigned)(n0 + wn * WN + j * 32) >> 5;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row0 = m0 + wm * WM + i * 32 + 4 * lh;
                        unsigned keep_word = 0u;    // ReLU bits: lane l31 == r keeps the word of accumulator row r
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = row0 + (r & 3) + 8 * (r >> 2);
                            float v = acc[i][j][r] + bj;
                            if (g.act == 1) v = as_relu(v);
                            if (rb_out != nullptr) {
                                const unsigned long long pos = __ballot(v > 0.f && col_ok);
                                const unsigned word = lh ? (unsigned)(pos >> 32) : (unsigned)pos;
                                keep_word = l31 == r ? word : keep_word;
                            }
                            if (g.mask_bits != nullptr) {
                                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)mw[i][j], r);
                                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)mw[i][j], 32 + r);
                                if (!(((lh ? hi : lo) >> l31) & 1u)) v = 0.f;
                            }
                            if (row < g.M && col_ok) C[(unsigned)row * ldc + (unsigned)col] = v;
                        }
                        if (rb_out != nullptr && l31 < 16) {
                            const int row = row0 + (l31 & 3) + 8 * (l31 >> 2);
                            if (row < g.M && cb < ncb) rb_out[(unsigned)row * ncb + cb] = keep_word;
                        }
                    }
                }
            } else if (whole && !g.accumulate) {
                float* c0 = C + (long)(m0 + wm * WM + 4 * lh) * g.ldc + n0 + wn * WN + l31;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * WN + j * 32 + l31;
                    if (col < g.N) {  // no loads inside: the stores below are issued back to back under the lane mask
                        const float bj = bias ? bias[col] : 0.f;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float v = acc[i][j][r] + bj;
                                if (g.act == 1) v = as_relu(v);
                                else if (g.act == 2) v = as_sigmoid(v);
                                else if (g.act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                                c0[(long)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + j * 32] = v;
                            }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * WN + j * 32 + l31;
                    if (col >= g.N) continue;
                    const float bj = bias ? bias[col] : 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            if (row >= g.M) continue;
                            float v = acc[i][j][r] + bj;
                            if (g.act == 1) v = as_relu(v);
                            else if (g.act == 2) v = as_sigmoid(v);
                            else if (g.act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                            float* c = C + (long)row * g.ldc + col;
                            if (g.accumulate) v += *c;
                            *c = v;
                        }
                    }
                }
            }
        }
        if (!more) break;
    }
}

// ---- split-precision variant for forward linears (both operands reduction-contiguous, "NT") -----------------------------
// Every fp32 operand element is split on the fly into PL bf16 pieces (x = h + m [+ l], each piece the bf16 rounding of what
// is left) and the product is rebuilt from the significant cross terms on the bf16 matrix core (16x the fp32 MFMA rate)
// with fp32 accumulation:  PL = 3: hh + hm + mh + hl + lh + mm (6 MFMAs, drops terms below 2^-24 of |a||b|: fp32-grade);
// PL = 2: hh + hm + mh (3 MFMAs, ~2^-16).  LDS image per piece: [row][40 bf16] (80-byte rows: conflict-free ds_read_b128 of
// the 8-element k groups a lane of v_mfma_f32_32x32x16_bf16 owns).  Same persistent tile walk and epilogue as the fp32 kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int PL>
__device__ __forceinline__ void split_store(__bf16* __restrict__ img, int plane_elems, int off, const float4& v) {
    float x[4] = {v.x, v.y, v.z, v.w};
    bf16x4 piece[PL];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float rest = x[e];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
            const __bf16 q = (__bf16)rest;
            piece[p][e] = q;
            rest -= (float)q;
        }
    }
#pragma unroll
    for (int p = 0; p < PL; ++p) *reinterpret_cast<bf16x4*>(img + p * plane_elems + off) = piece[p];
}

template <int BM, int BN, int PL>
__global__ __launch_bounds__(256, 2) void gemm_split_nt_kernel(GemmK g) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int LDK = BK + 8;  // bf16 elements per image row (80 bytes)
    __shared__ __attribute__((aligned(16))) __bf16 sA[PL * BM * LDK];
    __shared__ __attribute__((aligned(16))) __bf16 sB[PL * BN * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int per_batch = tiles_m * tiles_n;
    const long total = (long)per_batch * g.batch;
    const int kq = tid & 7, row0 = tid >> 3;  // staging: this thread's float4 column and first row
    struct Work { const float* A; const float* B; float* C; int m0, n0, bz; };
    auto decode = [&](long w) {
        Work x;
        x.bz = (int)(w / per_batch);
        const int xy = (int)(w - (long)x.bz * per_batch);
        const int tn = xy / tiles_m;
        x.m0 = (xy - tn * tiles_m) * BM;
        x.n0 = tn * BN;
        x.A = g.A + (g.a_off ? g.a_off[x.bz] : (long)x.bz * g.a_batch);
        x.B = g.B + (g.b_off ? g.b_off[x.bz] : (long)x.bz * g.b_batch);
        x.C = g.C + (g.c_off ? g.c_off[x.bz] : (long)x.bz * g.c_batch);
        return x;
    };
    long work = blockIdx.x;
    if (work >= total) return;
    Work w = decode(work);
    float4 ra[BM / 32], rb[BN / 32];
    tile_load<BM, true, true>(ra, w.A, g.a_i, 1, w.m0, 0, g.M, g.K, true, 0, 0, tid);
    tile_load<BN, true, true>(rb, w.B, g.b_j, 1, w.n0, 0, g.N, g.K, true, 0, 0, tid);
    const int nk = (g.K + BK - 1) / BK;
    for (;;) {
        const float* A = w.A;
        const float* B = w.B;
        float* C = w.C;
        const int m0 = w.m0, n0 = w.n0, bz = w.bz;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();  // every wave is done reading the previous image
#pragma unroll
            for (int pp = 0; pp < BM / 32; ++pp) split_store<PL>(sA, BM * LDK, (row0 + 32 * pp) * LDK + kq * 4, ra[pp]);
#pragma unroll
            for (int pp = 0; pp < BN / 32; ++pp) split_store<PL>(sB, BN * LDK, (row0 + 32 * pp) * LDK + kq * 4, rb[pp]);
            __syncthreads();
            if (kt + 1 < nk) {
                tile_load<BM, true, true>(ra, A, g.a_i, 1, m0, (kt + 1) * BK, g.M, g.K, true, 0, 0, tid);
                tile_load<BN, true, true>(rb, B, g.b_j, 1, n0, (kt + 1) * BK, g.N, g.K, true, 0, 0, tid);
            }
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 af[TM][PL], bf[TN][PL];
#pragma unroll
                for (int p = 0; p < PL; ++p) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[i][p] = *reinterpret_cast<const bf16x8*>(sA + p * BM * LDK + (wm * WM + i * 32 + l31) * LDK + ks * 16 + 8 * lh);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[j][p] = *reinterpret_cast<const bf16x8*>(sB + p * BN * LDK + (wn * WN + j * 32 + l31) * LDK + ks * 16 + 8 * lh);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 c = acc[i][j];
                        if (PL == 3) {  // smallest terms first
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][PL - 1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PL - 1], bf[j][0], c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
        }
        // the next tile's first operand tiles go in flight ahead of this tile's stores
        work += gridDim.x;
        const bool more = work < total;
        if (more) {
            w = decode(work);
            tile_load<BM, true, true>(ra, w.A, g.a_i, 1, w.m0, 0, g.M, g.K, true, 0, 0, tid);
            tile_load<BN, true, true>(rb, w.B, g.b_j, 1, w.n0, 0, g.N, g.K, true, 0, 0, tid);
        }
        const float* bias = g.bias ? g.bias + (g.bias_off ? g.bias_off[bz] : (long)bz * g.bias_batch) : nullptr;
        const bool whole = m0 + BM <= g.M;
        float* c0 = C + (long)(m0 + wm * WM + 4 * lh) * g.ldc + n0 + wn * WN + l31;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            if (col >= g.N) continue;
            const float bj = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
                    if (!whole && m0 + wm * WM + 4 * lh + rr >= g.M) continue;
                    float v = acc[i][j][r] + bj;
                    if (g.act == 1) v = as_relu(v);
                    else if (g.act == 2) v = as_sigmoid(v);
                    else if (g.act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    float* c = c0 + (long)rr * g.ldc + j * 32;
                    if (g.accumulate) v += *c;
                    *c = v;
                }
        }
        if (!more) break;
    }
}

// C (+)= sum over splits of the slab; the extra slab column (if any) is the fused column sum
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmK g) {
    const int ncs = g.colsum ? 1 : 0;
    const long W = g.N + ncs;
    const long per = (long)g.M * W;
    const long total = (long)g.batch * per;
    const long stride = (long)gridDim.x * 256;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        float s = 0.f;
        for (int k = 0; k < g.splitk; ++k) s += g.slab[(long)k * total + idx];
        const long bz = idx / per, rem = idx - bz * per;
        const long row = rem / W;
        const int col = (int)(rem - row * W);
        if (col < g.N) {
            float* c = g.C + bz * g.c_batch + row * g.ldc + col;
            *c = g.accumulate ? *c + s : s;
        } else {
            g.colsum[bz * g.colsum_batch + row] = s;
        }
    }
}

// Same sum for MANY slabs over FEW outputs (dW_hh: 50 slabs of 49 536 floats): a thread per output would walk the slabs
// alone.  Four quarter-workgroups each take every 4th slab of 64 consecutive outputs and the partials meet in LDS in a fixed
// order (deterministic).  The k order differs from splitk_reduce_kernel's, so one shape always takes the same kernel.
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(GemmK g) {
    __shared__ float part[3][64];
    const int ncs = g.colsum ? 1 : 0;
    const long W = g.N + ncs;
    const long per = (long)g.M * W;
    const long total = (long)g.batch * per;
    const int q = threadIdx.x >> 6, i = threadIdx.x & 63;
    const long idx = (long)blockIdx.x * 64 + i;
    const long src = idx < total ? idx : total - 1;
    float s = 0.f;
    for (int k = q; k < g.splitk; k += 4) s += g.slab[(long)k * total + src];
    if (q > 0) part[q - 1][i] = s;
    __syncthreads();
    if (q > 0 || idx >= total) return;
    s = ((s + part[0][i]) + part[1][i]) + part[2][i];
    const long bz = idx / per, rem = idx - bz * per;
    const long row = rem / W;
    const int col = (int)(rem - row * W);
    if (col < g.N) {
        float* c = g.C + bz * g.c_batch + row * g.ldc + col;
        *c = g.accumulate ? *c + s : s;
    } else {
        g.colsum[bz * g.colsum_batch + row] = s;
    }
}

// resident workgroups of one kernel instance on the whole device (queried once per instance)
template <typename Kern>
int resident_blocks(Kern kern) {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    return per_cu * cus;
}

// Arrival counters for the in-kernel split-K reduction: one zeroed array per (device, stream), created on first use (like
// the side stream: never inside a stream capture) and left zero by every launch.  nullptr = fall back to the reduce kernel.
constexpr int COUNTERS = 8192;
int* counters_for(hipStream_t st) {
    struct Slot { int dev; hipStream_t st; int* p; };
    static std::mutex mu;
    static std::vector<Slot> slots;
    static const bool off = AS_DIAG_SET("AS_GEMM_NO_FIXUP");  // ablation: always the separate reduce kernel
    if (off) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (const Slot& s : slots)
        if (s.dev == dev && s.st == st) return s.p;
    if (slots.size() >= 64) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;  // no allocation inside a stream capture: the reduce kernel then
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    int* p = nullptr;
    if (hipMalloc(&p, COUNTERS * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemsetAsync(p, 0, COUNTERS * sizeof(int), st) != hipSuccess) { (void)hipFree(p); return nullptr; }
    slots.push_back({dev, st, p});
    return p;
}

}  // namespace
// see gemm_internal.h: the last word of the stream's counter block (the split-K kernel uses the first tiles-per-launch words; all
// users run on the stream in order and leave their counters zero)
int* as_arrival_counter(hipStream_t st) {
    int* p = counters_for(st);
    return p ? p + (COUNTERS - 1) : nullptr;
}
namespace {

template <int BM, int BN>
int launch(const GemmK& k, int batch, bool a_kc, bool b_kc, hipStream_t st) {
    const long work = (long)as_cdiv(k.M, BM) * as_cdiv(k.N, BN) * k.splitk * batch;
    dim3 block(256);
    // FAST needs whole float4s: aligned operands and contiguous extents that are multiples of 4
    const bool fast = k.a_vec && k.b_vec && (a_kc ? k.K % 4 == 0 : k.M % 4 == 0) && (b_kc ? k.K % 4 == 0 : k.N % 4 == 0);
    // the call's last kernel (no reduce kernel behind it) may carry a fork event (gemm_internal.h, as_stop_event_set)
    hipEvent_t stop_ev = (k.splitk == 1 || k.counters != nullptr) ? as_stop_event_take() : nullptr;
#define AS_GEMM_LAUNCH(AK, BK_, F)                                                                   \
    do {                                                                                             \
        static const int slots = resident_blocks(gemm_f32_kernel<BM, BN, AK, BK_, F>);               \
        const dim3 grid((unsigned)(work < slots ? work : slots));                                    \
        if (stop_ev) hipExtLaunchKernelGGL((gemm_f32_kernel<BM, BN, AK, BK_, F>), grid, block, 0, st, nullptr, stop_ev, 0, k); \
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AK, BK_, F>), grid, block, 0, st, k);       \
    } while (0)
    if (k.res || k.mask_bits || k.relu_bits || k.k_seg > 0 || k.k_tri != 0) {   // the extended instantiations (operands checked by as_gemm_f32)
#define AS_GEMM_LAUNCH_EXT(AK, BK_)                                                                    \
    do {                                                                                               \
        static const int slots = resident_blocks(gemm_f32_kernel<BM, BN, AK, BK_, true, true>);        \
        const dim3 grid((unsigned)(work < slots ? work : slots));                                      \
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AK, BK_, true, true>), grid, block, 0, st, k);     \
    } while (0)
        if constexpr (BM == BN) {   // 128 x 128 and 64 x 64 only
            if (a_kc && b_kc) AS_GEMM_LAUNCH_EXT(true, true);
            else if (a_kc) AS_GEMM_LAUNCH_EXT(true, false);
            else if (!b_kc) AS_GEMM_LAUNCH_EXT(false, false);   // (k_tri = 2: dQ = dS K reads dS^T through its transpose)
            else AS_REQUIRE(false, AS_ERR_BAD_ARG, "as_gemm_f32: the extended operands have no kernel for a strided A with a contiguous B");
        } else {
            AS_REQUIRE(false, AS_ERR_BAD_ARG, "as_gemm_f32: the extended operands need a square tile");
        }
#undef AS_GEMM_LAUNCH_EXT
    } else if (fast) {
        if (a_kc && b_kc) AS_GEMM_LAUNCH(true, true, true);
        else if (a_kc && !b_kc) AS_GEMM_LAUNCH(true, false, true);
        else if (!a_kc && b_kc) AS_GEMM_LAUNCH(false, true, true);
        else AS_GEMM_LAUNCH(false, false, true);
    } else {
        if (a_kc && b_kc) AS_GEMM_LAUNCH(true, true, false);
        else if (a_kc && !b_kc) AS_GEMM_LAUNCH(true, false, false);
        else if (!a_kc && b_kc) AS_GEMM_LAUNCH(false, true, false);
        else AS_GEMM_LAUNCH(false, false, false);
    }
#undef AS_GEMM_LAUNCH
    AS_LAUNCH_CHECK("as_gemm_f32");
    return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int as_gemm_f32(const as_gemm* g, void* stream) {
    AS_REQUIRE(g && g->A && g->B && g->C, AS_ERR_BAD_ARG, "as_gemm_f32: null pointer");
    AS_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0 && g->batch > 0, AS_ERR_BAD_ARG,
               "as_gemm_f32: non-positive size M=%d N=%d K=%d batch=%d", g->M, g->N, g->K, g->batch);
    AS_REQUIRE((g->a_i == 1) != (g->a_k == 1) || (g->a_i == 1 && g->a_k == 1 && (g->M == 1 || g->K == 1)),
               AS_ERR_BAD_ARG, "as_gemm_f32: exactly one of a_i/a_k must be 1 (a_i=%ld a_k=%ld)", (long)g->a_i, (long)g->a_k);
    AS_REQUIRE((g->b_j == 1) != (g->b_k == 1) || (g->b_j == 1 && g->b_k == 1 && (g->N == 1 || g->K == 1)),
               AS_ERR_BAD_ARG, "as_gemm_f32: exactly one of b_j/b_k must be 1 (b_j=%ld b_k=%ld)", (long)g->b_j, (long)g->b_k);
    AS_REQUIRE(g->act >= 0 && g->act <= 3, AS_ERR_BAD_ARG, "as_gemm_f32: act=%d", g->act);
    AS_REQUIRE(g->precision >= 0 && g->precision <= 3, AS_ERR_BAD_ARG, "as_gemm_f32: precision=%d", g->precision);
    const int prec = g->precision == 3 ? 0 : g->precision;   // 3 = "the library's matrix arithmetic, at any size" (below)
    // (M == 1 with both strides of A equal to 1 reads the same either way: output-contiguous then, the form the column sums take)
    const bool a_kc = g->a_k == 1 && !(g->a_i == 1 && g->colsum), b_kc = g->b_k == 1;
    AS_REQUIRE(!(g->b_kT > 0 && b_kc), AS_ERR_BAD_ARG, "as_gemm_f32: b_kshift needs a reduction-strided B operand");
    if (!a_kc && !b_kc && g->k_tri == 0) {  // weight-gradient shapes: the kernel of wgrad_f32.hip (it does not know k_tri)
        // precision == 3 in the split arithmetic: one workgroup per 128 x 128 output tile over the whole reduction, operand tiles
        // staged once per workgroup (gemm_s6.hip; the transformer's grouped weight gradients, 110 x [256 x 256 x 6400]: 704 us
        // with the stream-K kernel below, 597 - 659 us there; it declines launches of fewer than 256 tiles)
        if (g->precision == 3 && g->a_i == 1 && g->b_j == 1) {
            const int took = as_gemm_s6_nt_ext(g, (hipStream_t)stream);
            if (took != 0) return took < 0 ? took : 0;
        }
        const int taken = as_wgrad_try(g, (hipStream_t)stream);
        if (taken != 0) return taken < 0 ? taken : 0;
    }
    GemmK k;
    k.A = g->A; k.B = g->B; k.C = g->C; k.bias = g->bias;
    k.M = g->M; k.N = g->N; k.K = g->K;
    k.a_i = g->a_i; k.a_k = g->a_k; k.b_j = g->b_j; k.b_k = g->b_k; k.ldc = g->ldc;
    k.a_batch = g->a_batch; k.b_batch = g->b_batch; k.c_batch = g->c_batch; k.bias_batch = g->bias_batch;
    k.act = g->act; k.accumulate = g->accumulate; k.b_kshift = g->b_kshift; k.b_kT = g->b_kT; k.b_kshift_batch = g->b_kshift_batch;
    const long a_ld = a_kc ? g->a_i : g->a_k, b_ld = b_kc ? g->b_j : g->b_k;
    k.a_vec = aligned16(g->A) && a_ld % 4 == 0 && g->a_batch % 4 == 0;
    k.b_vec = aligned16(g->B) && b_ld % 4 == 0 && g->b_batch % 4 == 0;
    hipStream_t st = (hipStream_t)stream;
    k.splitk = 1; k.kchunk = g->K; k.batch = g->batch; k.slab = nullptr; k.counters = nullptr; k.xcd_chunks = 0; k.xcd_panels = 0;
    k.colsum = g->colsum; k.colsum_batch = g->colsum_batch;
    k.a_off = (const long*)g->a_off; k.b_off = (const long*)g->b_off; k.c_off = (const long*)g->c_off;
    k.bias_off = (const long*)g->bias_off;
    k.res = g->res; k.res_ld = g->res_ld; k.res_batch = g->res_batch; k.res_off = (const long*)g->res_off;
    k.mask_bits = g->mask_bits; k.mask_batch = g->mask_batch; k.relu_bits = g->relu_bits; k.relu_bits_batch = g->relu_bits_batch;
    k.k_seg = g->k_seg; k.a_seg_off = (const long*)g->a_seg_off; k.b_seg_off = (const long*)g->b_seg_off;
    k.k_tri = g->k_tri;
    AS_REQUIRE(g->k_tri >= 0 && g->k_tri <= 2, AS_ERR_BAD_ARG, "as_gemm_f32: k_tri=%d", g->k_tri);
    AS_REQUIRE(g->k_tri == 0 || (!g->colsum && !g->splitk_ws && prec == 0 && g->k_seg == 0 && !g->accumulate && g->act <= 1 &&
                                 !g->bias_off && (long)g->M * g->ldc < (1L << 31)),
               AS_ERR_BAD_ARG, "as_gemm_f32: k_tri goes with the extended general kernel only (no colsum, splitk_ws, split precision, "
               "k_seg, accumulate, act > 1)");
    // k_tri is a hint about zeros: operands the extended (float4) instantiation cannot take run the plain kernel over the full range
    if (g->k_tri != 0 && !(aligned16(g->A) && aligned16(g->B) && a_ld % 4 == 0 && b_ld % 4 == 0 && (a_kc ? g->K % 4 == 0 : g->M % 4 == 0) &&
                           (b_kc ? g->K % 4 == 0 : g->N % 4 == 0) && (g->a_off || g->a_batch % 4 == 0) && (g->b_off || g->b_batch % 4 == 0) &&
                           (a_kc || !b_kc)))
        k.k_tri = 0;
    const bool epi_ops = g->res || g->mask_bits || g->relu_bits, segmented = g->k_seg > 0;
    AS_REQUIRE(!(epi_ops || segmented) || (!g->colsum && !g->splitk_ws && !g->accumulate && prec == 0 && (a_kc || b_kc)),
               AS_ERR_BAD_ARG, "as_gemm_f32: res / mask_bits / relu_bits / k_seg go with the general kernel only (no colsum, splitk_ws, "
               "accumulate, split precision or weight-gradient shape)");
    AS_REQUIRE(!g->relu_bits || g->act == 1, AS_ERR_BAD_ARG, "as_gemm_f32: relu_bits is the bit image of a ReLU epilogue (act == 1)");
    AS_REQUIRE(!(epi_ops || segmented) || (a_kc && aligned16(g->A) && aligned16(g->B) && a_ld % 4 == 0 && b_ld % 4 == 0 && g->K % 4 == 0 &&
                                           (b_kc || g->N % 4 == 0) && g->act <= 1 && (g->a_off || segmented || g->a_batch % 4 == 0) &&
                                           (g->b_off || segmented || g->b_batch % 4 == 0)),
               AS_ERR_BAD_ARG, "as_gemm_f32: res / mask_bits / relu_bits / k_seg need a reduction-contiguous A, float4-clean operands "
               "and act <= 1");
    AS_REQUIRE(!(epi_ops || segmented) || ((long)g->M * g->ldc < (1L << 31) && (long)g->M * g->res_ld < (1L << 31)),
               AS_ERR_BAD_ARG, "as_gemm_f32: res / mask_bits / relu_bits / k_seg address one batch member's C and res with 32-bit offsets");
    AS_REQUIRE(!segmented || (g->k_seg % BK == 0 && g->K % g->k_seg == 0 && g->a_seg_off && g->b_seg_off && g->b_kT == 0),
               AS_ERR_BAD_ARG, "as_gemm_f32: k_seg=%d needs a multiple of %d that divides K=%d and both segment tables", g->k_seg, BK, g->K);
    const bool grouped = g->a_off || g->b_off || g->c_off || g->bias_off;
    if (grouped || segmented) {  // alignment of table offsets is the caller's contract (multiples of 4 floats) -- see header
        k.a_vec = aligned16(g->A) && a_ld % 4 == 0;
        k.b_vec = aligned16(g->B) && b_ld % 4 == 0;
    }
    AS_REQUIRE(!(g->colsum && a_kc), AS_ERR_BAD_ARG, "as_gemm_f32: colsum needs an output-contiguous A operand (a_i == 1)");
    // precision == 3: the library's split matrix arithmetic (as_set_matrix_arith(1)) for forward shapes -- both operands
    // reduction-contiguous, plain or ReLU-bit epilogue, linear or grouped batches -- on the bf16 matrix instruction with both
    // operands split inside the kernel (gemm_s6.hip: 1.36 x this kernel on the transformer's block groups, 110 x [6400 x 256 x
    // 256]).  Opt-in per call and independent of the launch's size: a size threshold would make the last bits of a result
    // depend on the batch it was computed in (measured: 7e-5 on the transformer's contours between batches of 4 and 32).
    // The input-gradient orientation (B column-contiguous, with res / mask_bits / k_seg) goes the same way.
    if (a_kc && (b_kc || g->b_j == 1) && g->precision == 3) {
        const int took = as_gemm_s6_nt_ext(g, st);
        if (took != 0) return took < 0 ? took : 0;
    }
    // split-precision request (forward linears only): both operands reduction-contiguous and float4-clean, else exact fp32
    if (prec != 0 && a_kc && b_kc && k.a_vec && k.b_vec && g->K % 4 == 0 && !g->colsum) {
        AS_REQUIRE(g->precision == 1 || g->precision == 2, AS_ERR_BAD_ARG, "as_gemm_f32: precision=%d", g->precision);
        const long work = (long)as_cdiv(g->M, 128) * as_cdiv(g->N, 128) * g->batch;
        if (prec == 2) {
            static const int slots = resident_blocks(gemm_split_nt_kernel<128, 128, 3>);
            hipLaunchKernelGGL((gemm_split_nt_kernel<128, 128, 3>), dim3((unsigned)(work < slots ? work : slots)), dim3(256), 0, st, k);
        } else {
            static const int slots = resident_blocks(gemm_split_nt_kernel<128, 128, 2>);
            hipLaunchKernelGGL((gemm_split_nt_kernel<128, 128, 2>), dim3((unsigned)(work < slots ? work : slots)), dim3(256), 0, st, k);
        }
        AS_LAUNCH_CHECK("as_gemm_f32(split)");
        return 0;
    }
    // 128x128 tiles once they fill the chip and N fills a tile (N = 100: 60 vs 72 us at 64x64), else 64x64 for more workgroups
    const long big = (long)as_cdiv(g->M, 128) * as_cdiv(g->N, 128) * g->batch;
    static const char* force = AS_DIAG_STR("AS_GEMM_TILE");  // tuning aid: "128x128" | "64x128" | "128x64" | "64x64"
    if (force && !strcmp(force, "128x128")) return launch<128, 128>(k, g->batch, a_kc, b_kc, st);
    if (force && !strcmp(force, "64x128")) return launch<64, 128>(k, g->batch, a_kc, b_kc, st);
    if (force && !strcmp(force, "128x64")) return launch<128, 64>(k, g->batch, a_kc, b_kc, st);
    if (!force && big >= 512 && g->N >= 128) {  // fewer 128x128 tiles leave most of the 768 resident slots empty: 64x64 then
                                                 // (measured 6400 x 768 x 256: 300 tiles 39.9 us, as 1200 64x64 tiles 30.1 us)
        // 128x128 tiles that do not fill the resident slots (3 per CU) with a long reduction: split K so that the persistent
        // workgroups get equal shares (measured: 440 tiles, K = 6400 run at 75 TF/s, 768 tiles of the same shape at 100)
        static const int slots = resident_blocks(gemm_f32_kernel<128, 128, true, true, true>);
        if (g->splitk_ws && !grouped && big < slots && g->K >= 2048 && !g->bias && g->act == 0) {
            const long per = (long)g->batch * g->M * (g->N + (g->colsum ? 1 : 0));
            long best = 1;
            double best_cost = 1.0;  // rounds of work per workgroup, in units of the unsplit tile time
            for (long sk = 2; sk <= 8 && sk * per <= g->splitk_ws_floats; ++sk) {
                const double cost = (double)((big * sk + slots - 1) / slots) / sk + 0.02 * sk;  // + slab traffic
                if (cost < best_cost - 1e-9) best_cost = cost, best = sk;
            }
            if (best > 1) {
                k.kchunk = (int)as_round_up(as_cdiv(g->K, best), BK);
                k.splitk = as_cdiv(g->K, k.kchunk);
                k.slab = g->splitk_ws;
                AS_TRY((launch<128, 128>(k, g->batch, a_kc, b_kc, st)));
                const long total = per;
                long blocks = (total + 255) / 256;
                if (blocks > 2048) blocks = 2048;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, st, k);
                AS_LAUNCH_CHECK("as_gemm_f32(splitk reduce)");
                return 0;
            }
        }
        static const bool no_panels = AS_DIAG_SET("AS_NO_XCD_PANELS");
        k.xcd_panels = a_kc && g->N > 128 && g->k_tri == 0 && big >= 2048 && slots % 8 == 0 && !no_panels;
        return launch<128, 128>(k, g->batch, a_kc, b_kc, st);
    }
    // few output tiles and a long reduction (weight gradients): split K over workgroups
    const long tiles = (long)as_cdiv(g->M, 64) * as_cdiv(g->N, 64) * g->batch;
    if (g->splitk_ws && !grouped && tiles < 512 && g->K >= 512 && !g->bias && g->act == 0) {
        long sk = (1024 + tiles - 1) / tiles;
        const int min_chunk = tiles * (g->K / 128) < 128 ? 64 : 128;  // a handful of tiles: shorter chunks, still >= 2 k-steps
        if (sk > g->K / min_chunk) sk = g->K / min_chunk;
        if (sk > 64) sk = 64;
        const long per = (long)g->batch * g->M * (g->N + (g->colsum ? 1 : 0));
        if (sk * per > g->splitk_ws_floats) sk = g->splitk_ws_floats / per;
        // weight-gradient shapes with many tiles per chunk: a multiple of 8 chunks, one XCD per chunk (GemmK::xcd_chunks)
        static const bool no_xcd = AS_DIAG_SET("AS_NO_XCD_CHUNKS");
        const bool want_xcd = !a_kc && !b_kc && sk >= 12 && tiles >= 16 && !no_xcd;
        if (want_xcd)   // the nearest multiple of 8 (downwards) whose BK-rounded chunks still number a multiple of 8
            for (long c = (sk + 4) / 8 * 8; c >= 8; c -= 8)
                if (as_cdiv(g->K, as_round_up(as_cdiv(g->K, c), BK)) % 8 == 0) { sk = c; break; }
        if (sk > 1) {
            k.kchunk = (int)as_round_up(as_cdiv(g->K, sk), BK);
            k.splitk = as_cdiv(g->K, k.kchunk);
            k.xcd_chunks = want_xcd && k.splitk % 8 == 0;
            k.slab = g->splitk_ws;
            // few slabs: the last workgroup to arrive at a tile sums them in the kernel (input gradient of GRU layer 1, 3 slabs:
            // 52 us against 36 + 45 for a reduce kernel that has to squeeze in beside the side stream's persistent GEMM).
            // Many slabs (a handful of tiles) would leave the sums to a handful of workgroups: the wide reduce kernel then.
            if (k.splitk <= 16 && tiles < COUNTERS) k.counters = counters_for(st);
        }
    }
    AS_TRY((launch<64, 64>(k, g->batch, a_kc, b_kc, st)));
    if (k.splitk > 1 && !k.counters) {
        const long total = (long)g->batch * g->M * (g->N + (g->colsum ? 1 : 0));
        if (k.splitk >= 16 && total <= (1L << 20)) {
            hipLaunchKernelGGL(splitk_reduce4_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, k);
        } else {
            long blocks = (total + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, st, k);
        }
        AS_LAUNCH_CHECK("as_gemm_f32(splitk reduce)");
    }
    return 0;
}
