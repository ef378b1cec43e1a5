// Row-wise and reduction kernels around the GEMMs of the contour-regression heads and the GRU:
// affine-free LayerNorm (the affine is folded into the following Linear, see fold_kernel), its
// backward fused with the ReLU mask, deterministic column sums (bias gradients), the fold/unfold of
// LayerNorm affine parameters, per-token segment sums (embedding / layer-0 input-projection
// gradients), gather, sigmoid backward and the flat Adam update.  All HBM-bound: one wave per row,
// coalesced 4-byte or 16-byte lanes, wave shuffles for the row reductions.
#include <algorithm>

#include "gemm_internal.h"
#include "rowops.h"

namespace {

constexpr int MAXC = 8;   // row length <= 64 * MAXC for the head kernels (register resident rows)
constexpr int MAXCW = 44;  // wide rows (transformer: LayerNorm over 10*d / A*d features)

// ---- x_hat = (x - mean) * rstd over the last dim (biased variance, eps) -------------------------
// one wave per row, NW = ceil(D / 64) values per lane; branch-free clamped loads (all in flight together)
template <int NW>
__global__ __launch_bounds__(256) void normalize_fwd_kernel(const float* __restrict__ x, float* __restrict__ xhat,
                                                            float* __restrict__ rstd, long rows, int D, float eps,
                                                            unsigned long long* __restrict__ pos_bits) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * D;
    float v[NW];
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int i = lane + 64 * c;
        v[c] = xr[i < D ? i : D - 1];
    }
    float s = 0.f;
    unsigned long long pos[NW];  // x > 0 per element: the compare result of a wave IS the 64-bit word
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        if (lane + 64 * c >= D) v[c] = 0.f;
        s += v[c];
        pos[c] = __ballot(v[c] > 0.f);
    }
    const float mean = as_wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const float d = lane + 64 * c < D ? v[c] - mean : 0.f;
        v[c] = d;
        q += d * d;
    }
    const float rs = 1.0f / sqrtf(as_wave_sum(q) / D + eps);
    float* o = xhat + row * D;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int i = lane + 64 * c;
        if (i < D) o[i] = v[c] * rs;
    }
    if (lane == 0) rstd[row] = rs;
    if (pos_bits && lane == 0) {  // ceil(D / 64) words per row: the ReLU mask the backward needs (instead of x itself)
        const int words = (D + 63) / 64;
#pragma unroll
        for (int c = 0; c < NW; ++c)
            if (c < words) pos_bits[row * words + c] = pos[c];
    }
}

// ---- dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat)) * (relu_src > 0) ------------------------
// dy and dx may alias (in-place): a lane reads all its elements before it writes any.
// MASKED = false: no ReLU mask operand (no registers for it: 88 instead of 132 live values per lane at MW = 44, one wave per
// SIMD became three)
// WHOLE: D is a multiple of 64, so a lane's element c is either in the row for the whole wave or for none of it: the index
// is lane + (a wave-uniform 64 c or 0) -- ONE address VGPR for all the loads instead of a clamped index per element
template <int MW, bool MASKED, bool WHOLE>
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float* dy, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ relu_src, float* dx,
                                                            long rows, int D, const unsigned long long* __restrict__ relu_bits,
                                                            int n_slab, long slab_stride) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* dyr = dy + row * D;
    const float* xr = xhat + row * D;
    float g[MW], h[MW], m[MASKED ? MW : 1];
#pragma unroll
    for (int c = 0; c < MW; ++c) {  // branch-free clamped loads: everything in flight together
        const int i = lane + 64 * c;
        const int ic = WHOLE ? lane + (64 * c < D ? 64 * c : 0) : (i < D ? i : D - 1);
        g[c] = dyr[ic];
        h[c] = xr[ic];
        if (MASKED) m[c] = 1.f;
    }
    // dy = the sum of n_slab partial results (a GEMM split over K, as_lin_plain_s6), added in slab order
    for (int sl = 1; sl < n_slab; ++sl) {
#pragma unroll
        for (int c = 0; c < MW; ++c) {
            const int i = lane + 64 * c;
            const int ic = WHOLE ? lane + (64 * c < D ? 64 * c : 0) : (i < D ? i : D - 1);
            g[c] += dyr[sl * slab_stride + ic];
        }
    }
    if (MASKED && relu_src) {  // (one uniform branch around the whole batch, not one per element)
        const float* rr = relu_src + row * D;
#pragma unroll
        for (int c = 0; c < MW; ++c) {
            const int i = lane + 64 * c;
            m[c] = rr[i < D ? i : D - 1];
        }
    }
    if (MASKED && relu_bits) {  // one 64-bit word per 64 elements (a wave-uniform load) instead of a third full-width operand
        const int words = (D + 63) / 64;
#pragma unroll
        for (int c = 0; c < MW; ++c) {
            const unsigned long long w = relu_bits[row * words + (c < words ? c : words - 1)];
            m[c] = (w >> lane) & 1ull ? 1.f : 0.f;
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MW; ++c) {
        if (lane + 64 * c >= D) g[c] = 0.f;
        s1 += g[c];
        s2 += g[c] * h[c];
    }
    const float m1 = as_wave_sum(s1) / D, m2 = as_wave_sum(s2) / D;
    const float rs = rstd[row];
    float* o = dx + row * D;
#pragma unroll
    for (int c = 0; c < MW; ++c) {
        const int i = lane + 64 * c;
        if (i < D) {
            float v = rs * (g[c] - m1 - h[c] * m2);
            if (MASKED && !(m[c] > 0.f)) v = 0.f;
            o[i] = v;
        }
    }
}


// ---- fold LayerNorm affine into the next Linear: Wf[n][k] = W[n][k] * gamma[k], bf[n] = b[n] + W[n].beta
// grid.x = heads * R rows (one wave per row), W [heads][R][K], gamma/beta [heads][K]
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ b,
                                                   float* __restrict__ Wf, float* __restrict__ bf, long total_rows, int R,
                                                   int K, int Rpad) {
    // Rpad >= R rows per head in Wf: rows R .. Rpad-1 are written as zeros (a reduction over them adds nothing)
    const long prow = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (prow >= total_rows) return;
    const long head = prow / Rpad;
    const int r = (int)(prow - head * Rpad);
    float* o = Wf + prow * K;
    if (r >= R) {
        for (int k = lane; k < K; k += 64) o[k] = 0.f;
        return;
    }
    const long row = head * R + r;
    const float* wr = W + row * K;
    const float* gm = gamma + head * K;
    const float* bt = beta + head * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = wr[k];
        o[k] = w * gm[k];
        s += w * bt[k];
    }
    s = as_wave_sum(s);
    if (lane == 0) bf[row] = b[row] + s;
}

// ---- a weight matrix as three bfloat16 planes for the split-arithmetic kernels (lin_f32.hip, lin_s6_kernel):
// x = hi + mid + lo exactly (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest even),
// out[plane][batch][Kpad / 16][rows_pad][16]: the 32 bytes a row contributes to a 16-deep k-step are one sector, and the
// rows of a k-step are consecutive (the B fragment of a wave = 1 KiB).  One thread per (batch, k-step, row).
struct PlanesJobs { as_planes_job j[8]; long first[9]; };

__global__ __launch_bounds__(256) void emit_planes_kernel(PlanesJobs jobs, int n_jobs) {
    // one thread per PAIR of consecutive k (one packed word of each plane): enough threads to hide the loads' latency -- a
    // thread per 16-deep row segment made the launch a chain of 12 k threads with 16 strided loads each (21 us for 0.4 M elements)
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= jobs.first[n_jobs]) return;
    int q = 0;
    while (q + 1 < n_jobs && gid >= jobs.first[q + 1]) ++q;
    const as_planes_job& J = jobs.j[q];
    long r = gid - jobs.first[q];
    const int e = (int)(r & 7);
    r >>= 3;
    const int n = (int)(r % J.rows_pad);
    r /= J.rows_pad;
    const int nks = J.Kpad / 16;
    const int ks = (int)(r % nks), b = (int)(r / nks);
    const float* src = J.B + (long)b * J.batch_stride + (long)n * J.n_stride;
    const int k = ks * 16 + 2 * e;
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) v[h] = (n < J.N && k + h < J.K) ? src[(long)(k + h) * J.k_stride] : 0.f;
    unsigned short p[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float x = v[h];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const __bf16 t = (__bf16)x;                       // round to nearest even
            p[pl][h] = __builtin_bit_cast(unsigned short, t);
            x -= (float)t;                                    // exact
        }
    }
    const long bstride = (long)nks * J.rows_pad * 16, pstride = (long)J.batch * bstride;
    unsigned* o = reinterpret_cast<unsigned*>(J.out + (long)b * bstride + ((long)ks * J.rows_pad + n) * 16) + e;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) o[pl * (pstride / 2)] = p[pl][0] | ((unsigned)p[pl][1] << 16);
}

// ---- unfold: dW[n][k] = dWf[n][k] * gamma[k] + dbf[n] * beta[k] ; dgamma[k] = sum_n dWf[n][k] W[n][k] ;
//      dbeta[k] = sum_n dbf[n] W[n][k]      (chain rule through W' = W.diag(gamma), b' = b + W.beta)
// grid (ceil(K/64), heads); block 1024 = 16 row-lanes x 64 columns (few, small matrices: favour
// parallelism per block over block count)
struct UnfoldJob {
    const float* dWf; const float* dbf; const float* W; const float* gamma; const float* beta;
    float* dW; float* dgamma; float* dbeta; int R, K;
};
struct UnfoldJobs { UnfoldJob j[3]; };

__device__ __forceinline__ void unfold_body(const float* __restrict__ dWf, const float* __restrict__ dbf,
                                            const float* __restrict__ W, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, float* __restrict__ dW,
                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int R, int K) {
    __shared__ float r1[16][64], r2[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    const long head = blockIdx.y;
    float sg = 0.f, sb = 0.f;
    if (k < K) {
        const float gm = gamma[head * K + k], bt = beta[head * K + k];
#pragma unroll 4
        for (int n = w; n < R; n += 16) {
            const long idx = (head * R + n) * K + k;
            const float dwf = dWf[idx], wv = W[idx], dbn = dbf[head * R + n];
            dW[idx] = dwf * gm + dbn * bt;
            sg += dwf * wv;
            sb += dbn * wv;
        }
    }
    r1[w][lane] = sg;
    r2[w][lane] = sb;
    __syncthreads();
    if (w == 0 && k < K) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a += r1[i][lane]; b += r2[i][lane]; }
        dgamma[head * K + k] = a;
        dbeta[head * K + k] = b;
    }
}
__global__ __launch_bounds__(1024) void unfold_kernel(const float* __restrict__ dWf, const float* __restrict__ dbf,
                                                      const float* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      float* __restrict__ dW, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, int R, int K) {
    unfold_body(dWf, dbf, W, gamma, beta, dW, dgamma, dbeta, R, K);
}
// the three LayerNorm affines of a head stack in ONE launch (blockIdx.z = layer): each is a handful of workgroups that
// live on memory latency, 7-10 us apiece when launched one after the other behind the weight-gradient GEMMs
__global__ __launch_bounds__(1024) void unfold3_kernel(UnfoldJobs jobs) {
    const UnfoldJob& j = jobs.j[blockIdx.z];
    if ((int)blockIdx.x * 64 >= j.K) return;   // workgroup-uniform: the grid is sized for the widest layer
    unfold_body(j.dWf, j.dbf, j.W, j.gamma, j.beta, j.dW, j.dgamma, j.dbeta, j.R, j.K);
}

// ---- out[v][:] = sum over rows m with token(m) == v of x[m][:]   (x [rows][C]) ---------------------
// grid (V, ceil(C/256)); the block scans the token array in chunks of 256, compacts the matches IN ORDER
// into LDS and then streams the matching rows with independent loads (bitwise reproducible).
__global__ __launch_bounds__(256) void token_segsum_kernel(const float* __restrict__ x, const int64_t* __restrict__ tokens,
                                                           long tok_stride, int T, long rows, int C,
                                                           float* __restrict__ out) {
    __shared__ int list[256];
    __shared__ int wcount[4];
    const int v = blockIdx.x;
    const int col = blockIdx.y * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (long base = 0; base < rows; base += 256) {
        const long m = base + threadIdx.x;
        bool match = false;
        if (m < rows) {
            const long b = m / T, t = m % T;
            match = tokens[b * tok_stride + t] == v;
        }
        // ordered (deterministic) compaction of the matching row indices: ballot + prefix popcount
        const unsigned long long bal = __ballot(match);
        if (lane == 0) wcount[w] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int i = 0; i < w; ++i) off += wcount[i];
        const int n = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        if (match) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)threadIdx.x;
        __syncthreads();
        if (col < C) {
            int i = 0;
            for (; i + 3 < n; i += 4) {  // four independent loads in flight; fixed summation order
                s0 += x[(base + list[i]) * C + col];
                s1 += x[(base + list[i + 1]) * C + col];
                s2 += x[(base + list[i + 2]) * C + col];
                s3 += x[(base + list[i + 3]) * C + col];
            }
            for (; i < n; ++i) s0 += x[(base + list[i]) * C + col];
        }
        __syncthreads();
    }
    if (col < C) out[(long)v * C + col] = (s0 + s1) + (s2 + s3);
}

// Two-stage form: grid (V, ceil(C/256), ceil(rows/256)) -- every block handles ONE 256-row chunk (one compaction, a handful
// of row loads) and writes its partial to part[chunk][v][:]; token_segsum_reduce_kernel adds the chunks in order.  Same
// result order for every launch (bitwise reproducible), ~25x the parallelism of the single-stage kernel at rows = 6400.
__global__ __launch_bounds__(256) void token_segsum_part_kernel(const float* __restrict__ x, const int64_t* __restrict__ tokens,
                                                                long tok_stride, int T, long rows, int C, int V,
                                                                float* __restrict__ part) {
    __shared__ int list[256];
    __shared__ int wcount[4];
    const int v = blockIdx.x;
    const int col = blockIdx.y * 256 + threadIdx.x;
    const long base = (long)blockIdx.z * 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long m = base + threadIdx.x;
    bool match = false;
    if (m < rows) match = tokens[(m / T) * tok_stride + m % T] == v;
    const unsigned long long bal = __ballot(match);
    if (lane == 0) wcount[w] = __popcll(bal);
    __syncthreads();
    int off = 0;
    for (int i = 0; i < w; ++i) off += wcount[i];
    const int n = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    if (match) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)threadIdx.x;
    __syncthreads();
    if (col >= C) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 3 < n; i += 4) {
        s0 += x[(base + list[i]) * C + col];
        s1 += x[(base + list[i + 1]) * C + col];
        s2 += x[(base + list[i + 2]) * C + col];
        s3 += x[(base + list[i + 3]) * C + col];
    }
    for (; i < n; ++i) s0 += x[(base + list[i]) * C + col];
    part[((long)blockIdx.z * V + v) * C + col] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void token_segsum_reduce_kernel(const float* __restrict__ part, long n, int chunks,
                                                                  float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // eight chunk partials in flight at a time (a chain of dependent loads would cost a memory round trip per chunk); the
    // order of the additions is fixed, so the result is reproducible
    float s = 0.f;
    int c = 0;
    for (; c + 7 < chunks; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long)(c + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < chunks; ++c) s += part[(long)c * n + i];
    out[i] = s;
}

// Column-sliced form for wide rows (C a multiple of 64, V <= 128): grid (C / 64, chunks), one workgroup = 64 columns x one
// chunk of rows.  Four row lanes (threadIdx / 64) walk the chunk with stride 4 and add into four private [V][64] tables in
// LDS (1 KB x V in all), which are combined in a fixed order at the end: no atomics, bitwise reproducible, one read of x.
// 192 workgroups at rows = 6400, C = 768 instead of 3375 that each compact a token chunk and touch a handful of rows.
__global__ __launch_bounds__(256) void token_segsum_cols_kernel(const float* __restrict__ x, const int64_t* __restrict__ tokens,
                                                                long tok_stride, int T, long rows, int C, int V, int rows_per_chunk,
                                                                float* __restrict__ part) {
    extern __shared__ float tab[];   // [4][V][64]
    const int c64 = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c64;
    for (int i = threadIdx.x; i < 4 * V * 64; i += 256) tab[i] = 0.f;
    __syncthreads();
    float* mine = tab + (long)rl * V * 64 + c64;
    const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    // (utterance, frame) of this lane's current row, advanced by 4 rows per visit: no 64-bit division per row
    long r = r0 + rl;
    long tb = r / T;
    int tt = (int)(r - tb * T);
    auto next_tok = [&]() {
        const int v = (int)min(max(tokens[tb * tok_stride + tt], (int64_t)0), (int64_t)V - 1);   // never outside the LDS tables
        tt += 4;
        while (tt >= T) { tt -= T; ++tb; }
        return v;
    };
    for (; r + 28 < r1; r += 32) {   // eight rows of this lane in flight: the kernel lives on memory latency
        float xv[8];
        int tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            xv[u] = x[(r + 4 * u) * C + col];
            tv[u] = next_tok();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) mine[tv[u] * 64] += xv[u];
    }
    for (; r < r1; r += 4) mine[next_tok() * 64] += x[r * C + col];
    __syncthreads();
    for (int i = threadIdx.x; i < V * 64; i += 256) {
        const float sum = ((tab[i] + tab[V * 64 + i]) + tab[2 * V * 64 + i]) + tab[3 * V * 64 + i];
        part[((long)blockIdx.y * V + (i >> 6)) * C + blockIdx.x * 64 + (i & 63)] = sum;
    }
}

// Gradients of the embedding / layer-0 input projection from the per-token sums dtab [V][C] (C = 2 * 3H gate rows):
//   dW[c][e] = sum_v dtab[v][c] * emb[v][e],  db[c] = sum_v dtab[v][c]      blocks [0, C / 4): one wave per row c
//   demb[v][e] = sum_c dtab[v][c] * W[c][e]                                  blocks [C / 4, C / 4 + V): one block per token
// Two 64 x 64-tile GEMM launches under reductions of 45 and 768 took 10 + 21 us on the step's critical tail; this is one
// launch of a few microseconds.  E <= 256.
__global__ __launch_bounds__(1024) void emb_grads_kernel(const float* __restrict__ dtab, const float* __restrict__ emb,
                                                         const float* __restrict__ W, int V, int C, int E,
                                                         float* __restrict__ dW, float* __restrict__ db, float* __restrict__ demb) {
    // Everything is staged through LDS with all loads of a phase in flight at once: these are tiny reductions whose time is
    // the number of dependent memory round trips, not bytes or FLOPs.
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < C / 16) {
        // 16 rows c of dW: stage emb [V][E] and dtab[:, c0 .. c0+15], then thread (c, e) sums over v from LDS
        float* s_emb = sm;                 // V * E
        float* s_d = sm + (long)V * E;     // V * 16
        const int c0 = blockIdx.x * 16;
        for (int i = tid; i < V * E; i += 1024) s_emb[i] = emb[i];
        for (int i = tid; i < V * 16; i += 1024) s_d[i] = dtab[(long)(i >> 4) * C + c0 + (i & 15)];
        __syncthreads();
        for (int o = tid; o < 16 * E; o += 1024) {
            const int cc = o / E, e = o - cc * E;
            float a = 0.f;
            for (int v = 0; v < V; ++v) a += s_d[v * 16 + cc] * s_emb[v * E + e];
            dW[(long)(c0 + cc) * E + e] = a;
        }
        if (tid < 16) {
            float bsum = 0.f;
            for (int v = 0; v < V; ++v) bsum += s_d[v * 16 + tid];
            db[c0 + tid] = bsum;
        }
        return;
    }
    // one block per token v: demb[v][e] = sum_c dtab[v][c] * W[c][e]; thread (part, e): c = part, part + P, ... with 16 loads
    // of W in flight, partials combined through LDS in a fixed order
    const int v = blockIdx.x - C / 16;
    if (v >= V) return;
    float* s_row = sm;          // C
    float* s_part = sm + C;     // P * E
    for (int i = tid; i < C; i += 1024) s_row[i] = dtab[(long)v * C + i];
    __syncthreads();
    const int P = 1024 / E > 0 ? 1024 / E : 1;   // parts (E <= 1024)
    const int e = tid % E, part = tid / E;
    if (part < P) {
        float a = 0.f;
        int c = part;
        for (; c + 15 * P < C; c += 16 * P) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = W[(long)(c + u * P) * E + e];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += s_row[c + u * P] * wv[u];
        }
        for (; c < C; c += P) a += s_row[c] * W[(long)c * E + e];
        s_part[part * E + e] = a;
    }
    __syncthreads();
    if (tid < E) {
        float a = 0.f;
        for (int p = 0; p < P; ++p) a += s_part[p * E + tid];
        demb[(long)v * E + tid] = a;
    }
}

// ---- out[m][:] = table[token(m)][:] --------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ tokens,
                                                          long tok_stride, int T, long rows, int C,
                                                          float* __restrict__ out, int V) {
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const long b = m / T, t = m % T;
    const long v = V > 0 ? min(max(tokens[b * tok_stride + t], (int64_t)0), (int64_t)V - 1) : tokens[b * tok_stride + t];
    for (int c = threadIdx.x & 63; c < C; c += 64) out[m * C + c] = table[v * C + c];
}

// ---- token table of the layer-0 input projections: tab[v][c] = b[c] + sum_e emb[v][e] * W[c][e]  (V x C outputs, E <= 256).
// It is the FIRST kernel of every forward and the recurrence waits for it; as a general-GEMM launch (12 workgroups walking
// four dependent k-tiles) it took 10 us for 4.4 MFLOP.  Here: one workgroup per (64 output columns, 4 table rows) -- 144
// workgroups at V = 45, C = 768 --, its four embedding rows and 64 weight rows staged in LDS once (coalesced), one output per thread.
__global__ __launch_bounds__(256) void token_table_kernel(const float* __restrict__ emb, const float* __restrict__ W,
                                                          const float* __restrict__ bias, int V, int C, int E,
                                                          float* __restrict__ tab) {
    extern __shared__ __attribute__((aligned(16))) float tt_s[];   // four emb rows [4][E], then this workgroup's 64 weight rows [64][E + 1]
    float* emb_s = tt_s;
    float* w_s = tt_s + 4 * E;
    const int c0 = blockIdx.x * 64;
    const int v0 = blockIdx.y * 4;                         // this workgroup's four table rows (one per wave)
    for (int i = threadIdx.x; i < 4 * E; i += 256) emb_s[i] = v0 + i / E < V ? emb[(long)v0 * E + i] : 0.f;
    for (int i = threadIdx.x; i < 64 * E; i += 256) {      // coalesced: consecutive threads read consecutive floats of a row
        const int r = i / E, e = i - r * E;
        w_s[r * (E + 1) + e] = c0 + r < C ? W[(long)(c0 + r) * E + e] : 0.f;
    }
    __syncthreads();
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;   // a thread per (column, one of 4 row groups)
    const int c = c0 + cl;
    if (c >= C) return;
    const float bc = bias[c];
    const float* wr = w_s + cl * (E + 1);                      // row stride E + 1: the 64 lanes hit 64 different banks
    const int v = v0 + rg;
    if (v >= V) return;
    const float* er = emb_s + rg * E;                           // wave-uniform: broadcast reads
    float acc = bc;
#pragma unroll 8
    for (int e = 0; e < E; ++e) acc = fmaf(er[e], wr[e], acc);
    tab[(long)v * C + c] = acc;
}

// ---- batched collate on the device: out[b][t][:] = t < len[b] ? src[first[b] + t][:] : pad  (pad_sequence of a batch whose
// unpadded utterances lie back to back in one resident buffer; dataset.py:27-65 of the reference does this on the host)
template <typename T>
__global__ __launch_bounds__(256) void gather_pad_kernel(const T* __restrict__ src, const long* __restrict__ first,
                                                         const int* __restrict__ len, int Tmax, long row_elems, T pad,
                                                         T* __restrict__ out, long total) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const long row = i / row_elems, e = i - row * row_elems;
        const long b = row / Tmax;
        const int t = (int)(row - b * Tmax);
        out[i] = t < len[b] ? src[(first[b] + t) * row_elems + e] : pad;
    }
}

// ---- *count = ids outside [0, V) (nn.Embedding raises for them; the kernels clamp, the host reads this word) ----------
__global__ __launch_bounds__(1024) void count_bad_tokens_kernel(const int64_t* __restrict__ tokens, long tok_stride, int T, long rows,
                                                                int V, int* __restrict__ count) {
    __shared__ int part[16];
    int bad = 0;
    for (long m = threadIdx.x; m < rows; m += 1024) {
        const long b = m / T;
        const int64_t v = tokens[b * tok_stride + (m - b * T)];
        bad += (v < 0 || v >= V) ? 1 : 0;
    }
    bad = (int)as_wave_sum((float)bad);   // exact below 2^24 ids
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < 16; ++i) s += part[i];
        *count = s;
    }
}

// ---- dpre = dout * out * (1 - out) ----------------------------------------------------------------
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                          float* __restrict__ dpre, long n) {
    const long stride = (long)gridDim.x * 256;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 o = reinterpret_cast<const float4*>(out)[i];
        const float4 d = reinterpret_cast<const float4*>(dout)[i];
        float4 r;
        r.x = d.x * o.x * (1.f - o.x); r.y = d.y * o.y * (1.f - o.y);
        r.z = d.z * o.z * (1.f - o.z); r.w = d.w * o.w * (1.f - o.w);
        reinterpret_cast<float4*>(dpre)[i] = r;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        dpre[i] = dout[i] * out[i] * (1.f - out[i]);
}

// ---- y = relu-masked copy: dst = (src_mask > 0) ? g : 0 ---------------------------------------------
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ g, const float* __restrict__ act,
                                                        float* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = act[i] > 0.f ? g[i] : 0.f;
}

// ---- torch.optim.Adam (weight decay added to the gradient), flat ----------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float gscale) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float pi = p[i];
        float gi = g[i] * gscale;
        if (wd != 0.f) gi += wd * pi;
        const float mi = m[i] + (1.f - b1) * (gi - m[i]);      // lerp form, as torch's _single_tensor_adam
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// ---- inverted dropout with a counter-based generator: y[i] = x[i] * keep(seed, i) / (1 - p) -----------
// keep() is a pure function of (seed, element index) (splitmix64 finaliser), so the backward regenerates
// the very same mask from the seed instead of storing it.
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0, 1)
    return u >= p ? inv_keep : 0.f;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* x, float* y, long n, float p, float inv_keep,
                                                      unsigned long long seed) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = x[i] * dropout_scale(seed, i, p, inv_keep);
}

// ---- LayerNorm with optional residual input and grouped affine ------------------------------------
// one wave per row, NW = ceil(D / 64) values per lane (template: rows are 20 .. 2816 wide in the models); branch-free
// clamped loads so that all of a lane's loads are in flight together.
template <int NW, bool AFFINE, bool WHOLE = false>   // WHOLE: D % 64 == 0, one address VGPR for all loads (see normalize_bwd_kernel)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float* __restrict__ xhat,
                                                            float* __restrict__ rstd, long rows, int D, long group_rows,
                                                            float eps, int res_block, long res_rows) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * D;
    // res_block > 0 (as_layernorm_fwd_blockres): x rows are [channel][res_rows][per * res_block] -- the concatenation of `per`
    // blocks of res_block features -- and the residual is block-major, res[(channel * per + j)][row in channel][res_block]
    const float* rr = res ? res + (res_block > 0 ? ((row / res_rows) * (D / res_block) * res_rows + row % res_rows) * res_block : row * D) : nullptr;
    const long res_jstride = res_block > 0 ? res_rows * res_block - res_block : 0;   // block j's row starts j * res_rows * res_block further
    // group_rows > 0: consecutive blocks of rows share a parameter set; < 0: parameter set = row % (-group_rows)
    const long g = group_rows > 0 ? row / group_rows : (group_rows < 0 ? row % (-group_rows) : 0);
    float v[NW], ga[AFFINE ? NW : 1], be[AFFINE ? NW : 1];   // (affine-free: no registers for the parameters)
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int i = lane + 64 * c;
        const int ic = WHOLE ? lane + (64 * c < D ? 64 * c : 0) : (i < D ? i : D - 1);
        long ro = ic;
        if (res_block > 0) {
            // block index of feature ic = lane + 64 c: wave-uniform c / (res_block / 64) when the blocks are whole multiples of the wave
            // (lanes beyond the row re-read its last feature: clamp the block index with them)
            const int j = res_block % 64 == 0 ? (64 * c < D ? 64 * c : 0) / res_block : ic / res_block;
            ro += j * res_jstride;
        }
        v[c] = rr ? xr[ic] + rr[ro] : xr[ic];
        if (AFFINE) {
            ga[c] = gamma[g * D + ic];
            be[c] = beta[g * D + ic];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        if (lane + 64 * c >= D) v[c] = 0.f;
        s += v[c];
    }
    const float mean = as_wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const float d = lane + 64 * c < D ? v[c] - mean : 0.f;
        v[c] = d;
        q += d * d;
    }
    const float rs = 1.0f / sqrtf(as_wave_sum(q) / D + eps);
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int i = lane + 64 * c;
        if (i < D) {
            const float xh = v[c] * rs;
            if (xhat) xhat[row * D + i] = xh;
            if (y) y[row * D + i] = AFFINE ? xh * ga[c] + be[c] : xh;
        }
    }
    if (rstd && lane == 0) rstd[row] = rs;
}


// ---- wide rows (the transformer's LayerNorms over 10 d / A d features, D % 256 == 0): 16-byte accesses, a lane owns
// float4 lane + 64 c of its row -- a quarter of the load / store instructions of the scalar kernels above, and one address
// register for all of them (wave-uniform offsets).  NV = max float4 per lane (D <= 1024 NV).
constexpr int WIDE_NV = 11;   // 2816 = A d at A = 11, d = 256

// x_hat = affine-free LayerNorm(x [+ res]); res_block > 0: block-major residual (as_layernorm_fwd_blockres)
__global__ __launch_bounds__(256) void layernorm_fwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                 float* __restrict__ xhat, float* __restrict__ rstd, long rows, int D,
                                                                 float eps, int res_block, long res_rows) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = D >> 8;   // float4 per lane
    const float4* xr = reinterpret_cast<const float4*>(x + row * D) + lane;
    const float* rr = res ? res + (res_block > 0 ? ((row / res_rows) * (D / res_block) * res_rows + row % res_rows) * res_block : row * D) : nullptr;
    const long res_jstride = res_block > 0 ? res_rows * res_block - res_block : 0;
    float4 v[WIDE_NV];
#pragma unroll
    for (int c = 0; c < WIDE_NV; ++c) {
        const int cc = c < nv ? c : 0;                       // (wave-uniform: lanes beyond the row re-read its first chunk)
        v[c] = xr[64 * cc];
        if (rr) {
            const int e = 4 * lane + 256 * cc;               // first feature of this float4 (never straddles a block: block % 4 == 0)
            const float4 t = *reinterpret_cast<const float4*>(rr + e + (res_block > 0 ? (e / res_block) * res_jstride : 0));
            v[c].x += t.x; v[c].y += t.y; v[c].z += t.z; v[c].w += t.w;
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < WIDE_NV; ++c) {
        if (c >= nv) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    }
    const float mean = as_wave_sum(sum) / D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < WIDE_NV; ++c) {
        if (c < nv) {
            v[c].x -= mean; v[c].y -= mean; v[c].z -= mean; v[c].w -= mean;
            q += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
        }
    }
    const float rs = 1.0f / sqrtf(as_wave_sum(q) / D + eps);
    float4* o = reinterpret_cast<float4*>(xhat + row * D) + lane;
#pragma unroll
    for (int c = 0; c < WIDE_NV; ++c)
        if (c < nv) o[64 * c] = make_float4(v[c].x * rs, v[c].y * rs, v[c].z * rs, v[c].w * rs);
    if (rstd && lane == 0) rstd[row] = rs;
}

// dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat)), no ReLU mask
__global__ __launch_bounds__(256) void normalize_bwd_wide_kernel(const float* dy, const float* __restrict__ xhat,
                                                                 const float* __restrict__ rstd, float* dx, long rows, int D) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = D >> 8;
    const float4* gr = reinterpret_cast<const float4*>(dy + row * D) + lane;
    const float4* hr = reinterpret_cast<const float4*>(xhat + row * D) + lane;
    float4 g[WIDE_NV], h[WIDE_NV];
#pragma unroll
    for (int c = 0; c < WIDE_NV; ++c) {
        const int cc = c < nv ? c : 0;
        g[c] = gr[64 * cc];
        h[c] = hr[64 * cc];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < WIDE_NV; ++c) {
        if (c >= nv) g[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        s1 += (g[c].x + g[c].y) + (g[c].z + g[c].w);
        s2 += (g[c].x * h[c].x + g[c].y * h[c].y) + (g[c].z * h[c].z + g[c].w * h[c].w);
    }
    const float m1 = as_wave_sum(s1) / D, m2 = as_wave_sum(s2) / D;
    const float rs = rstd[row];
    float4* o = reinterpret_cast<float4*>(dx + row * D) + lane;   // (dy and dx may alias: everything is read before the first store)
#pragma unroll
    for (int c = 0; c < WIDE_NV; ++c)
        if (c < nv)
            o[64 * c] = make_float4(rs * (g[c].x - m1 - h[c].x * m2), rs * (g[c].y - m1 - h[c].y * m2), rs * (g[c].z - m1 - h[c].z * m2),
                                    rs * (g[c].w - m1 - h[c].w * m2));
}

inline bool wide_ok(int D, const void* a, const void* b, const void* c) {
    return D % 256 == 0 && D > 1024 && D <= 256 * WIDE_NV &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// ---- masked softmax over the last dim of [Z][Tq][Tk], one wave per row ------------------------------
// NW = ceil(Tk / 64) values per lane (template: no wasted iterations); loads are branch-free (clamped index + select) so
// they all go in flight together -- a per-element `if (k < Tk)` load makes hipcc wait for each one.
template <int NW>
__global__ __launch_bounds__(256) void attn_softmax_kernel(float* __restrict__ s, long rows, int Tq, int Tk, int heads, int B,
                                                           float scale, const float* __restrict__ attn_mask,
                                                           const float* __restrict__ kpm) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const long z = row / Tq;
    const int q = (int)(row - z * Tq);
    const int b = (int)((z / heads) % B);
    float* sr = s + row * Tk;
    const float* am = attn_mask ? attn_mask + ((long)b * Tq + q) * Tk : nullptr;
    const float* km = kpm ? kpm + (long)b * Tk : nullptr;
    float v[NW], ma[NW], mk[NW];
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int k = lane + 64 * c;
        const int kc = k < Tk ? k : Tk - 1;
        v[c] = sr[kc];
        ma[c] = am ? am[kc] : 0.f;
        mk[c] = km ? km[kc] : 0.f;
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const float x = lane + 64 * c < Tk ? v[c] * scale + ma[c] + mk[c] : -INFINITY;
        v[c] = x;
        m = fmaxf(m, x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const float e = lane + 64 * c < Tk ? __expf(v[c] - m) : 0.f;  // all -inf row: -inf - -inf = NaN, like PyTorch
        v[c] = e;
        sum += e;
    }
    sum = as_wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int k = lane + 64 * c;
        if (k < Tk) sr[k] = v[c] * inv;
    }
}

// ---- softmax backward over the last dim, in place on dP: dS = P * (dP - sum_k dP*P) * scale --------
template <int NW>
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long rows,
                                                               int Tk, float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + row * Tk;
    float* dr = dp + row * Tk;
    float pv[NW], dv[NW];
#pragma unroll
    for (int c = 0; c < NW; ++c) {  // branch-free: clamped index, padded lanes contribute zero below
        const int k = lane + 64 * c;
        const int kc = k < Tk ? k : Tk - 1;
        pv[c] = pr[kc];
        dv[c] = dr[kc];
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        if (lane + 64 * c >= Tk) pv[c] = 0.f;
        s += pv[c] * dv[c];
    }
    s = as_wave_sum(s);
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const int k = lane + 64 * c;
        if (k < Tk) dr[k] = pv[c] * (dv[c] - s) * scale;
    }
}

// ---- dst[c][i] = sum over groups g with src[g] == c of part[g][i]   (i over `len` contiguous floats) ---
// deterministic: groups are visited in index order.  grid.y = channel, grid.x over the row.
__global__ __launch_bounds__(256) void group_reduce_kernel(const float* __restrict__ part, const int* __restrict__ src, int G,
                                                           long len, float* __restrict__ dst) {
    const int c = blockIdx.y;
    const long n4 = len / 4;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < G; ++g)
            if (src[g] == c) {
                const float4 v = reinterpret_cast<const float4*>(part + (long)g * len)[i];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        reinterpret_cast<float4*>(dst + (long)c * len)[i] = acc;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < len; i += stride) {
        float acc = 0.f;
        for (int g = 0; g < G; ++g)
            if (src[g] == c) acc += part[(long)g * len + i];
        dst[(long)c * len + i] = acc;
    }
}

__global__ __launch_bounds__(256) void embed_posenc_kernel(const int64_t* __restrict__ tokens, long tok_stride,
                                                           const float* __restrict__ table, const float* __restrict__ pe,
                                                           float* __restrict__ out, long rows, int T, int D) {
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const long b = m / T, t = m % T;
    const float* src = table ? table + tokens[b * tok_stride + t] * D : out + m * D;
    for (int c = threadIdx.x & 63; c < D; c += 64) out[m * D + c] = src[c] + pe[t * D + c];
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = b ? a[i] + b[i] : a[i];
}
__global__ __launch_bounds__(256) void row_scale_kernel(const float* __restrict__ a, const float* __restrict__ rs,
                                                        float* __restrict__ dst, long n, int row_len) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = a[i] * rs[i / row_len];
}

inline int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

int as_normalize_fwd(const float* x, float* xhat, float* rstd, long rows, int D, hipStream_t st, unsigned long long* pos_bits) {
    AS_REQUIRE(D > 0 && D <= 64 * MAXC, AS_ERR_UNSUPPORTED, "normalize: row length %d > %d", D, 64 * MAXC);
#define AS_NORM_FWD(NW) hipLaunchKernelGGL(normalize_fwd_kernel<NW>, dim3(as_cdiv(rows, 4)), dim3(256), 0, st, x, xhat, rstd, rows, D, 1e-5f, pos_bits)
    if (D <= 64) AS_NORM_FWD(1);
    else if (D <= 128) AS_NORM_FWD(2);
    else if (D <= 256) AS_NORM_FWD(4);
    else AS_NORM_FWD(MAXC);
#undef AS_NORM_FWD
    AS_LAUNCH_CHECK("normalize_fwd");
    return 0;
}
int as_normalize_bwd(const float* dy, const float* xhat, const float* rstd, const float* relu_src, float* dx, long rows,
                     int D, hipStream_t st, const unsigned long long* relu_bits, int n_slab, long slab_stride) {
    AS_REQUIRE(D > 0 && D <= 64 * MAXCW, AS_ERR_UNSUPPORTED, "normalize: row length %d > %d", D, 64 * MAXCW);
    if (n_slab < 1) n_slab = 1;
    if (!relu_src && !relu_bits && n_slab == 1 && wide_ok(D, dy, xhat, dx)) {
        hipLaunchKernelGGL(normalize_bwd_wide_kernel, dim3(as_cdiv(rows, 4)), dim3(256), 0, st, dy, xhat, rstd, dx, rows, D);
        AS_LAUNCH_CHECK("normalize_bwd");
        return 0;
    }
#define AS_NORM_BWD(MW)                                                                                                                \
    do {                                                                                                                               \
        if (relu_src || relu_bits)                                                                                                     \
            hipLaunchKernelGGL((normalize_bwd_kernel<MW, true, false>), dim3(as_cdiv(rows, 4)), dim3(256), 0, st, dy, xhat, rstd, relu_src, \
                               dx, rows, D, relu_bits, n_slab, slab_stride);                                                           \
        else if (D % 64 == 0)                                                                                                          \
            hipLaunchKernelGGL((normalize_bwd_kernel<MW, false, true>), dim3(as_cdiv(rows, 4)), dim3(256), 0, st, dy, xhat, rstd, relu_src, \
                               dx, rows, D, relu_bits, n_slab, slab_stride);                                                           \
        else                                                                                                                           \
            hipLaunchKernelGGL((normalize_bwd_kernel<MW, false, false>), dim3(as_cdiv(rows, 4)), dim3(256), 0, st, dy, xhat, rstd, relu_src, \
                               dx, rows, D, relu_bits, n_slab, slab_stride);                                                           \
    } while (0)
    if (D <= 128) AS_NORM_BWD(2);
    else if (D <= 256) AS_NORM_BWD(4);
    else if (D <= 64 * MAXC) AS_NORM_BWD(MAXC);
    else AS_NORM_BWD(MAXCW);
#undef AS_NORM_BWD
    AS_LAUNCH_CHECK("normalize_bwd");
    return 0;
}
int as_fold(const float* W, const float* gamma, const float* beta, const float* b, float* Wf, float* bf, int heads, int R,
            int K, hipStream_t st, int Rpad) {
    if (Rpad < R) Rpad = R;
    const long total = (long)heads * Rpad;
    hipLaunchKernelGGL(fold_kernel, dim3(as_cdiv(total, 4)), dim3(256), 0, st, W, gamma, beta, b, Wf, bf, total, R, K, Rpad);
    AS_LAUNCH_CHECK("fold");
    return 0;
}
int as_emit_planes(const as_planes_job* jobs, int n, hipStream_t st) {
    AS_REQUIRE(jobs && n >= 1 && n <= 8, AS_ERR_BAD_ARG, "as_emit_planes: 1..8 jobs");
    PlanesJobs pj{};
    long total = 0;
    for (int i = 0; i < n; ++i) {
        const as_planes_job& J = jobs[i];
        AS_REQUIRE(J.B && J.out && J.batch > 0 && J.N > 0 && J.K > 0 && J.rows_pad >= J.N && J.Kpad >= J.K && J.Kpad % 16 == 0 &&
                   (reinterpret_cast<uintptr_t>(J.out) & 15) == 0, AS_ERR_BAD_ARG, "as_emit_planes: bad job %d", i);
        pj.j[i] = J;
        pj.first[i] = total;
        total += (long)J.batch * (J.Kpad / 16) * J.rows_pad * 8;
    }
    pj.first[n] = total;
    hipLaunchKernelGGL(emit_planes_kernel, dim3(as_cdiv(total, 256)), dim3(256), 0, st, pj, n);
    AS_LAUNCH_CHECK("emit_planes");
    return 0;
}
int as_unfold(const float* dWf, const float* dbf, const float* W, const float* gamma, const float* beta, float* dW,
              float* dgamma, float* dbeta, int heads, int R, int K, hipStream_t st) {
    hipLaunchKernelGGL(unfold_kernel, dim3(as_cdiv(K, 64), heads), dim3(1024), 0, st, dWf, dbf, W, gamma, beta, dW, dgamma, dbeta, R, K);
    AS_LAUNCH_CHECK("unfold");
    return 0;
}
int as_unfold3(const float* const dWf[3], const float* const dbf[3], const float* const W[3], const float* const gamma[3],
               const float* const beta[3], float* const dW[3], float* const dgamma[3], float* const dbeta[3], const int R[3],
               const int K[3], int heads, hipStream_t st, int count) {
    UnfoldJobs jobs{};
    int kmax = 0;
    for (int i = 0; i < count; ++i) {
        jobs.j[i] = UnfoldJob{dWf[i], dbf[i], W[i], gamma[i], beta[i], dW[i], dgamma[i], dbeta[i], R[i], K[i]};
        kmax = K[i] > kmax ? K[i] : kmax;
    }
    hipLaunchKernelGGL(unfold3_kernel, dim3(as_cdiv(kmax, 64), heads, count), dim3(1024), 0, st, jobs);
    AS_LAUNCH_CHECK("unfold3");
    return 0;
}
int as_token_segsum(const float* x, const int64_t* tokens, long tok_stride, int T, long rows, int C, int V, float* out,
                    hipStream_t st, float* scratch, long scratch_floats) {
    if (scratch && C % 64 == 0 && V <= 128 && rows >= 1024) {   // wide rows: column-sliced single pass + ordered reduce
        const int chunks = (int)as_cdiv(rows, 128);   // 32 rows per row lane: four batches of eight loads
        const int rpc = (int)as_round_up(as_cdiv(rows, chunks), 4);
        if ((long)chunks * V * C <= scratch_floats) {
            hipLaunchKernelGGL(token_segsum_cols_kernel, dim3(C / 64, as_cdiv(rows, rpc)), dim3(256), (size_t)4 * V * 64 * sizeof(float), st, x,
                               tokens, tok_stride, T, rows, C, V, rpc, scratch);
            hipLaunchKernelGGL(token_segsum_reduce_kernel, dim3(as_cdiv((long)V * C, 256)), dim3(256), 0, st, scratch, (long)V * C,
                               as_cdiv(rows, rpc), out);
            AS_LAUNCH_CHECK("token_segsum");
            return 0;
        }
    }
    const int chunks = as_cdiv(rows, 256);
    if (scratch && chunks > 1 && chunks <= 65535 && (long)chunks * V * C <= scratch_floats) {
        hipLaunchKernelGGL(token_segsum_part_kernel, dim3(V, as_cdiv(C, 256), chunks), dim3(256), 0, st, x, tokens, tok_stride, T, rows,
                           C, V, scratch);
        hipLaunchKernelGGL(token_segsum_reduce_kernel, dim3(as_cdiv((long)V * C, 256)), dim3(256), 0, st, scratch, (long)V * C, chunks,
                           out);
    } else {
        hipLaunchKernelGGL(token_segsum_kernel, dim3(V, as_cdiv(C, 256)), dim3(256), 0, st, x, tokens, tok_stride, T, rows, C, out);
    }
    AS_LAUNCH_CHECK("token_segsum");
    return 0;
}
int as_sum_partials(const float* part, long n, int chunks, float* out, hipStream_t st) {
    hipLaunchKernelGGL(token_segsum_reduce_kernel, dim3(as_cdiv(n, 256)), dim3(256), 0, st, part, n, chunks, out);
    AS_LAUNCH_CHECK("sum_partials");
    return 0;
}
int as_emb_grads(const float* dtab, const float* emb, const float* W, int V, int C, int E, float* dW, float* db, float* demb,
                 hipStream_t st) {
    AS_REQUIRE(C % 16 == 0 && E <= 256 && V <= 128, AS_ERR_UNSUPPORTED, "emb_grads: V=%d C=%d E=%d", V, C, E);
    const long f1 = (long)V * E + (long)V * 16, f2 = (long)C + (long)(1024 / E) * E;
    const size_t shm = (size_t)(f1 > f2 ? f1 : f2) * sizeof(float);
    hipLaunchKernelGGL(emb_grads_kernel, dim3(C / 16 + V), dim3(1024), shm, st, dtab, emb, W, V, C, E, dW, db, demb);
    AS_LAUNCH_CHECK("emb_grads");
    return 0;
}
int as_gather_rows(const float* table, const int64_t* tokens, long tok_stride, int T, long rows, int C, float* out,
                   hipStream_t st, int V) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(as_cdiv(rows, 4)), dim3(256), 0, st, table, tokens, tok_stride, T, rows, C, out, V);
    AS_LAUNCH_CHECK("gather_rows");
    return 0;
}
int as_token_table(const float* emb, const float* W, const float* bias, int V, int C, int E, float* tab, hipStream_t st) {
    hipLaunchKernelGGL(token_table_kernel, dim3(as_cdiv(C, 64), as_cdiv(V, 4)), dim3(256), ((size_t)4 * E + 64 * (E + 1)) * sizeof(float), st, emb,
                       W, bias, V, C, E, tab);
    AS_LAUNCH_CHECK("token_table");
    return 0;
}
int as_count_bad_tokens(const int64_t* tokens, long tok_stride, int T, long rows, int V, int* count, hipStream_t st) {
    hipLaunchKernelGGL(count_bad_tokens_kernel, dim3(1), dim3(1024), 0, st, tokens, tok_stride, T, rows, V, count);
    AS_LAUNCH_CHECK("count_bad_tokens");
    return 0;
}
int as_sigmoid_bwd(const float* out, const float* dout, float* dpre, long n, hipStream_t st) {
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, st, out, dout, dpre, n);
    AS_LAUNCH_CHECK("sigmoid_bwd");
    return 0;
}
int as_dropout(const float* x, float* y, long n, float p, unsigned long long seed, hipStream_t st) {
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, y, n, p, 1.0f / (1.0f - p), seed);
    AS_LAUNCH_CHECK("dropout");
    return 0;
}
int as_relu_mask(const float* g, const float* act, float* dst, long n, hipStream_t st) {
    hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, st, g, act, dst, n);
    AS_LAUNCH_CHECK("relu_mask");
    return 0;
}

extern "C" int as_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* xhat,
                                float* rstd, int64_t rows, int32_t D, int64_t group_rows, void* stream) {
    AS_REQUIRE(x && (y || xhat) && rows > 0 && D > 0 && (!gamma == !beta), AS_ERR_BAD_ARG, "as_layernorm_fwd: bad argument");
    AS_REQUIRE(D <= 64 * 44, AS_ERR_UNSUPPORTED, "as_layernorm_fwd: row length %d > %d", D, 64 * 44);
    // (the 16-byte forward kernel is built and tested, but not dispatched: it sums a row in a different order, and the full-width
    // transformer's contours -- 0.89 of the 1e-4 band against the reference with the scalar kernel -- land at 1.03 with it:
    // rounding noise either way, but the band is the contract; the gain was 0.6 ms of a 70 ms forward)
    static const bool wide_fwd = AS_DIAG_SET("AS_LN_WIDE_FWD");
    if (wide_fwd && !gamma && !y && wide_ok(D, x, res, xhat)) {   // wide affine-free rows: 16-byte accesses
        hipLaunchKernelGGL(layernorm_fwd_wide_kernel, dim3(as_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, res, xhat, rstd, (long)rows,
                           D, 1e-5f, 0, 0L);
        AS_LAUNCH_CHECK("as_layernorm_fwd");
        return 0;
    }
#define AS_LN_FWD(NW)                                                                                                                  \
    do {                                                                                                                               \
        if (gamma)                                                                                                                     \
            hipLaunchKernelGGL((layernorm_fwd_kernel<NW, true>), dim3(as_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, res, gamma,  \
                               beta, y, xhat, rstd, (long)rows, D, (long)group_rows, 1e-5f, 0, 0L);                                         \
        else if (D % 64 == 0)                                                                                                          \
            hipLaunchKernelGGL((layernorm_fwd_kernel<NW, false, true>), dim3(as_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, res,  \
                               gamma, beta, y, xhat, rstd, (long)rows, D, (long)group_rows, 1e-5f, 0, 0L);                                  \
        else                                                                                                                           \
            hipLaunchKernelGGL((layernorm_fwd_kernel<NW, false>), dim3(as_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, res, gamma, \
                               beta, y, xhat, rstd, (long)rows, D, (long)group_rows, 1e-5f, 0, 0L);                                         \
    } while (0)
    if (D <= 64) AS_LN_FWD(1);
    else if (D <= 128) AS_LN_FWD(2);
    else if (D <= 256) AS_LN_FWD(4);
    else if (D <= 512) AS_LN_FWD(8);
    else if (D <= 1024) AS_LN_FWD(16);
    else AS_LN_FWD(44);
#undef AS_LN_FWD
    AS_LAUNCH_CHECK("as_layernorm_fwd");
    return 0;
}

extern "C" int as_layernorm_fwd_blockres(const float* x, const float* res, float* xhat, float* rstd, int32_t channels, int64_t rows,
                                         int32_t per, int32_t block, void* stream) {
    AS_REQUIRE(x && res && xhat && channels > 0 && rows > 0 && per > 0 && block > 0, AS_ERR_BAD_ARG, "as_layernorm_fwd_blockres: bad argument");
    const int D = per * block;
    AS_REQUIRE(D <= 64 * 44, AS_ERR_UNSUPPORTED, "as_layernorm_fwd_blockres: row length %d > %d", D, 64 * 44);
    const long total = (long)channels * rows;
    static const bool wide_fwd = AS_DIAG_SET("AS_LN_WIDE_FWD");   // (see as_layernorm_fwd)
    if (wide_fwd && block % 4 == 0 && wide_ok(D, x, res, xhat)) {
        hipLaunchKernelGGL(layernorm_fwd_wide_kernel, dim3(as_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, x, res, xhat, rstd, total, D,
                           1e-5f, block, (long)rows);
        AS_LAUNCH_CHECK("as_layernorm_fwd_blockres");
        return 0;
    }
#define AS_LN_FWD(NW)                                                                                                                \
    do {                                                                                                                             \
        if (block % 64 == 0)                                                                                                         \
            hipLaunchKernelGGL((layernorm_fwd_kernel<NW, false, true>), dim3(as_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, x, res, \
                               (const float*)nullptr, (const float*)nullptr, (float*)nullptr, xhat, rstd, total, D, 0L, 1e-5f, block,      \
                               (long)rows);                                                                                          \
        else                                                                                                                         \
            hipLaunchKernelGGL((layernorm_fwd_kernel<NW, false>), dim3(as_cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, x, res,      \
                               (const float*)nullptr, (const float*)nullptr, (float*)nullptr, xhat, rstd, total, D, 0L, 1e-5f, block,      \
                               (long)rows);                                                                                          \
    } while (0)
    if (D <= 64) AS_LN_FWD(1);
    else if (D <= 128) AS_LN_FWD(2);
    else if (D <= 256) AS_LN_FWD(4);
    else if (D <= 512) AS_LN_FWD(8);
    else if (D <= 1024) AS_LN_FWD(16);
    else AS_LN_FWD(44);
#undef AS_LN_FWD
    AS_LAUNCH_CHECK("as_layernorm_fwd_blockres");
    return 0;
}

extern "C" int as_fold_ln(const float* W, const float* gamma, const float* beta, const float* b, float* Wf, float* bf,
                          int32_t heads, int32_t R, int32_t K, void* stream) {
    AS_REQUIRE(W && gamma && beta && b && Wf && bf && heads > 0 && R > 0 && K > 0, AS_ERR_BAD_ARG, "as_fold_ln: bad argument");
    return as_fold(W, gamma, beta, b, Wf, bf, heads, R, K, (hipStream_t)stream);
}

extern "C" int as_attn_softmax(float* scores, int64_t Z, int32_t Tq, int32_t Tk, int32_t heads, int32_t B, float scale,
                               const float* attn_mask, const float* key_padding_mask, void* stream) {
    AS_REQUIRE(scores && Z > 0 && Tq > 0 && Tk > 0 && heads > 0 && B > 0, AS_ERR_BAD_ARG, "as_attn_softmax: bad argument");
    AS_REQUIRE(Tk <= 1024, AS_ERR_UNSUPPORTED, "as_attn_softmax: Tk=%d > 1024", Tk);
    const long rows = (long)Z * Tq;
#define AS_SOFTMAX(NW)                                                                                                      \
    hipLaunchKernelGGL(attn_softmax_kernel<NW>, dim3(as_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, scores, rows, Tq, Tk, \
                       heads, B, scale, attn_mask, key_padding_mask)
    if (Tk <= 64) AS_SOFTMAX(1);
    else if (Tk <= 128) AS_SOFTMAX(2);
    else if (Tk <= 256) AS_SOFTMAX(4);
    else if (Tk <= 512) AS_SOFTMAX(8);
    else AS_SOFTMAX(16);
#undef AS_SOFTMAX
    AS_LAUNCH_CHECK("as_attn_softmax");
    return 0;
}

extern "C" int as_attn_softmax_bwd(const float* probs, float* dprobs, int64_t Z, int32_t Tq, int32_t Tk, float scale,
                                   void* stream) {
    AS_REQUIRE(probs && dprobs && Z > 0 && Tq > 0 && Tk > 0, AS_ERR_BAD_ARG, "as_attn_softmax_bwd: bad argument");
    AS_REQUIRE(Tk <= 1024, AS_ERR_UNSUPPORTED, "as_attn_softmax_bwd: Tk=%d > 1024", Tk);
    const long rows = (long)Z * Tq;
#define AS_SOFTMAX_BWD(NW) \
    hipLaunchKernelGGL(attn_softmax_bwd_kernel<NW>, dim3(as_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, probs, dprobs, rows, Tk, scale)
    if (Tk <= 64) AS_SOFTMAX_BWD(1);
    else if (Tk <= 128) AS_SOFTMAX_BWD(2);
    else if (Tk <= 256) AS_SOFTMAX_BWD(4);
    else if (Tk <= 512) AS_SOFTMAX_BWD(8);
    else AS_SOFTMAX_BWD(16);
#undef AS_SOFTMAX_BWD
    AS_LAUNCH_CHECK("as_attn_softmax_bwd");
    return 0;
}

extern "C" int as_group_reduce(const float* part, const int32_t* src, int32_t G, int32_t C, int64_t len, float* dst, void* stream) {
    AS_REQUIRE(part && src && dst && G > 0 && C > 0 && len > 0, AS_ERR_BAD_ARG, "as_group_reduce: bad argument");
    AS_REQUIRE((reinterpret_cast<uintptr_t>(part) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && len % 4 == 0,
               AS_ERR_BAD_ARG, "as_group_reduce: buffers must be 16-byte aligned and len a multiple of 4");
    long bx = (len / 4 + 255) / 256;
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(group_reduce_kernel, dim3((int)bx, C), dim3(256), 0, (hipStream_t)stream, part, src, G, (long)len, dst);
    AS_LAUNCH_CHECK("as_group_reduce");
    return 0;
}

extern "C" int as_layernorm_bwd(const float* dxhat, const float* xhat, const float* rstd, const float* relu_src, float* dx,
                                int64_t rows, int32_t D, void* stream) {
    AS_REQUIRE(dxhat && xhat && rstd && dx && rows > 0 && D > 0, AS_ERR_BAD_ARG, "as_layernorm_bwd: bad argument");
    return as_normalize_bwd(dxhat, xhat, rstd, relu_src, dx, (long)rows, D, (hipStream_t)stream);
}

extern "C" int as_unfold_ln(const float* dWf, const float* dbf, const float* W, const float* gamma, const float* beta, float* dW,
                            float* dgamma, float* dbeta, int32_t heads, int32_t R, int32_t K, void* stream) {
    AS_REQUIRE(dWf && dbf && W && gamma && beta && dW && dgamma && dbeta && heads > 0 && R > 0 && K > 0, AS_ERR_BAD_ARG,
               "as_unfold_ln: bad argument");
    return as_unfold(dWf, dbf, W, gamma, beta, dW, dgamma, dbeta, heads, R, K, (hipStream_t)stream);
}

extern "C" int as_relu_bwd(const float* g, const float* act, float* dst, int64_t n, void* stream) {
    AS_REQUIRE(g && act && dst && n > 0, AS_ERR_BAD_ARG, "as_relu_bwd: bad argument");
    return as_relu_mask(g, act, dst, (long)n, (hipStream_t)stream);
}

extern "C" int as_gather_pad_rows(const void* src, const int64_t* first_row, const int32_t* lengths, int32_t B, int32_t T,
                                  int64_t row_elems, int32_t elem_bytes, double pad_value, void* out, void* stream) {
    AS_REQUIRE(src && first_row && lengths && out && B > 0 && T > 0 && row_elems > 0, AS_ERR_BAD_ARG, "as_gather_pad_rows: bad argument");
    AS_REQUIRE(elem_bytes == 4 || elem_bytes == 8, AS_ERR_UNSUPPORTED, "as_gather_pad_rows: element size %d", elem_bytes);
    const long total = (long)B * T * row_elems;
    const unsigned grid = (unsigned)(total / 256 + 1 < 16384 ? total / 256 + 1 : 16384);
    if (elem_bytes == 4)
        hipLaunchKernelGGL(gather_pad_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)src, (const long*)first_row,
                           lengths, T, (long)row_elems, (float)pad_value, (float*)out, total);
    else   // int64 token ids
        hipLaunchKernelGGL(gather_pad_kernel<long>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long*)src, (const long*)first_row,
                           lengths, T, (long)row_elems, (long)pad_value, (long*)out, total);
    AS_LAUNCH_CHECK("as_gather_pad_rows");
    return 0;
}

extern "C" int as_embed_posenc(const int64_t* tokens, int64_t tok_stride, const float* table, const float* pe, float* out,
                               int64_t rows, int32_t T, int32_t D, void* stream) {
    AS_REQUIRE(pe && out && rows > 0 && T > 0 && D > 0 && (!table || tokens), AS_ERR_BAD_ARG, "as_embed_posenc: bad argument");
    hipLaunchKernelGGL(embed_posenc_kernel, dim3(as_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, tokens, (long)tok_stride, table,
                       pe, out, (long)rows, T, D);
    AS_LAUNCH_CHECK("as_embed_posenc");
    return 0;
}

extern "C" int as_add(const float* a, const float* b, float* dst, int64_t n, void* stream) {
    AS_REQUIRE(a && dst && n > 0, AS_ERR_BAD_ARG, "as_add: bad argument");
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, dst, (long)n);
    AS_LAUNCH_CHECK("as_add");
    return 0;
}

extern "C" int as_row_scale(const float* a, const float* row_scale, float* dst, int64_t rows, int32_t row_len, void* stream) {
    AS_REQUIRE(a && row_scale && dst && rows > 0 && row_len > 0, AS_ERR_BAD_ARG, "as_row_scale: bad argument");
    const long n = (long)rows * row_len;
    hipLaunchKernelGGL(row_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, row_scale, dst, n, row_len);
    AS_LAUNCH_CHECK("as_row_scale");
    return 0;
}

extern "C" int as_dropout_fwd(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
    AS_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, AS_ERR_BAD_ARG, "as_dropout_fwd: bad argument");
    return as_dropout(x, y, (long)n, p, seed, (hipStream_t)stream);
}

extern "C" int as_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                            void* stream) {
    AS_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, AS_ERR_BAD_ARG, "as_adam_step: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
    AS_LAUNCH_CHECK("as_adam_step");
    return 0;
}

// ---- streaming copy probe (bench.py: the box's measured HBM copy rate, the denominator beside the 8 TB/s specification):
// float4 grid-stride loop, 16 B per lane per access, non-temporal loads and stores are left to the cache policy's default.
namespace {
__global__ __launch_bounds__(256) void copy_f32x4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
    // eight 16-byte loads in flight per lane before the first store (MI355X_MICROARCH.md: >= 8 outstanding loads per lane)
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
}  // namespace
extern "C" int as_copy_f32(const float* src, float* dst, int64_t n, void* stream) {
    AS_REQUIRE(src && dst && n > 0 && n % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
               AS_ERR_BAD_ARG, "as_copy_f32: n %% 4 == 0, 16-byte aligned buffers");
    const long n4 = n / 4;
    const long blocks = std::min<long>((n4 + 255) / 256, 256L * 16);   // 16 workgroups per CU
    hipLaunchKernelGGL(copy_f32x4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), n4);
    AS_LAUNCH_CHECK("as_copy_f32");
    return 0;
}
