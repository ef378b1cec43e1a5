// Loss / metric / DSP kernels of the hot path -- all HBM-bound streaming or tiny-tile work:
//   EuclideanDistance (+ fused length-masked mean and its gradient), MeanP2CPDistance tiles,
//   P2CPDistance utterance means, tract variables (min pairwise distance + closest pair) and the
//   vocal-tract area function (fp64).  One wave per tile/frame where a tile is small; coalesced
//   4-byte lanes over the (.., 2, N) contour rows (N = 50 floats = 200 B rows are only 8-byte aligned).
#include "gemm_internal.h"

namespace {

// ---------------------------------------------------------------- EuclideanDistance ("none")
__global__ __launch_bounds__(256) void euclid_fwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                         long points, int N, float* __restrict__ dist) {
    const long stride = (long)gridDim.x * 256;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < points; p += stride) {
        const long row = p / N;  // (frame, articulator)
        const int n = (int)(p - row * N);
        const long base = row * 2 * N + n;
        const float dx = out[base] - tgt[base], dy = out[base + N] - tgt[base + N];
        dist[p] = sqrtf(dx * dx + dy * dy);
    }
}
__global__ __launch_bounds__(256) void euclid_bwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                         const float* __restrict__ ddist, long points, int N,
                                                         float* __restrict__ dout) {
    const long stride = (long)gridDim.x * 256;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < points; p += stride) {
        const long row = p / N;
        const int n = (int)(p - row * N);
        const long base = row * 2 * N + n;
        const float dx = out[base] - tgt[base], dy = out[base + N] - tgt[base + N];
        const float g = ddist[p] / sqrtf(dx * dx + dy * dy);  // NaN at zero distance, as torch autograd
        dout[base] = dx * g;
        dout[base + N] = dy * g;
    }
}

// ---------------------------------------------------------------- fused masked mean + gradient
constexpr int LOSS_BLOCKS = 1024;

template <bool PRESIG>
__global__ __launch_bounds__(256) void euclid_masked_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                            long tgt_T, const int* __restrict__ lengths, int T, int A,
                                                            int N, long points, float scale, float* __restrict__ dout,
                                                            float* __restrict__ partial) {
    __shared__ float red[4];
    const long stride = (long)gridDim.x * 256;
    const int AN = A * N;
    float s = 0.f;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < points; p += stride) {
        const long frame = p / AN;
        const int rem = (int)(p - frame * AN);
        const int a = rem / N, n = rem - a * N;
        const long b = frame / T;
        const int t = (int)(frame - b * T);
        const long ob = (frame * A + a) * 2 * N + n;
        if (t < lengths[b]) {
            const long tb = ((b * tgt_T + t) * A + a) * 2 * N + n;
            const float ox = out[ob], oy = out[ob + N];
            const float dx = ox - tgt[tb], dy = oy - tgt[tb + N];
            const float d = sqrtf(dx * dx + dy * dy);
            s += d;
            if (dout) {
                const float g = scale / d;
                // PRESIG: through the model's final sigmoid as well (same product order as sigmoid_bwd_kernel)
                dout[ob] = PRESIG ? dx * g * ox * (1.f - ox) : dx * g;
                dout[ob + N] = PRESIG ? dy * g * oy * (1.f - oy) : dy * g;
            }
        } else if (dout) {
            dout[ob] = 0.f;
            dout[ob + N] = 0.f;
        }
    }
    s = as_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ partial, int n, float scale,
                                                         float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = as_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

// ---------------------------------------------------------------- MeanP2CPDistance
// one wave per (u, v) tile: both point sets staged in LDS; lane i scans all v for u_i (row minima),
// lane j scans all u for v_j (column minima).  min over squared distances, sqrt once (monotone).
constexpr int P2CP_MAXPTS = 256;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// min_j |p - q_j|^2 over the n points (qx, qy) in LDS (padded to a multiple of 4 with +inf coordinates): four points per
// step from two broadcast ds_read_b128, the arithmetic on float pairs (v_pk_add / v_pk_mul: half the instructions of the
// scalar form), v_min3 to fold two candidates at once.  The kernel lives on vector-instruction issue (50 x 50 pair
// distances twice per 800-byte tile), not on HBM: ~3.5 instructions per pair instead of ~8.
__device__ __forceinline__ float p2cp_scan(float px, float py, const float* __restrict__ qx, const float* __restrict__ qy, int n4) {
    const f32x2 px2 = {px, px}, py2 = {py, py};
    float m = INFINITY;
    for (int j = 0; j < n4; j += 4) {
        const float4 x4 = *reinterpret_cast<const float4*>(qx + j), y4 = *reinterpret_cast<const float4*>(qy + j);
        const f32x2 dxa = px2 - f32x2{x4.x, x4.y}, dxb = px2 - f32x2{x4.z, x4.w};
        const f32x2 dya = py2 - f32x2{y4.x, y4.y}, dyb = py2 - f32x2{y4.z, y4.w};
        const f32x2 sa = dxa * dxa + dya * dya, sb = dxb * dxb + dyb * dyb;
        m = fminf(fminf(m, sa.x), sa.y);
        m = fminf(fminf(m, sb.x), sb.y);
    }
    return m;
}

__global__ __launch_bounds__(256) void p2cp_kernel(const float* __restrict__ u, long u_tile, long u_pt, long u_xy, int nu,
                                                   const float* __restrict__ v, long v_tile, long v_pt, long v_xy, int nv,
                                                   long tiles, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + wave;
    const int nu4 = (nu + 3) & ~3, nv4 = (nv + 3) & ~3;   // padded with +inf: a padding point is never a minimum
    float* ux = smem + (long)wave * 2 * (nu4 + nv4);
    float* uy = ux + nu4;
    float* vx = uy + nu4;
    float* vy = vx + nv4;
    if (tile < tiles) {
        const float* up = u + tile * u_tile;
        const float* vp = v + tile * v_tile;
        for (int i = lane; i < nu4; i += 64) { ux[i] = i < nu ? up[i * u_pt] : INFINITY; uy[i] = i < nu ? up[i * u_pt + u_xy] : INFINITY; }
        for (int i = lane; i < nv4; i += 64) { vx[i] = i < nv ? vp[i * v_pt] : INFINITY; vy[i] = i < nv ? vp[i * v_pt + v_xy] : INFINITY; }
    }
    __syncthreads();
    if (tile >= tiles) return;
    float su = 0.f, sv = 0.f;
    for (int i = lane; i < nu; i += 64) su += sqrtf(p2cp_scan(ux[i], uy[i], vx, vy, nv4));   // row minima
    for (int j = lane; j < nv; j += 64) sv += sqrtf(p2cp_scan(vx[j], vy[j], ux, uy, nu4));   // column minima
    su = as_wave_sum(su);
    sv = as_wave_sum(sv);
    if (lane == 0) out[tile] = (su / nu + sv / nv) * 0.5f;
}

// P2CPDistance: mean_b( mean_{t<len_b, a} p2cp[b][t][a] * to_mm ); one block, wave per utterance
__global__ __launch_bounds__(256) void p2cp_utt_mean_kernel(const float* __restrict__ p2cp, const int* __restrict__ lengths,
                                                            int B, int T, int A, float to_mm, float* __restrict__ result) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int b = wave; b < B; b += 4) {
        const int n = lengths[b] * A;
        const float* p = p2cp + (long)b * T * A;
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += p[i] * to_mm;
        s = as_wave_sum(s);
        acc += s / n;
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) result[0] = ((red[0] + red[1]) + (red[2] + red[3])) / B;
}

// ---------------------------------------------------------------- tract variables
// Correctly rounded fp32 square root from v_sqrt_f32 (1 ulp) + two fused residual tests against the neighbouring floats
// (the sequence LLVM emits for an IEEE sqrtf on this target, written out so that it does not depend on compiler flags):
// bit-identical to sqrtf on the host, which is what makes the arg-min pairs reproducible.  (Round 2 went through
// v_sqrt_f64: ~3x the instructions.)  s >= 0, finite; tiny arguments are scaled out of the denormal range first.
__device__ __forceinline__ float as_sqrt_rn(float s) {
    const bool tiny = s < 0x1p-96f;
    const float x = tiny ? s * 0x1p+32f : s;
    float r = __builtin_amdgcn_sqrtf(x);
    const float dn = __int_as_float(__float_as_int(r) - 1), up = __int_as_float(__float_as_int(r) + 1);
    const float vp = __builtin_fmaf(-dn, r, x), vs = __builtin_fmaf(-up, r, x);
    r = vp <= 0.f ? dn : r;
    r = vs > 0.f ? up : r;
    r = tiny ? r * 0x1p-16f : r;
    return (x == 0.f || x == INFINITY) ? x : r;
}

// One workgroup per FRAME, one wave per variable (LA, TTCD, TBCD, VEL: spec rows).  The articulator rows the four variables
// slice are staged in LDS once per frame (coalesced 4-byte lanes over the (A, 2, N) rows), so every pair distance reads
// broadcast LDS words instead of global memory.  Lane j owns arr2 point j and scans arr1; the result must be torch's
// min(dim=0) of the DISTANCES: the first index whose correctly rounded sqrt(dx^2 + dy^2) is minimal.  The kernel lives on
// vector-instruction issue (28 M pair distances per 6400 frames), so the scan is arranged for few instructions per pair:
//   pass 1  minimum of the SQUARED distances, two arr1 points per step on float pairs (v_pk_add / v_pk_mul, v_min3): sqrt
//           is monotone, so d_min = sqrt_rn(s_min) -- one square root per lane instead of one per pair;
//   s_hi    the largest float whose correctly rounded root is still d_min (at most three floats share a root);
//   pass 2  the first i with s_i <= s_hi: exactly the first index whose distance equals d_min, ties of distinct squared
//           distances that round to the same root included.
// Squared distances use un-contracted fp32 ops (sub, mul, mul, add), component for component what the scalar formula gives:
// values and arg-min pairs are bit-identical to the element-wise evaluation on the host.  Then the wave takes the first
// minimum over j.  (Round 2: one wave per (frame, variable), operands from global memory, a v_sqrt_f64 per pair.)
constexpr int TV_MAXPTS = 64;   // points per slice (the reference's slices hold 15 .. 50)

__global__ __launch_bounds__(256) void tv_kernel(const float* __restrict__ contours, long frames, int A, int N,
                                                 const int* __restrict__ spec, int n_tv, float* __restrict__ values,
                                                 float* __restrict__ poc1, float* __restrict__ poc2, int* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) float fr_s[];   // [A][2][N] the frame, then [4][2][TV_MAXPTS] arr1 copies
    const long f = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* fr = contours + f * (long)A * 2 * N;
    // only the articulators some variable slices (6 of 11 in the reference's configuration: 2.4 of 4.4 KB per frame)
    unsigned need = 0;
    for (int v = 0; v < n_tv; ++v)
        for (int q = 0; q < 3; ++q) {
            const int c = spec[v * 9 + 3 * q];
            if (c >= 0 && c < 32) need |= 1u << c;
        }
    for (int i = threadIdx.x; i < A * 2 * N; i += 256) {
        const int c = i / (2 * N);
        if (c >= 32 || ((need >> c) & 1u)) fr_s[i] = fr[i];
    }
    __syncthreads();
    float* ax = fr_s + ((A * 2 * N + 3) & ~3) + wave * 2 * TV_MAXPTS;   // this wave's arr1, 8-byte aligned pairs, padded with +inf
    float* ay = ax + TV_MAXPTS;
    for (int v = wave; v < n_tv; v += 4) {
        const int* sp = spec + v * 9;
        const int c1 = sp[0], s1 = sp[1], n1 = sp[2] - sp[1];
        const int c2a = sp[3], s2a = sp[4], n2a = sp[5] - sp[4];
        const int c2b = sp[6], s2b = sp[7], n2b = c2b >= 0 ? sp[8] - sp[7] : 0;
        const int n2 = n2a + n2b;
        const float* x1 = fr_s + c1 * 2 * N + s1;
        const int n1e = (n1 + 1) & ~1;
        __builtin_amdgcn_wave_barrier();
        if (lane < n1e) {
            ax[lane] = lane < n1 ? x1[lane] : INFINITY;
            ay[lane] = lane < n1 ? x1[N + lane] : INFINITY;
        }
        __builtin_amdgcn_wave_barrier();
        float best = INFINITY;
        int bi = 0, bj = 0x7fffffff;
        float bx2 = 0.f, by2 = 0.f;
        for (int j = lane; j < n2; j += 64) {
            const float* p2 = j < n2a ? fr_s + c2a * 2 * N + s2a + j : fr_s + c2b * 2 * N + s2b + (j - n2a);
            const float qx = p2[0], qy = p2[N];
            const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy};
            float smin = INFINITY;
            for (int i = 0; i < n1e; i += 2) {
                const f32x2 dx = *reinterpret_cast<const f32x2*>(ax + i) - qx2, dy = *reinterpret_cast<const f32x2*>(ay + i) - qy2;
                const f32x2 sq = dx * dx + dy * dy;
                smin = fminf(fminf(smin, sq.x), sq.y);
            }
            const float m = as_sqrt_rn(smin);
            float s_hi = smin;
#pragma unroll
            for (int k = 0; k < 3; ++k) {   // at most three consecutive floats share a correctly rounded root
                const float nxt = __int_as_float(__float_as_int(s_hi) + 1);
                s_hi = (s_hi < INFINITY && as_sqrt_rn(nxt) == m) ? nxt : s_hi;
            }
            int mi = 0;
            bool found = false;
            for (int i = 0; i < n1e; i += 2) {
                const f32x2 dx = *reinterpret_cast<const f32x2*>(ax + i) - qx2, dy = *reinterpret_cast<const f32x2*>(ay + i) - qy2;
                const f32x2 sq = dx * dx + dy * dy;
                const bool c0 = sq.x <= s_hi, c1b = sq.y <= s_hi;
                mi = found ? mi : (c0 ? i : (c1b ? i + 1 : mi));
                found = found || c0 || c1b;
            }
            if (m < best) { best = m; bi = mi; bj = j; bx2 = qx; by2 = qy; }  // j ascending per lane: first min kept
        }
        // wave arg-min with smallest-j tie break: (best, bj) travel, the winner's other fields are read out afterwards
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        const int wl = bj < 0x7fffffff ? (bj & 63) : 0;   // the lane that owns arr2 point bj
        const int wbi = __shfl(bi, wl, 64);
        const float wx = __shfl(bx2, wl, 64), wy = __shfl(by2, wl, 64);
        if (lane == 0) {
            const long item = f * n_tv + v;
            values[item] = best;
            poc1[item * 2] = x1[wbi];
            poc1[item * 2 + 1] = x1[N + wbi];
            poc2[item * 2] = wx;
            poc2[item * 2 + 1] = wy;
            if (idx) { idx[item * 2] = wbi; idx[item * 2 + 1] = bj; }
        }
    }
}

// ---------------------------------------------------------------- area function (fp64)
// one wave per frame.  Lanes compute mid points, radii and fx in parallel; the arc length is the
// SEQUENTIAL running sum of the reference's loop (lane 0), so dists is bit-identical to a serial
// fp64 evaluation.  un-contracted arithmetic throughout.
__global__ __launch_bounds__(256) void area_kernel(const double* __restrict__ wi, const double* __restrict__ we,
                                                   long frame_stride, long pt_stride, long xy_stride, long frames, int n,
                                                   double alpha, double beta, int beta_is_two, double* __restrict__ dists,
                                                   double* __restrict__ fx) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long f = (long)blockIdx.x * 4 + wave;
    double* mx = dsm + (long)wave * 2 * n;
    double* my = mx + n;
    if (f < frames) {
        const double* a = wi + f * frame_stride;
        const double* b = we + f * frame_stride;
        for (int i = lane; i < n; i += 64) {
            const double ax = a[i * pt_stride], ay = a[i * pt_stride + xy_stride];
            const double bx = b[i * pt_stride], by = b[i * pt_stride + xy_stride];
            // (x / 2 as x * 0.5: a power-of-two scaling is exact either way, and an fp64 division is ~30 instructions --
            // the kernel is bound by its fp64 instruction count, not by the 4.8 KB it moves per frame)
            mx[i] = __dadd_rn(fmin(ax, bx), __dmul_rn(fabs(__dsub_rn(ax, bx)), 0.5));
            my[i] = __dadd_rn(fmin(ay, by), __dmul_rn(fabs(__dsub_rn(ay, by)), 0.5));
            const double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by);
            const double r = __dmul_rn(__dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))), 0.5);
            fx[f * n + i] = __dmul_rn(alpha, beta_is_two ? __dmul_rn(r, r) : pow(r, beta));
        }
    }
    __syncthreads();
    if (f >= frames) return;
    // The ordered running sum of the reference's loop (d_i = d_{i-1} + seg_i, same order: bit-identical), 64 points at a
    // time: lane l computes the segment length of point 64 c + l, then every lane runs the same chain on wave-uniform
    // operands read out of the owners' registers (v_readlane: no LDS round trip per step, the only dependent operation per
    // step is the add) and keeps the value that belongs to its own point; the row is stored with all lanes (round 2: lane 0
    // walked LDS and issued 100 single-lane 8-byte stores per frame).
    double d = 0.0;
    for (int c = 0; 64 * c < n; ++c) {
        const int i = lane + 64 * c;
        double seg = 0.0;                                   // seg of point 0 is 0.0: d_0 = 0
        if (i >= 1 && i < n) {
            const double dx = __dsub_rn(mx[i], mx[i - 1]), dy = __dsub_rn(my[i], my[i - 1]);
            seg = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
        }
        const int lim = min(64, n - 64 * c);
        double mine = 0.0;
        for (int l = 0; l < lim; ++l) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(seg), l), hi = __builtin_amdgcn_readlane(__double2hiint(seg), l);
            d = __dadd_rn(__hiloint2double(hi, lo), d);
            mine = l == lane ? d : mine;
        }
        if (i < n) dists[f * n + i] = mine;
    }
}

// ---------------------------------------------------------------- evenly spaced resampling of fx over x (fp64 in, fp32 out)
// out[f][0][s] = x_s = x0 + s * (x_last - x0) / (n - 1) (np.linspace), out[f][1][s] = piecewise-linear fx(x_s):
// the vertical-line / polyline intersection of area_function.py:145-159 for increasing x.  One thread per sample,
// binary search of the segment.
__global__ __launch_bounds__(256) void resample_kernel(const double* __restrict__ x, const double* __restrict__ fx, long frames,
                                                       int n_pts, int n_samples, float* __restrict__ out) {
    // one workgroup per frame: the frame's abscissae and values staged in LDS once (the binary search of every sample was a
    // chain of dependent global loads), then one thread per sample
    extern __shared__ __attribute__((aligned(16))) double rs[];   // [2][n_pts]
    const long f = blockIdx.x;
    double* xf = rs;
    double* yf = rs + n_pts;
    for (int i = threadIdx.x; i < n_pts; i += 256) {
        xf[i] = x[f * n_pts + i];
        yf[i] = fx[f * n_pts + i];
    }
    __syncthreads();
    const double x0 = xf[0], x1 = xf[n_pts - 1];
    const double step = n_samples > 1 ? __ddiv_rn(__dsub_rn(x1, x0), (double)(n_samples - 1)) : 0.0;
    for (int s = threadIdx.x; s < n_samples; s += 256) {
        const double xq = s == n_samples - 1 && n_samples > 1 ? x1 : __dadd_rn(x0, __dmul_rn((double)s, step));
        // largest i with xf[i] <= xq (clamped to the last segment)
        int lo = 0, hi = n_pts - 1;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (xf[mid] <= xq) lo = mid; else hi = mid;
        }
        double y;
        if (n_pts == 1 || xf[hi] == xf[lo]) y = yf[lo];
        else {
            const double slope = __ddiv_rn(__dsub_rn(yf[hi], yf[lo]), __dsub_rn(xf[hi], xf[lo]));
            y = __dadd_rn(__dmul_rn(slope, __dsub_rn(xq, xf[lo])), yf[lo]);
            if (xq >= xf[hi]) y = yf[hi];
        }
        out[(f * 2) * n_samples + s] = (float)xq;
        out[(f * 2 + 1) * n_samples + s] = (float)y;
    }
}

// ---------------------------------------------------------------- wall / semipolar-grid intersection (fp64)
// intersect_semipolar_grid (area_function.py:175-223): one wave per (frame, grid line).  The lanes test all (grid segment,
// wall segment) pairs of both walls (p + t r = q + u s, half-open parameter ranges so that a junction hit is reported
// once), hits are collected in LDS with their position along the grid line, lane 0 orders them along the line and applies
// the reference's selection: the wall's hit closest to the other wall's hits (first minimum of the distance matrix in
// row-major order), or -- when the other wall is not crossed -- closest to the EXTERNAL wall's end points, whose nearer
// end then stands in for the missing point (also in the external branch, :211).
constexpr int GI_MAXP = 16;  // hits kept per wall and grid line

struct GiHit { double key; double x, y; int ws; };

__device__ __forceinline__ int gi_argmin(const GiHit* a, int na, const double (*b)[2], int nb, int* j_out) {
    double best = 0.0;
    int bi = 0, bj = 0;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            const double dx = __dsub_rn(a[i].x, b[j][0]), dy = __dsub_rn(a[i].y, b[j][1]);
            const double d = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
            if ((i == 0 && j == 0) || d < best) best = d, bi = i, bj = j;
        }
    *j_out = bj;
    return bi;
}

__global__ __launch_bounds__(256) void grid_intersect_kernel(const double* __restrict__ air, const double* __restrict__ grid,
                                                             long frames, int Nw, int L, int G, int* __restrict__ flags,
                                                             double* __restrict__ p_int, double* __restrict__ p_ext) {
    __shared__ GiHit hits[4][2][GI_MAXP];
    __shared__ int cnt[4][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + wave;
    const bool live = item < frames * L;
    const long f = live ? item / L : 0;
    const int l = live ? (int)(item - f * L) : 0;
    if (lane < 2) cnt[wave][lane] = 0;
    __syncthreads();
    const double* gl = grid + (long)l * G * 2;
    const int pairs = (G - 1) * (Nw - 1);
    if (live) {
        for (int wall = 0; wall < 2; ++wall) {
            const double* wx = air + (f * 2 + wall) * 2 * Nw;  // x coordinates; y at + Nw
            const double* wy = wx + Nw;
            for (int idx = lane; idx < pairs; idx += 64) {
                const int gs = idx / (Nw - 1), ws = idx - gs * (Nw - 1);
                const double px = gl[2 * gs], py = gl[2 * gs + 1];
                const double rx = __dsub_rn(gl[2 * gs + 2], px), ry = __dsub_rn(gl[2 * gs + 3], py);
                const double qx = wx[ws], qy = wy[ws];
                const double sx = __dsub_rn(wx[ws + 1], qx), sy = __dsub_rn(wy[ws + 1], qy);
                const double den = __dsub_rn(__dmul_rn(rx, sy), __dmul_rn(ry, sx));
                if (den == 0.0) continue;
                const double dqx = __dsub_rn(qx, px), dqy = __dsub_rn(qy, py);
                const double t = __ddiv_rn(__dsub_rn(__dmul_rn(dqx, sy), __dmul_rn(dqy, sx)), den);
                const double u = __ddiv_rn(__dsub_rn(__dmul_rn(dqx, ry), __dmul_rn(dqy, rx)), den);
                const bool t_ok = 0.0 <= t && (t < 1.0 || (gs == G - 2 && t <= 1.0));
                const bool u_ok = 0.0 <= u && (u < 1.0 || (ws == Nw - 2 && u <= 1.0));
                if (t_ok && u_ok) {
                    const int slot = atomicAdd(&cnt[wave][wall], 1);
                    if (slot < GI_MAXP) {
                        GiHit h;
                        h.key = __dadd_rn((double)gs, t);
                        h.x = __dadd_rn(px, __dmul_rn(t, rx));
                        h.y = __dadd_rn(py, __dmul_rn(t, ry));
                        h.ws = ws;
                        hits[wave][wall][slot] = h;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!live || lane != 0) return;
    int n[2];
    int overflow = 0;
    for (int wall = 0; wall < 2; ++wall) {
        n[wall] = cnt[wave][wall];
        if (n[wall] > GI_MAXP) n[wall] = GI_MAXP, overflow = 4;
        GiHit* h = hits[wave][wall];
        for (int i = 1; i < n[wall]; ++i) {  // order along the grid line (then along the wall): insertion sort, <= 16 items
            const GiHit v = h[i];
            int j = i - 1;
            while (j >= 0 && (h[j].key > v.key || (h[j].key == v.key && h[j].ws > v.ws))) { h[j + 1] = h[j]; --j; }
            h[j + 1] = v;
        }
    }
    const double* ix = air + (f * 2 + 0) * 2 * Nw;
    const double* ex = air + (f * 2 + 1) * 2 * Nw;
    const double ends[2][2] = {{ex[0], ex[Nw]}, {ex[Nw - 1], ex[2 * Nw - 1]}};  // the external wall's first / last point
    const bool ic = n[0] > 0, ec = n[1] > 0;
    double oi[2] = {0.0, 0.0}, oe[2] = {0.0, 0.0};
    double other[GI_MAXP][2];
    if (ic) {
        int nb = 2, jm;
        if (ec) { nb = n[1]; for (int j = 0; j < nb; ++j) other[j][0] = hits[wave][1][j].x, other[j][1] = hits[wave][1][j].y; }
        const int im = gi_argmin(hits[wave][0], n[0], ec ? other : ends, nb, &jm);
        oi[0] = hits[wave][0][im].x; oi[1] = hits[wave][0][im].y;
        if (!ec) { const int e = jm ? Nw - 1 : 0; oe[0] = ex[e]; oe[1] = ex[Nw + e]; }
    }
    if (ec) {
        int nb = 2, jm;
        if (ic) { nb = n[0]; for (int j = 0; j < nb; ++j) other[j][0] = hits[wave][0][j].x, other[j][1] = hits[wave][0][j].y; }
        const int im = gi_argmin(hits[wave][1], n[1], ic ? other : ends, nb, &jm);
        oe[0] = hits[wave][1][im].x; oe[1] = hits[wave][1][im].y;
        if (!ic) { const int e = jm ? Nw - 1 : 0; oi[0] = ix[e]; oi[1] = ix[Nw + e]; }
    }
    flags[item] = (ic ? 1 : 0) | (ec ? 2 : 0) | overflow;
    p_int[2 * item] = oi[0]; p_int[2 * item + 1] = oi[1];
    p_ext[2 * item] = oe[0]; p_ext[2 * item + 1] = oe[1];
}

inline int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}


// ---------------------------------------------------------------- Pearson correlation over time (root metrics.py:9-35)
// One thread per (utterance, articulator, plane, point) column; consecutive threads read consecutive floats of a frame's
// [A][2][N] block, so every pass over T is coalesced.  Two passes: means, then centred sums (double accumulators).
// The x plane's TARGETS are centred with the x OUTPUTS' mean, as the reference does (metrics.py:22); y with its own (:30).
__global__ __launch_bounds__(256) void pearson_kernel(const float* __restrict__ out, long out_b, long out_t,
                                                      const float* __restrict__ tgt, long tgt_b, long tgt_t, int B, int T,
                                                      int A, int N, float eps, float* __restrict__ x_corr,
                                                      float* __restrict__ y_corr) {
    const long cols = (long)A * 2 * N;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * cols) return;
    const long b = idx / cols;
    const int c = (int)(idx - b * cols);
    const int a = c / (2 * N), plane = (c / N) & 1, n = c % N;
    const float* o = out + b * out_b + c;
    const float* g = tgt + b * tgt_b + c;
    double so = 0.0, sg = 0.0;
    for (int t = 0; t < T; ++t) {
        so += (double)o[(long)t * out_t];
        sg += (double)g[(long)t * tgt_t];
    }
    const float mo = (float)(so / T);
    const float mg = plane == 0 ? mo : (float)(sg / T);
    double sog = 0.0, soo = 0.0, sgg = 0.0;
    for (int t = 0; t < T; ++t) {
        const float vo = o[(long)t * out_t] - mo, vg = g[(long)t * tgt_t] - mg;
        sog += (double)vo * vg;
        soo += (double)vo * vo;
        sgg += (double)vg * vg;
    }
    const float r = (float)sog / (sqrtf((float)soo) * sqrtf((float)sgg) + eps);
    (plane == 0 ? x_corr : y_corr)[(b * A + a) * N + n] = r;
}

}  // namespace

extern "C" int as_euclid_fwd(const float* out, const float* tgt, int64_t frames, int32_t A, int32_t N, float* dist,
                             void* stream) {
    AS_REQUIRE(out && tgt && dist && frames > 0 && A > 0 && N > 0, AS_ERR_BAD_ARG, "as_euclid_fwd: bad argument");
    const long points = (long)frames * A * N;
    hipLaunchKernelGGL(euclid_fwd_kernel, dim3(ew_grid(points)), dim3(256), 0, (hipStream_t)stream, out, tgt, points, N, dist);
    AS_LAUNCH_CHECK("as_euclid_fwd");
    return 0;
}

extern "C" int as_euclid_bwd(const float* out, const float* tgt, const float* ddist, int64_t frames, int32_t A, int32_t N,
                             float* dout, void* stream) {
    AS_REQUIRE(out && tgt && ddist && dout && frames > 0 && A > 0 && N > 0, AS_ERR_BAD_ARG, "as_euclid_bwd: bad argument");
    const long points = (long)frames * A * N;
    hipLaunchKernelGGL(euclid_bwd_kernel, dim3(ew_grid(points)), dim3(256), 0, (hipStream_t)stream, out, tgt, ddist, points, N, dout);
    AS_LAUNCH_CHECK("as_euclid_bwd");
    return 0;
}

extern "C" int32_t as_euclid_masked_partials(void) { return LOSS_BLOCKS; }

// internal (gemm_internal.h): the final, fixed-order sum of workgroup partials written by another kernel
int as_loss_final(const float* partial, int n, float scale, float* loss, hipStream_t st) {
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, partial, n, scale, loss);
    AS_LAUNCH_CHECK("as_loss_final");
    return 0;
}

namespace {
template <bool PRESIG>
int euclid_masked_launch(const char* who, const float* out, const float* tgt, int64_t tgt_T, const int32_t* lengths, int32_t B, int32_t T,
                         int32_t A, int32_t N, float scale, float* loss, float* dout, float* partial, void* stream) {
    AS_REQUIRE(out && tgt && lengths && loss && partial, AS_ERR_BAD_ARG, "%s: null pointer", who);
    AS_REQUIRE(B > 0 && T > 0 && A > 0 && N > 0 && tgt_T >= T, AS_ERR_BAD_ARG, "%s: B=%d T=%d A=%d N=%d tgt_T=%ld", who, B, T, A, N,
               (long)tgt_T);
    const long points = (long)B * T * A * N;
    int blocks = ew_grid(points);
    if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(euclid_masked_kernel<PRESIG>, dim3(blocks), dim3(256), 0, st, out, tgt, (long)tgt_T, lengths, T, A, N, points,
                       scale, dout, partial);
    AS_LAUNCH_CHECK(who);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, partial, blocks, scale, loss);
    AS_LAUNCH_CHECK(who);
    return 0;
}
}  // namespace

extern "C" int as_euclid_masked_fwd_bwd(const float* out, const float* tgt, int64_t tgt_T, const int32_t* lengths, int32_t B,
                                        int32_t T, int32_t A, int32_t N, float scale, float* loss, float* dout,
                                        float* partial, void* stream) {
    return euclid_masked_launch<false>("as_euclid_masked_fwd_bwd", out, tgt, tgt_T, lengths, B, T, A, N, scale, loss, dout, partial,
                                       stream);
}

extern "C" int as_euclid_masked_fwd_bwd_presigmoid(const float* out, const float* tgt, int64_t tgt_T, const int32_t* lengths,
                                                   int32_t B, int32_t T, int32_t A, int32_t N, float scale, float* loss,
                                                   float* dout, float* partial, void* stream) {
    AS_REQUIRE(dout, AS_ERR_BAD_ARG, "as_euclid_masked_fwd_bwd_presigmoid: dout is required");
    return euclid_masked_launch<true>("as_euclid_masked_fwd_bwd_presigmoid", out, tgt, tgt_T, lengths, B, T, A, N, scale, loss, dout,
                                      partial, stream);
}

extern "C" int as_p2cp_fwd(const float* u, int64_t u_tile, int64_t u_pt, int64_t u_xy, int32_t n_u, const float* v,
                           int64_t v_tile, int64_t v_pt, int64_t v_xy, int32_t n_v, int64_t tiles, float* out,
                           void* stream) {
    AS_REQUIRE(u && v && out && tiles > 0, AS_ERR_BAD_ARG, "as_p2cp_fwd: bad argument");
    AS_REQUIRE(n_u > 0 && n_v > 0 && n_u <= P2CP_MAXPTS && n_v <= P2CP_MAXPTS, AS_ERR_UNSUPPORTED,
               "as_p2cp_fwd: point counts %d, %d must be in [1, %d]", n_u, n_v, P2CP_MAXPTS);
    const size_t shm = (size_t)4 * 2 * (((n_u + 3) & ~3) + ((n_v + 3) & ~3)) * sizeof(float);
    hipLaunchKernelGGL(p2cp_kernel, dim3(as_cdiv(tiles, 4)), dim3(256), shm, (hipStream_t)stream, u, (long)u_tile, (long)u_pt,
                       (long)u_xy, n_u, v, (long)v_tile, (long)v_pt, (long)v_xy, n_v, (long)tiles, out);
    AS_LAUNCH_CHECK("as_p2cp_fwd");
    return 0;
}

extern "C" int as_pearson_fwd(const float* out, int64_t out_b, int64_t out_t, const float* tgt, int64_t tgt_b, int64_t tgt_t,
                              int32_t B, int32_t T, int32_t A, int32_t N, float eps, float* x_corr, float* y_corr, void* stream) {
    AS_REQUIRE(out && tgt && x_corr && y_corr && B > 0 && T > 0 && A > 0 && N > 0, AS_ERR_BAD_ARG, "as_pearson_fwd: bad argument");
    const long total = (long)B * A * 2 * N;
    hipLaunchKernelGGL(pearson_kernel, dim3(as_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, (long)out_b, (long)out_t,
                       tgt, (long)tgt_b, (long)tgt_t, B, T, A, N, eps, x_corr, y_corr);
    AS_LAUNCH_CHECK("as_pearson_fwd");
    return 0;
}

extern "C" int as_p2cp_utterance_mean(const float* p2cp, const int32_t* lengths, int32_t B, int32_t T, int32_t A,
                                      float to_mm, float* result, void* stream) {
    AS_REQUIRE(p2cp && lengths && result && B > 0 && T > 0 && A > 0, AS_ERR_BAD_ARG, "as_p2cp_utterance_mean: bad argument");
    hipLaunchKernelGGL(p2cp_utt_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p2cp, lengths, B, T, A, to_mm, result);
    AS_LAUNCH_CHECK("as_p2cp_utterance_mean");
    return 0;
}

extern "C" int as_tract_variables_fwd(const float* contours, int64_t frames, int32_t A, int32_t N, const int32_t* spec,
                                      int32_t n_tv, float* values, float* poc1, float* poc2, int32_t* idx, void* stream) {
    AS_REQUIRE(contours && spec && values && poc1 && poc2 && frames > 0 && A > 0 && N > 0 && n_tv > 0, AS_ERR_BAD_ARG,
               "as_tract_variables_fwd: bad argument");
    const size_t shm = ((size_t)((A * 2 * N + 3) & ~3) + 4 * 2 * TV_MAXPTS) * sizeof(float);
    AS_REQUIRE(shm <= 64 * 1024 && frames < (1LL << 31), AS_ERR_UNSUPPORTED, "as_tract_variables_fwd: frame of %d x 2 x %d floats", A, N);
    AS_REQUIRE(N <= TV_MAXPTS, AS_ERR_UNSUPPORTED, "as_tract_variables_fwd: %d points per contour > %d", N, TV_MAXPTS);
    hipLaunchKernelGGL(tv_kernel, dim3((unsigned)frames), dim3(256), shm, (hipStream_t)stream, contours,
                       (long)frames, A, N, spec, n_tv, values, poc1, poc2, idx);
    AS_LAUNCH_CHECK("as_tract_variables_fwd");
    return 0;
}

extern "C" int as_area_function_fwd(const double* internal_wall, const double* external_wall, int64_t frame_stride,
                                    int64_t pt_stride, int64_t xy_stride, int64_t frames, int32_t n_pts, double alpha,
                                    double beta, double* dists, double* fx, void* stream) {
    AS_REQUIRE(internal_wall && external_wall && dists && fx && frames > 0, AS_ERR_BAD_ARG, "as_area_function_fwd: bad argument");
    AS_REQUIRE(n_pts > 0 && n_pts <= 1024, AS_ERR_UNSUPPORTED, "as_area_function_fwd: n_pts=%d must be in [1, 1024]", n_pts);
    const size_t shm = (size_t)4 * 2 * n_pts * sizeof(double);
    hipLaunchKernelGGL(area_kernel, dim3(as_cdiv(frames, 4)), dim3(256), shm, (hipStream_t)stream, internal_wall, external_wall,
                       (long)frame_stride, (long)pt_stride, (long)xy_stride, (long)frames, n_pts, alpha, beta,
                       beta == 2.0 ? 1 : 0, dists, fx);
    AS_LAUNCH_CHECK("as_area_function_fwd");
    return 0;
}

extern "C" int as_evenly_spaced_fx(const double* x, const double* fx, int64_t frames, int32_t n_pts, int32_t n_samples, float* out,
                                   void* stream) {
    AS_REQUIRE(x && fx && out && frames > 0 && n_pts > 0 && n_samples > 0, AS_ERR_BAD_ARG, "as_evenly_spaced_fx: bad argument");
    AS_REQUIRE(n_pts <= 4096 && frames < (1LL << 31), AS_ERR_UNSUPPORTED, "as_evenly_spaced_fx: n_pts=%d must be <= 4096", n_pts);
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)frames), dim3(256), (size_t)2 * n_pts * sizeof(double), (hipStream_t)stream, x, fx,
                       (long)frames, n_pts, n_samples, out);
    AS_LAUNCH_CHECK("as_evenly_spaced_fx");
    return 0;
}

extern "C" int as_intersect_semipolar_grid(const double* air_column, const double* grid, int64_t frames, int32_t n_pts, int32_t n_lines,
                                           int32_t grid_res, int32_t* flags, double* p_int, double* p_ext, void* stream) {
    AS_REQUIRE(air_column && grid && flags && p_int && p_ext && frames > 0 && n_pts >= 2 && n_lines > 0 && grid_res >= 2, AS_ERR_BAD_ARG,
               "as_intersect_semipolar_grid: bad argument");
    const long items = (long)frames * n_lines;
    hipLaunchKernelGGL(grid_intersect_kernel, dim3(as_cdiv(items, 4)), dim3(256), 0, (hipStream_t)stream, air_column, grid, (long)frames,
                       n_pts, n_lines, grid_res, flags, p_int, p_ext);
    AS_LAUNCH_CHECK("as_intersect_semipolar_grid");
    return 0;
}
