// Kernels of the DeepSpeech2-style articulatory scorer (phoneme_recognition/deepspeech2.py:15-47, 159-195) on
// channels-last feature maps  x[b][t][d][c]  (c innermost: the 32 channels of a position are one 128-byte line, and
// the D*32 features of a frame are one contiguous row -- the `view(B, C*D, T).permute(2, 0, 1)` before the scorer's
// Linear (deepspeech2.py:183-186) becomes a plain row-major GEMM over a column-permuted weight):
//   conv3x3_halo_kernel  : Conv2d(32 -> 32, 3x3, stride 1, padding 1) as an implicit GEMM on the fp32 matrix core:
//                          M = positions (b, t, d), N = 32 output channels, K = 9 taps x 32 input channels.  A workgroup
//                          walks tiles of 128 consecutive (t, d) positions of one utterance; the tile plus its halo
//                          (D + 1 positions either side) is staged in LDS ONCE and all nine taps read it at shifted
//                          offsets (feature-axis edges point at a zero row); the 9 x 32 x 32 weights live in registers
//                          as MFMA B fragments for the whole (persistent) workgroup.  Optional residual input fused
//                          into the epilogue (ResidualCNN's `out += x`).
//   conv3x3_mfma_kernel  : same contraction, one K tile per tap re-staged from global memory; any D (fallback when the
//                          halo does not fit in LDS).
//   conv3x3_small_kernel : the stem Conv2d(C_in -> 32) for small C_in (2 coordinate planes) + optional per-(b, t) voicing
//                          bias (deepspeech2.py:176-178), plain FMAs; reads the planar input through explicit strides.
//   ln_feat_gelu_kernel  : LayerNorm over the FEATURE axis d (the reference transposes (B, C, D, T) -> (B, C, T, D),
//                          normalises, transposes back: deepspeech2.py:31-34) followed by exact GELU, in place of four
//                          transposes and two elementwise passes; coalesced over c.
#include "as_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int CO = 32;  // output channels of every convolution of the scorer (deepspeech2.py:104)

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ res,
                                                           float* __restrict__ y, int B, int D, int T) {
    constexpr int CI = 32, BM = 128, LDW = CI + 1;
    __shared__ __attribute__((aligned(16))) float sA[BM * LDW];
    __shared__ __attribute__((aligned(16))) float sB[CO * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const long P = (long)B * D * T;
    const long p0 = (long)blockIdx.x * BM;
    // this thread stages 4 rows (positions) x one float4 of channels per tap
    const int kq = tid & 7, row0 = tid >> 3;
    int pd[4], pt[4];
    long pbase[4];
    bool pok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long p = p0 + row0 + 32 * i;
        pok[i] = p < P;
        const long pp = pok[i] ? p : 0;
        pd[i] = (int)(pp % D);
        pt[i] = (int)((pp / D) % T);
        pbase[i] = pp * CI;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int kd = tap / 3 - 1, kt = tap % 3 - 1;
        float4 ra[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = pd[i] + kd, t = pt[i] + kt;
            const bool ok = pok[i] && d >= 0 && d < D && t >= 0 && t < T;
            const float4 v = *reinterpret_cast<const float4*>(x + (ok ? pbase[i] + ((long)kt * D + kd) * CI + kq * 4 : 0L));
            ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 rb = *reinterpret_cast<const float4*>(w + ((long)tap * CO + row0) * CI + kq * 4);  // w[tap][co][ci]
        __syncthreads();  // previous tap's MFMAs are done with the images
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* dst = sA + (row0 + 32 * i) * LDW + kq * 4;
            dst[0] = ra[i].x; dst[1] = ra[i].y; dst[2] = ra[i].z; dst[3] = ra[i].w;
        }
        {
            float* dst = sB + row0 * LDW + kq * 4;
            dst[0] = rb.x; dst[1] = rb.y; dst[2] = rb.z; dst[3] = rb.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < CI; kk += 2) {
            const float av = sA[(wave * 32 + l31) * LDW + kk + lh];
            const float bv = sB[l31 * LDW + kk + lh];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    const float bj = bias[l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long p = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (p < P) {
            float v = acc[r] + bj;
            if (res) v += res[p * CO + l31];
            y[p * CO + l31] = v;
        }
    }
}

// see the file header.  LDS image: halo position h (flattened q = q0 - D - 1 + h) at sA[h * 33 + c], NHP >= NH rows so that
// every staging pass is a full one; row NHP is zeros.  The accumulators start from the skip input, loaded before the halo
// is staged so that its latency hides behind the staging loads; the first MFMA is the first consumer.
template <bool HAS_RES>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ res,
                                                              float* __restrict__ y, int B, int D, int T, int tiles_per_b, int NHP) {
    constexpr int CI = 32, BM = 128, LDW = CI + 1, STAGE = 5;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int TD = T * D;
    // B fragments: lane (l31, lh) holds w[tap][co = l31][ci = 2 * j + lh]; the 36 KB of weights pass through LDS once
    // (coalesced float4 reads, rows padded to 33) instead of 144 line-per-lane gathers
    float bw[9][16];
    for (int e = tid; e < 9 * CO * CI / 4; e += 256) {
        const float4 v = reinterpret_cast<const float4*>(w)[e];
        float* dst = sA + (e >> 3) * LDW + (e & 7) * 4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < 16; ++j) bw[tap][j] = sA[(tap * CO + l31) * LDW + 2 * j + lh];
    const float bj = bias[l31];
    __syncthreads();
    if (tid < LDW) sA[NHP * LDW + tid] = 0.f;
    const int total = B * tiles_per_b;
    const int sh = tid >> 3, sc4 = (tid & 7) * 4;  // staging: halo row (+32 per pass) and channel quad of this thread
    const int r = wave * 32 + l31;                 // this lane's A row (position q0 + r)
    const int ro = wave * 32 + 4 * lh;             // first of this lane's 16 C rows: ro + (i & 3) + 8 * (i >> 2)
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int q0 = (tile - b * tiles_per_b) * BM;
        const float* xb = x + (long)b * TD * CI;
        const float* rb = res + (long)b * TD * CO;             // wave-uniform bases + 32-bit lane offsets (saddr addressing)
        float* yb = y + (long)b * TD * CO;
        const unsigned ob = (unsigned)(q0 + ro) * CO + l31;    // C row i of this lane sits at ob + ((i & 3) + 8 * (i >> 2)) * CO
        const int nq = TD - q0 - ro;                           // rows of this lane that exist: (i & 3) + 8 * (i >> 2) < nq
        const bool whole = q0 + BM <= TD;  // wave-uniform: whole tiles use immediate-offset loads / stores, no predicates
        f32x16 acc;
        if (whole) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = HAS_RES ? rb[ob + ((i & 3) + 8 * (i >> 2)) * CO] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = (i & 3) + 8 * (i >> 2);
                acc[i] = HAS_RES && k < nq ? rb[ob + k * CO] : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // skip-input loads first, the halo loads queue up behind them
        __syncthreads();                    // the previous tile's fragment reads are done
        for (int h0 = sh; h0 < NHP; h0 += 32 * STAGE) {  // STAGE loads in flight per thread, then their LDS writes
            float4 v[STAGE];
#pragma unroll
            for (int i = 0; i < STAGE; ++i) {
                const int q = q0 - D - 1 + h0 + 32 * i;
                const bool ok = q >= 0 && q < TD;
                v[i] = *reinterpret_cast<const float4*>(xb + (unsigned)((ok ? q : 0) * CI + sc4));
                if (!ok) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < STAGE; ++i) {
                float* dst = sA + (h0 + 32 * i) * LDW + sc4;
                dst[0] = v[i].x; dst[1] = v[i].y; dst[2] = v[i].z; dst[3] = v[i].w;
            }
        }
        __syncthreads();
        const int d = (q0 + r) % D;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kd = tap / 3 - 1, kt = tap % 3 - 1;
            int h = r + D + 1 + kt * D + kd;
            if (kd == -1) h = d == 0 ? NHP : h;  // feature-axis edge: the neighbour is padding, not the adjacent frame's row
            if (kd == 1) h = d == D - 1 ? NHP : h;
            const float* ap = sA + h * LDW + lh;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * j], bw[tap][j], acc, 0, 0, 0);
        }
        if (whole) {
#pragma unroll
            for (int i = 0; i < 16; ++i) yb[ob + ((i & 3) + 8 * (i >> 2)) * CO] = acc[i] + bj;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = (i & 3) + 8 * (i >> 2);
                if (k < nq) yb[ob + k * CO] = acc[i] + bj;
            }
        }
    }
}

// stem: x planar, element (b, ci, d, t) at x[b*sb + ci*sc + d*sd + t*st]; w [9][32][Cin]; one thread per (position, co)
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ voicing,
                                                            float* __restrict__ y, int B, int D, int T, int Cin, long sb, long sc,
                                                            long sd, long st) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long P = (long)B * D * T;
    if (idx >= P * CO) return;
    const long p = idx / CO;
    const int co = (int)(idx - p * CO);
    const int d = (int)(p % D), t = (int)((p / D) % T);
    const long b = p / ((long)D * T);
    float acc = 0.f;
    for (int ci = 0; ci < Cin; ++ci) {  // torch's direct convolution order is not defined; fp32 parity is by tolerance
        for (int tap = 0; tap < 9; ++tap) {
            const int dd = d + tap / 3 - 1, tt = t + tap % 3 - 1;
            if (dd < 0 || dd >= D || tt < 0 || tt >= T) continue;
            acc = fmaf(w[((long)tap * CO + co) * Cin + ci], x[b * sb + ci * sc + dd * sd + tt * st], acc);
        }
    }
    acc += bias[co];
    if (voicing) acc += voicing[b * T + t];
    y[idx] = acc;
}

// stem for a compile-time plane count: one thread per position computes all 32 output channels (weights are wave-uniform
// scalar loads), the tile's outputs are transposed through LDS so that the channels-last store is fully coalesced
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ voicing,
                                                           float* __restrict__ y, int B, int D, int T, long sb, long sc, long sd,
                                                           long st) {
    __shared__ float sOut[256 * (CO + 1)];
    const int tid = threadIdx.x;
    const long P = (long)B * D * T;
    const long p0 = (long)blockIdx.x * 256;
    const long p = p0 + tid;
    const bool live = p < P;
    const long pp = live ? p : 0;
    const int d = (int)(pp % D), t = (int)((pp / D) % T);
    const long b = pp / ((long)D * T);
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dd = d + tap / 3 - 1, tt = t + tap % 3 - 1;
            const bool ok = dd >= 0 && dd < D && tt >= 0 && tt < T;
            float xv = x[ok ? b * sb + ci * sc + dd * sd + tt * st : 0L];
            xv = ok ? xv : 0.f;
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[co] = fmaf(w[(tap * CO + co) * CIN + ci], xv, acc[co]);
        }
    }
    const float vb = voicing ? voicing[b * T + t] : 0.f;
#pragma unroll
    for (int co = 0; co < CO; ++co) sOut[tid * (CO + 1) + co] = acc[co] + bias[co] + vb;
    __syncthreads();
    const long n = (P - p0 < 256 ? P - p0 : 256) * CO;
#pragma unroll 4
    for (int i = tid; i < n; i += 256) y[p0 * CO + i] = sOut[(i >> 5) * (CO + 1) + (i & 31)];
}

// y[r][d][c] = gelu(LayerNorm_d(x[r][:, c]) * gamma[d] + beta[d]) for rows r = (b, t); one thread per (r, c) column, the
// column (D <= MAXD values) stays in registers between the statistics and the output pass
template <int MAXD>
__global__ __launch_bounds__(256) void ln_feat_gelu_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y, long R, int D,
                                                           int Cc, float eps) {
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    if (col >= R * Cc) return;
    const long r = col / Cc;
    const int c = (int)(col - r * Cc);
    const float* xp = x + r * D * Cc + c;
    float v[MAXD];
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) v[d] = xp[(long)(d < D ? d : D - 1) * Cc];  // branch-free: all loads in flight together
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
        if (d >= D) v[d] = 0.f;
        s += v[d];
    }
    const float mean = s / D;
    float q = 0.f;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
        const float e = v[d] - mean;
        q += d < D ? e * e : 0.f;
    }
    const float rs = 1.0f / sqrtf(q / D + eps);
    float* yp = y + r * D * Cc + c;
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
        if (d < D) yp[(long)d * Cc] = gelu_exact((v[d] - mean) * rs * gamma[d] + beta[d]);
}

// any D: three strided passes
__global__ __launch_bounds__(256) void ln_feat_gelu_loop_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y, long R,
                                                                int D, int Cc, float eps) {
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    if (col >= R * Cc) return;
    const long r = col / Cc;
    const int c = (int)(col - r * Cc);
    const float* xp = x + r * D * Cc + c;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += xp[(long)d * Cc];
    const float mean = s / D;
    float q = 0.f;
    for (int d = 0; d < D; ++d) {
        const float e = xp[(long)d * Cc] - mean;
        q += e * e;
    }
    const float rs = 1.0f / sqrtf(q / D + eps);
    float* yp = y + r * D * Cc + c;
    for (int d = 0; d < D; ++d) yp[(long)d * Cc] = gelu_exact((xp[(long)d * Cc] - mean) * rs * gamma[d] + beta[d]);
}

__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = gelu_exact(x[i]);
}

}  // namespace

extern "C" int as_conv3x3_c32(const float* x, const float* w, const float* bias, const float* res, float* y, int32_t B,
                              int32_t T, int32_t D, void* stream) {
    AS_REQUIRE(x && w && bias && y && B > 0 && D > 0 && T > 0, AS_ERR_BAD_ARG, "as_conv3x3_c32: bad argument");
    AS_REQUIRE((long)T * D * 32 < (1L << 31), AS_ERR_UNSUPPORTED, "as_conv3x3_c32: T * D = %ld too large for 32-bit frame offsets", (long)T * D);
    const long P = (long)B * D * T;
    const int nhp = (128 + 2 * D + 2 + 159) / 160 * 160;  // halo rows rounded up to whole staging passes (32 rows x 5)
    size_t halo_bytes = (size_t)(nhp + 1) * 33 * sizeof(float);
    if (halo_bytes <= 64 * 1024) {
        if (halo_bytes < 9 * 32 * 33 * sizeof(float)) halo_bytes = 9 * 32 * 33 * sizeof(float);  // the weights pass through it first
        const int tiles_per_b = as_cdiv((long)T * D, 128);
        const long tiles = (long)B * tiles_per_b;
        const dim3 grid((int)(tiles < 512 ? tiles : 512));  // 2 resident workgroups per CU walk the tiles
        if (res)
            hipLaunchKernelGGL(conv3x3_halo_kernel<true>, grid, dim3(256), halo_bytes, (hipStream_t)stream, x, w, bias, res, y, B, D, T,
                               tiles_per_b, nhp);
        else
            hipLaunchKernelGGL(conv3x3_halo_kernel<false>, grid, dim3(256), halo_bytes, (hipStream_t)stream, x, w, bias, res, y, B, D, T,
                               tiles_per_b, nhp);
    } else {
        hipLaunchKernelGGL(conv3x3_mfma_kernel, dim3(as_cdiv(P, 128)), dim3(256), 0, (hipStream_t)stream, x, w, bias, res, y, B, D, T);
    }
    AS_LAUNCH_CHECK("as_conv3x3_c32");
    return 0;
}

extern "C" int as_conv3x3_stem(const float* x, int64_t sb, int64_t sc, int64_t sd, int64_t st, const float* w, const float* bias,
                               const float* voicing, float* y, int32_t B, int32_t T, int32_t D, int32_t Cin, void* stream) {
    AS_REQUIRE(x && w && bias && y && B > 0 && D > 0 && T > 0 && Cin > 0, AS_ERR_BAD_ARG, "as_conv3x3_stem: bad argument");
    const long P = (long)B * D * T;
    hipStream_t s = (hipStream_t)stream;
#define AS_STEM(CIN)                                                                                                              \
    hipLaunchKernelGGL(conv3x3_stem_kernel<CIN>, dim3(as_cdiv(P, 256)), dim3(256), 0, s, x, w, bias, voicing, y, B, D, T, (long)sb, \
                       (long)sc, (long)sd, (long)st)
    switch (Cin) {
        case 1: AS_STEM(1); break;
        case 2: AS_STEM(2); break;
        case 3: AS_STEM(3); break;
        default:
            hipLaunchKernelGGL(conv3x3_small_kernel, dim3(as_cdiv(P * CO, 256)), dim3(256), 0, s, x, w, bias, voicing, y, B, D, T, Cin,
                               (long)sb, (long)sc, (long)sd, (long)st);
    }
#undef AS_STEM
    AS_LAUNCH_CHECK("as_conv3x3_stem");
    return 0;
}

extern "C" int as_ln_feat_gelu(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t D, int32_t Cc,
                               void* stream) {
    AS_REQUIRE(x && gamma && beta && y && rows > 0 && D > 0 && Cc > 0, AS_ERR_BAD_ARG, "as_ln_feat_gelu: bad argument");
    const dim3 grid(as_cdiv((long)rows * Cc, 256));
    hipStream_t st = (hipStream_t)stream;
    if (D <= 80)
        hipLaunchKernelGGL(ln_feat_gelu_kernel<80>, grid, dim3(256), 0, st, x, gamma, beta, y, (long)rows, D, Cc, 1e-5f);
    else if (D <= 128)
        hipLaunchKernelGGL(ln_feat_gelu_kernel<128>, grid, dim3(256), 0, st, x, gamma, beta, y, (long)rows, D, Cc, 1e-5f);
    else
        hipLaunchKernelGGL(ln_feat_gelu_loop_kernel, grid, dim3(256), 0, st, x, gamma, beta, y, (long)rows, D, Cc, 1e-5f);
    AS_LAUNCH_CHECK("as_ln_feat_gelu");
    return 0;
}

extern "C" int as_gelu(const float* x, float* y, int64_t n, void* stream) {
    AS_REQUIRE(x && y && n > 0, AS_ERR_BAD_ARG, "as_gelu: bad argument");
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gelu_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
    AS_LAUNCH_CHECK("as_gelu");
    return 0;
}
