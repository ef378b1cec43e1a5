// Bidirectional LSTM recurrence (nn.LSTM semantics: the RNNType.LSTM switch of phoneme_to_articulation/__init__.py:47-49,
// used by principal_components/models/rnn.py:58-68) as persistent kernels, same design as gru.hip: one workgroup per
// (utterance, direction) walks its own sequence (packed-sequence semantics for free), W_hh (4H x H) lives in registers
// spread over 4 lanes per hidden unit (128 weight VGPRs per lane at H = 128), h_{t-1} in double-buffered LDS with one
// barrier per step, dot products finished with DPP quad permutes, operands loaded two steps ahead (three name-rotated sets, gru.hip).
//   i = s(gi_i + W_hi h + b_hi)  f = s(gi_f + ...)  g = tanh(gi_g + ...)  o = s(gi_o + ...)     (gate row order i, f, g, o)
//   c' = f c + i g,  h' = o tanh(c')
// The forward keeps i, f, g, o and c' per frame for the backward; the backward emits the pre-activation gradients (one
// array: input- and hidden-side pre-activations share them), weight gradients are time-batched GEMMs over them.
#include <cstdlib>
#include <type_traits>

#include "as_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    return v;
}

constexpr int LPU = 4;  // lanes per hidden unit

template <int H, bool TRAIN, bool TOK>
__global__ __launch_bounds__(LPU * H) void lstm_fwd_kernel(const float* __restrict__ gi, const int64_t* __restrict__ tokens,
                                                           long tok_stride, const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_hh, const int* __restrict__ lengths, int T,
                                                           float* __restrict__ y, float* __restrict__ gates) {
    constexpr int CW = 4 * LPU, NC = H / CW, NT = LPU * H;
    __shared__ __attribute__((aligned(16))) float hbuf[2][H];
    extern __shared__ int tok_s[];
    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, j = tid / LPU, q = tid % LPU;
    const int len = lengths[b];

    f32x2 w[4][NC * 2];
    {
        const float* wd = w_hh + (long)dir * 4 * H * H;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(wd + (long)(g * H + j) * H + CW * c + 4 * q);
                w[g][2 * c] = f32x2{v.x, v.y};
                w[g][2 * c + 1] = f32x2{v.z, v.w};
            }
    }
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = b_hh[dir * 4 * H + g * H + j] * (1.0f / LPU);   // a quarter per lane of the unit: see below

    for (long i = (long)len * H + tid; i < (long)T * H; i += NT)  // pad_packed_sequence: zeros at padded frames
        y[((long)b * T + i / H) * 2 * H + dir * H + (i % H)] = 0.f;
    if (tid < H) hbuf[0][tid] = 0.f;
    if (TOK)
        for (int t = tid; t < len; t += NT) tok_s[t] = (int)tokens[(long)b * tok_stride + t] * (8 * H);   // element offset of the row
    __syncthreads();
    if (len <= 0) return;

    const int t0 = dir ? len - 1 : 0;
    const int dt = dir ? -1 : 1;
    float* yb = y + dir * H + j;                    // + frame * 2H
    float* gb = gates + (long)dir * 5 * H + j;      // + frame * 10H, planes i, f, g, o, c' at + plane * H   (TRAIN)
    const float* gib = gi + (long)dir * 4 * H + j;  // + row * 8H
    const int m0 = q == 0 ? -1 : 0, m1 = q == 1 ? -1 : 0, m2 = q == 2 ? -1 : 0, m3 = q == 3 ? -1 : 0;

    float cst = 0.f;
    struct Gi { float x[4]; };
    // input projection of step u (clamped to the last step: the look-ahead stays inside the sequence)
    auto load_step = [&](int u) {
        const int uc = u < len ? u : len - 1;
        const int tu = t0 + uc * dt;
        const float* p = gib + (TOK ? (long)tok_s[tu] : ((long)b * T + tu) * 8 * H);
        Gi v;
#pragma unroll
        for (int g = 0; g < 4; ++g) v.x[g] = p[g * H];
        return v;
    };
    // One recurrent step: consumes `ci` (loaded two steps ago), starts the loads of step s + 2 into `fill` (gru.hip: the
    // operand sets rotate by NAME through a loop unrolled by three, so that nothing consumes a load in the step that issued it)
    auto step = [&](int s, const Gi& ci, Gi& fill) {
        const int cur = s & 1;
        const long fr = (long)b * T + t0 + (long)s * dt;
        fill = load_step(s + 2);
        const float4* hp = reinterpret_cast<const float4*>(hbuf[cur]);
        // the recurrent biases ride in as the accumulators' start value (gru.hip)
        f32x2 a[4] = {{bh[0], 0.f}, {bh[1], 0.f}, {bh[2], 0.f}, {bh[3], 0.f}};
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += 8) {  // 8 LDS reads in flight, then their FMAs
            float4 hv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < NC) hv[u] = hp[LPU * (c0 + u) + q];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < NC) {
                    const int c = c0 + u;
                    const f32x2 lo = {hv[u].x, hv[u].y}, hi = {hv[u].z, hv[u].w};
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        a[g] = __builtin_elementwise_fma(w[g][2 * c], lo, a[g]);
                        a[g] = __builtin_elementwise_fma(w[g][2 * c + 1], hi, a[g]);
                    }
                }
        }
        const float gi_ = as_sigmoid(ci.x[0] + quad_sum(a[0].x + a[0].y));
        const float gf = as_sigmoid(ci.x[1] + quad_sum(a[1].x + a[1].y));
        const float gg = as_tanh(ci.x[2] + quad_sum(a[2].x + a[2].y));
        const float go = as_sigmoid(ci.x[3] + quad_sum(a[3].x + a[3].y));
        cst = gf * cst + gi_ * gg;
        const float hnew = go * as_tanh(cst);
        hbuf[cur ^ 1][j] = hnew;  // the lanes of a unit hold identical values: all store the same word
        yb[fr * 2 * H] = hnew;
        if (TRAIN) {
            const int gv = (__float_as_int(gi_) & m0) | (__float_as_int(gf) & m1) | (__float_as_int(gg) & m2) | (__float_as_int(go) & m3);
            gb[fr * 10 * H + q * H] = __int_as_float(gv);
            gb[fr * 10 * H + 4 * H] = cst;
        }
        __syncthreads();
    };
    Gi a = load_step(0), bq = load_step(1), c;
    int s = 0;
    for (; s + 2 < len; s += 3) {
        step(s, a, c);
        step(s + 1, bq, a);
        step(s + 2, c, bq);
    }
    if (s < len) step(s, a, c);
    if (s + 1 < len) step(s + 1, bq, a);
}

template <int H>
__global__ __launch_bounds__(LPU * H) void lstm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ gates,
                                                           const float* __restrict__ w_hh, const int* __restrict__ lengths, int T,
                                                           float* __restrict__ dg) {
    constexpr int CW = 4 * LPU, NC = 4 * H / CW, NT = LPU * H;
    __shared__ __attribute__((aligned(16))) float gbuf[2][4 * H];
    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, k = tid / LPU, q = tid % LPU;
    const int len = lengths[b];

    f32x2 wt[NC * 2];  // W_hh^T: rows i = CW*c + 4q + ii of column k
    {
        const float* wd = w_hh + (long)dir * 4 * H * H;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            wt[2 * c] = f32x2{wd[(long)(CW * c + 4 * q) * H + k], wd[(long)(CW * c + 4 * q + 1) * H + k]};
            wt[2 * c + 1] = f32x2{wd[(long)(CW * c + 4 * q + 2) * H + k], wd[(long)(CW * c + 4 * q + 3) * H + k]};
        }
    }
    for (long i = (long)len * 4 * H + tid; i < (long)T * 4 * H; i += NT) {  // padded frames feed the time-batched GEMMs as zeros
        const long t = i / (4 * H), c = i % (4 * H);
        dg[(((long)b * T + t) * 2 + dir) * 4 * H + c] = 0.f;
    }
    if (len <= 0) return;

    const int t0 = dir ? 0 : len - 1;  // opposite to the forward walk
    const int dt = dir ? 1 : -1;
    const float* gtb = gates + (long)dir * 5 * H + k;  // + frame * 10H
    const float* dyb = dy + dir * H + k;               // + frame * 2H
    float* dgb = dg + (long)dir * 4 * H + k;           // + frame * 8H + plane * H
    const int m0 = q == 0 ? -1 : 0, m1 = q == 1 ? -1 : 0, m2 = q == 2 ? -1 : 0, m3 = q == 3 ? -1 : 0;
    struct In { float i, f, g, o, c, cprev, dyv; };
    // c_{prev} of frame t is the cell state of the frame this walk visits next (t + dt); zero beyond the sequence start
    auto load = [&](long fr, bool has_prev) {
        In v;
        const float* gp = gtb + fr * 10 * H;
        v.i = gp[0]; v.f = gp[H]; v.g = gp[2 * H]; v.o = gp[3 * H]; v.c = gp[4 * H];
        const float cp = gtb[(fr + (has_prev ? dt : 0)) * 10 * H + 4 * H];
        v.cprev = has_prev ? cp : 0.f;
        v.dyv = dyb[fr * 2 * H];
        return v;
    };
    long fr = (long)b * T + t0;
    float dh = 0.f, dc = 0.f;
    In cur_in = load(fr, len > 1);
    for (int s = 0; s < len; ++s) {
        const int cur = s & 1;
        const int adv = s + 1 < len ? dt : 0;
        const In nxt = load(fr + adv, s + 2 < len);
        const float dht = dh + cur_in.dyv;
        const float tc = as_tanh(cur_in.c);
        const float dct = dc + dht * cur_in.o * (1.f - tc * tc);
        const float p_o = dht * tc * cur_in.o * (1.f - cur_in.o);
        const float p_i = dct * cur_in.g * cur_in.i * (1.f - cur_in.i);
        const float p_f = dct * cur_in.cprev * cur_in.f * (1.f - cur_in.f);
        const float p_g = dct * cur_in.i * (1.f - cur_in.g * cur_in.g);
        dc = dct * cur_in.f;
        const float v = __int_as_float((__float_as_int(p_i) & m0) | (__float_as_int(p_f) & m1) | (__float_as_int(p_g) & m2) |
                                       (__float_as_int(p_o) & m3));
        gbuf[cur][q * H + k] = v;
        dgb[fr * 8 * H + q * H] = v;
        __syncthreads();
        const float4* gq = reinterpret_cast<const float4*>(gbuf[cur]);
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += 8) {
            float4 gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < NC) gv[u] = gq[LPU * (c0 + u) + q];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < NC) {
                    const int c = c0 + u;
                    a0 = __builtin_elementwise_fma(wt[2 * c], f32x2{gv[u].x, gv[u].y}, a0);
                    a1 = __builtin_elementwise_fma(wt[2 * c + 1], f32x2{gv[u].z, gv[u].w}, a1);
                }
        }
        dh = quad_sum((a0.x + a0.y) + (a1.x + a1.y));
        cur_in = nxt;
        fr += dt;
        // gbuf is double buffered: the next step writes gbuf[cur^1], whose readers all passed the barrier above
    }
}

// Backward recurrence, row layout (as gru_bwd_row_kernel): the unit layout above reads all 4H gate gradients per lane quad
// (32 ds_read_b128 per lane and step at H = 128: 8 waves x 32 x 8 cycles = 2048 LDS cycles, the whole measured step).  Here a
// ROW of 16 lanes owns 4 hidden units: a lane holds W_hh^T for those 4 columns over 1/16 of the gate rows (the same 128
// weight VGPRs), reads 4H/16 gate gradients per step (8 ds_read_b128) and the four partial sums are reduce-scattered over
// the row (rotated accumulator slots: 5 DPP adds, every lane ends with the total of ITS unit).  Lane r of a row plays
// unit 4 row + (r & 3), gate plane r >> 2 (i, f, g, o).
template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bwd_row_kernel(const float* __restrict__ dy, const float* __restrict__ gates,
                                                             const float* __restrict__ w_hh, const int* __restrict__ lengths, int T,
                                                             float* __restrict__ dg) {
    constexpr int NT = 4 * H;
    constexpr int VL = 4 * H / 16;  // gate rows per lane
    constexpr int NCH = VL / 4;     // ds_read_b128 per lane and step
    __shared__ __attribute__((aligned(16))) float gbuf[2][4 * H];
    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, row = tid >> 4, r = tid & 15;
    const int qp = r & 3, pl = r >> 2;
    const int k0 = row * 4, k = k0 + qp;
    const int len = lengths[b];

    // slot s works for unit (s + qp) & 3; wt[s][2c + pp] = W_hh^T rows (c*16 + r)*4 + 2pp (+1) of that unit's column
    f32x2 wt[4][VL / 2];
    {
        const float* wd = w_hh + (long)dir * 4 * H * H;
        auto pick = [](const float4& v, int u) { return u == 0 ? v.x : u == 1 ? v.y : u == 2 ? v.z : v.w; };
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int i = (c * 16 + r) * 4 + 2 * pp;
                const float4 lo = *reinterpret_cast<const float4*>(wd + (long)i * H + k0);
                const float4 hi = *reinterpret_cast<const float4*>(wd + (long)(i + 1) * H + k0);
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) wt[sl][2 * c + pp] = f32x2{pick(lo, (sl + qp) & 3), pick(hi, (sl + qp) & 3)};
            }
    }
    for (long i = (long)len * 4 * H + tid; i < (long)T * 4 * H; i += NT) {  // padded frames feed the time-batched GEMMs as zeros
        const long t = i / (4 * H), c = i % (4 * H);
        dg[(((long)b * T + t) * 2 + dir) * 4 * H + c] = 0.f;
    }
    if (len <= 0) return;

    const int t0 = dir ? 0 : len - 1;  // opposite to the forward walk
    const int dt = dir ? 1 : -1;
    const float* gtb = gates + (long)dir * 5 * H + k;        // + frame * 10H
    const float* dyb = dy + dir * H + k;                     // + frame * 2H
    float* dgb = dg + (long)dir * 4 * H + pl * H + k;        // + frame * 8H
    const int m0 = pl == 0 ? -1 : 0, m1 = pl == 1 ? -1 : 0, m2 = pl == 2 ? -1 : 0, m3 = pl == 3 ? -1 : 0;
    // nothing in `load` may consume a loaded value (gru.hip): has_prev is applied where cprev is used
    struct In { float i, f, g, o, c, cprev, dyv; bool has_prev; };
    auto load = [&](long fr, bool has_prev) {
        In v;
        const float* gp = gtb + fr * 10 * H;
        v.i = gp[0]; v.f = gp[H]; v.g = gp[2 * H]; v.o = gp[3 * H]; v.c = gp[4 * H];
        v.cprev = gtb[(fr + (has_prev ? dt : 0)) * 10 * H + 4 * H];
        v.has_prev = has_prev;
        v.dyv = dyb[fr * 2 * H];
        return v;
    };
    auto dpp_add = [](float acc, float v, auto ctrl) {
        return acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    const long fbase = (long)b * T + t0;   // frame of step u: fbase + u * dt
    float dh = 0.f, dc = 0.f;
    auto load_step = [&](int u) {
        const int uc = u < len ? u : len - 1;
        return load(fbase + (long)uc * dt, uc + 1 < len);
    };
    // One step: consumes `cur_in` (loaded two steps ago), starts the loads of step s + 2 into `fill`
    auto step = [&](int s, const In& cur_in, In& fill) {
        const int cur = s & 1;
        const long fr = fbase + (long)s * dt;
        fill = load_step(s + 2);
        const float dht = dh + cur_in.dyv;
        const float tc = as_tanh(cur_in.c);
        const float dct = dc + dht * cur_in.o * (1.f - tc * tc);
        const float p_o = dht * tc * cur_in.o * (1.f - cur_in.o);
        const float p_i = dct * cur_in.g * cur_in.i * (1.f - cur_in.i);
        const float p_f = dct * (cur_in.has_prev ? cur_in.cprev : 0.f) * cur_in.f * (1.f - cur_in.f);
        const float p_g = dct * cur_in.i * (1.f - cur_in.g * cur_in.g);
        dc = dct * cur_in.f;
        const float v = __int_as_float((__float_as_int(p_i) & m0) | (__float_as_int(p_f) & m1) | (__float_as_int(p_g) & m2) |
                                       (__float_as_int(p_o) & m3));
        gbuf[cur][pl * H + k] = v;
        dgb[fr * 8 * H] = v;
        __syncthreads();
        const float4* gq = reinterpret_cast<const float4*>(gbuf[cur]);
        float4 gv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) gv[c] = gq[c * 16 + r];
        __builtin_amdgcn_sched_barrier(0);
        f32x2 a[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const f32x2 lo = {gv[c].x, gv[c].y}, hi = {gv[c].z, gv[c].w};
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                a[sl] = __builtin_elementwise_fma(wt[sl][2 * c], lo, a[sl]);
                a[sl] = __builtin_elementwise_fma(wt[sl][2 * c + 1], hi, a[sl]);
            }
        }
        float acc = a[0].x + a[0].y;
        acc = dpp_add(acc, a[3].x + a[3].y, std::integral_constant<int, 0x39>{});   // quad_perm [1,2,3,0]
        acc = dpp_add(acc, a[2].x + a[2].y, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
        acc = dpp_add(acc, a[1].x + a[1].y, std::integral_constant<int, 0x93>{});   // quad_perm [3,0,1,2]
        acc = dpp_add(acc, acc, std::integral_constant<int, 0x124>{});              // row_ror 4
        acc = dpp_add(acc, acc, std::integral_constant<int, 0x128>{});              // row_ror 8
        dh = acc;
        // gbuf is double buffered: the next step writes gbuf[cur^1], whose readers all passed the barrier above
    };
    In a = load_step(0), bq = load_step(1), c;
    int s = 0;
    for (; s + 2 < len; s += 3) {
        step(s, a, c);
        step(s + 1, bq, a);
        step(s + 2, c, bq);
    }
    if (s < len) step(s, a, c);
    if (s + 1 < len) step(s + 1, bq, a);
}

// ---- any hidden size (nn.LSTM takes any, principal_components/models/rnn.py:58-68): plain kernels for the sizes the
// register-resident ones above are not built for (gru.hip has the same pair for the GRU).  One workgroup of 1024 threads per
// (utterance, direction), h and c in LDS, W_hh streamed from L2 every step.  Same gates / y / dg layouts and packed-sequence
// semantics; the reduction order over k differs from the kernels above in the last bits.  A correct fallback, not a tuned path.
constexpr int GEN_THREADS = 1024;

// Forward: four adjacent lanes share a hidden unit; lane q takes the 16-byte chunks q, q + 4, ... of the unit's four W_hh rows
// (the quad reads 64 consecutive bytes of a row), four chunks = 16 global loads in flight per pass.  H % 4 == 0.
template <bool TRAIN, bool TOK>
__global__ __launch_bounds__(GEN_THREADS) void lstm_fwd_generic_kernel(const float* __restrict__ gi, const int64_t* __restrict__ tokens,
                                                                      long tok_stride, const float* __restrict__ w_hh,
                                                                      const float* __restrict__ b_hh, const int* __restrict__ lengths,
                                                                      int T, int H, float* __restrict__ y, float* __restrict__ gates) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];   // h double buffer [2][H], c [H]
    float* cb = gsm + 2 * H;
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int q = tid & 3;
    const int len = lengths[b];
    for (long i = (long)len * H + tid; i < (long)T * H; i += GEN_THREADS)   // pad_packed_sequence: exact zeros
        y[((long)b * T + i / H) * 2 * H + dir * H + (i % H)] = 0.f;
    for (int j = tid; j < 3 * H; j += GEN_THREADS) gsm[j] = 0.f;
    __syncthreads();
    if (len <= 0) return;
    const float* wd = w_hh + (long)dir * 4 * H * H;
    const float* bd = b_hh + (long)dir * 4 * H;
    const int nch = H >> 2;   // 16-byte chunks per row
    for (int s = 0; s < len; ++s) {
        const int t = dir ? len - 1 - s : s;
        const long frame = (long)b * T + t;
        const float* hc = gsm + (s & 1) * H;
        float* hn_ = gsm + ((s & 1) ^ 1) * H;
        const float* gr = gi + (TOK ? tokens[(long)b * tok_stride + t] * 8L * H : frame * 8L * H) + (long)dir * 4 * H;
        for (int j = tid >> 2; j < H; j += GEN_THREADS / 4) {   // (the four lanes of a quad share j: the DPP sums are whole)
            const float4* h4 = reinterpret_cast<const float4*>(hc);
            float sg[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c0 = q; c0 < nch; c0 += 16) {
                float4 wv[4][4], hv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {      // branch-free: chunks beyond the row re-read its last one against a zero h
                    const int cc = c0 + 4 * u;
                    const int ci = cc < nch ? cc : nch - 1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) wv[g][u] = reinterpret_cast<const float4*>(wd + (long)(g * H + j) * H)[ci];
                    hv[u] = h4[ci];
                    if (cc >= nch) hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        sg[g] += wv[g][u].x * hv[u].x + wv[g][u].y * hv[u].y + wv[g][u].z * hv[u].z + wv[g][u].w * hv[u].w;
            }
            const float gi_ = as_sigmoid(gr[j] + quad_sum(sg[0]) + bd[j]);
            const float gf = as_sigmoid(gr[H + j] + quad_sum(sg[1]) + bd[H + j]);
            const float gg = as_tanh(gr[2 * H + j] + quad_sum(sg[2]) + bd[2 * H + j]);
            const float go = as_sigmoid(gr[3 * H + j] + quad_sum(sg[3]) + bd[3 * H + j]);
            const float cnew = gf * cb[j] + gi_ * gg;     // (the quad's lanes read c before lane 0 writes it: one wave, program order)
            const float hnew = go * as_tanh(cnew);
            if (q == 0) {
                cb[j] = cnew;
                hn_[j] = hnew;
                y[frame * 2 * H + dir * H + j] = hnew;
                if (TRAIN) {
                    float* gp = gates + frame * 10L * H + (long)dir * 5 * H + j;
                    gp[0] = gi_; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go; gp[4 * H] = cnew;
                }
            }
        }
        __syncthreads();
    }
}

// Backward: the four gate gradients of a step by one thread per hidden unit; then dh = W_hh^T p with the 4H gate rows dealt
// over four thread groups (lanes = consecutive hidden columns: a wave reads 256 consecutive bytes of a row, eight rows in
// flight), the four partial sums meeting in LDS in a fixed order.
__global__ __launch_bounds__(GEN_THREADS) void lstm_bwd_generic_kernel(const float* __restrict__ dy, const float* __restrict__ gates,
                                                                      const float* __restrict__ w_hh, const int* __restrict__ lengths,
                                                                      int T, int H, float* __restrict__ dg) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];   // p [4H], dh carried [H], dc carried [H], partial sums [4][H]
    float* gb = gsm;
    float* dhb = gsm + 4 * H;
    float* dcb = gsm + 5 * H;
    float* part = gsm + 6 * H;
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int len = lengths[b];
    for (long i = (long)len * 4 * H + tid; i < (long)T * 4 * H; i += GEN_THREADS) {   // padded frames feed the time-batched GEMMs as zeros
        const long t = i / (4 * H), c = i % (4 * H);
        dg[(((long)b * T + t) * 2 + dir) * 4 * H + c] = 0.f;
    }
    for (int j = tid; j < 2 * H; j += GEN_THREADS) dhb[j] = 0.f;   // dh and dc
    __syncthreads();
    if (len <= 0) return;
    const float* wd = w_hh + (long)dir * 4 * H * H;
    const int dt = dir ? 1 : -1;   // opposite to the forward walk; c_prev of frame t is the cell state of the frame visited NEXT
    const int kq = tid >> 8, kl = tid & 255;   // row group (0..3), column within a block of 256
    for (int s = 0; s < len; ++s) {
        const int t = dir ? s : len - 1 - s;
        const long frame = (long)b * T + t;
        const bool has_prev = s + 1 < len;
        for (int j = tid; j < H; j += GEN_THREADS) {
            const float* gp = gates + frame * 10L * H + (long)dir * 5 * H + j;
            const float gi_ = gp[0], gf = gp[H], gg = gp[2 * H], go = gp[3 * H], c = gp[4 * H];
            const float cprev = has_prev ? gates[(frame + dt) * 10L * H + (long)dir * 5 * H + 4 * H + j] : 0.f;
            const float dht = dhb[j] + dy[frame * 2 * H + dir * H + j];
            const float tc = as_tanh(c);
            const float dct = dcb[j] + dht * go * (1.f - tc * tc);
            const float p_o = dht * tc * go * (1.f - go);
            const float p_i = dct * gg * gi_ * (1.f - gi_);
            const float p_f = dct * cprev * gf * (1.f - gf);
            const float p_g = dct * gi_ * (1.f - gg * gg);
            dcb[j] = dct * gf;
            float* d = dg + (frame * 2 + dir) * 4L * H + j;
            d[0] = p_i; d[H] = p_f; d[2 * H] = p_g; d[3 * H] = p_o;
            gb[j] = p_i; gb[H + j] = p_f; gb[2 * H + j] = p_g; gb[3 * H + j] = p_o;
        }
        __syncthreads();
        for (int k0 = 0; k0 < H; k0 += 256) {
            const int k = k0 + kl;
            const int kc = k < H ? k : H - 1;
            float acc = 0.f;
            for (int i0 = kq; i0 < 4 * H; i0 += 32) {   // rows kq, kq + 4, ...: eight of them in flight
                float wv[8], gv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 4 * u;
                    const int ic = i < 4 * H ? i : 4 * H - 1;
                    wv[u] = wd[(long)ic * H + kc];
                    gv[u] = i < 4 * H ? gb[ic] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += gv[u] * wv[u];
            }
            if (k < H) part[kq * H + k] = acc;
        }
        __syncthreads();
        for (int k = tid; k < H; k += GEN_THREADS) dhb[k] = (part[k] + part[H + k]) + (part[2 * H + k] + part[3 * H + k]);
        __syncthreads();
    }
}

}  // namespace

extern "C" int as_lstm_bidir_fwd(const float* gi, const int64_t* tokens, int64_t tok_stride, const float* w_hh, const float* b_hh,
                                 const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* y, float* gates, void* stream) {
    AS_REQUIRE(gi && w_hh && b_hh && lengths && y, AS_ERR_BAD_ARG, "as_lstm_bidir_fwd: null pointer");
    AS_REQUIRE(B > 0 && T > 0, AS_ERR_BAD_ARG, "as_lstm_bidir_fwd: B=%d T=%d", B, T);
    AS_REQUIRE(!tokens || T <= 32768, AS_ERR_UNSUPPORTED, "as_lstm_bidir_fwd: T=%d > 32768 with a token table", T);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B, 2);
    const size_t shm = tokens ? (size_t)T * sizeof(int) : 0;
#define AS_LSTM_LAUNCH(HH, TR, TK) \
    hipLaunchKernelGGL((lstm_fwd_kernel<HH, TR, TK>), grid, dim3(LPU * HH), shm, st, gi, tokens, (long)tok_stride, w_hh, b_hh, lengths, T, y, gates)
#define AS_LSTM_FWD(HH)                                  \
    if (gates && tokens) AS_LSTM_LAUNCH(HH, true, true); \
    else if (gates) AS_LSTM_LAUNCH(HH, true, false);     \
    else if (tokens) AS_LSTM_LAUNCH(HH, false, true);    \
    else AS_LSTM_LAUNCH(HH, false, false);
    switch (H) {
        case 32: AS_LSTM_FWD(32) break;
        case 64: AS_LSTM_FWD(64) break;
        case 128: AS_LSTM_FWD(128) break;
        default: {   // any other hidden size: the plain kernels
            const size_t gshm = (size_t)3 * H * sizeof(float);
            AS_REQUIRE(H > 0 && H % 4 == 0 && gshm <= 64 * 1024, AS_ERR_UNSUPPORTED, "as_lstm_bidir_fwd: hidden size %d (a multiple of 4 up to 5460)", H);
#define AS_LSTM_GEN(TR, TK) \
    hipLaunchKernelGGL((lstm_fwd_generic_kernel<TR, TK>), grid, dim3(GEN_THREADS), gshm, st, gi, tokens, (long)tok_stride, w_hh, b_hh, lengths, T, H, y, gates)
            if (gates && tokens) AS_LSTM_GEN(true, true);
            else if (gates) AS_LSTM_GEN(true, false);
            else if (tokens) AS_LSTM_GEN(false, true);
            else AS_LSTM_GEN(false, false);
#undef AS_LSTM_GEN
        }
    }
#undef AS_LSTM_FWD
#undef AS_LSTM_LAUNCH
    AS_LAUNCH_CHECK("as_lstm_bidir_fwd");
    return 0;
}

extern "C" int as_lstm_bidir_bwd(const float* dy, const float* gates, const float* w_hh, const int32_t* lengths, int32_t B, int32_t T,
                                 int32_t H, float* dg, void* stream) {
    AS_REQUIRE(dy && gates && w_hh && lengths && dg, AS_ERR_BAD_ARG, "as_lstm_bidir_bwd: null pointer");
    AS_REQUIRE(B > 0 && T > 0, AS_ERR_BAD_ARG, "as_lstm_bidir_bwd: B=%d T=%d", B, T);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B, 2);
#ifdef AS_DIAG
    static const bool unit_layout = AS_DIAG_SET("AS_LSTM_BWD_UNIT");  // ablation: the older 4-lanes-per-unit layout
#define AS_LSTM_BWD(HH)                                                                                                     \
    if (unit_layout) hipLaunchKernelGGL((lstm_bwd_kernel<HH>), grid, dim3(LPU * HH), 0, st, dy, gates, w_hh, lengths, T, dg); \
    else hipLaunchKernelGGL((lstm_bwd_row_kernel<HH>), grid, dim3(4 * HH), 0, st, dy, gates, w_hh, lengths, T, dg)
#else
#define AS_LSTM_BWD(HH) hipLaunchKernelGGL((lstm_bwd_row_kernel<HH>), grid, dim3(4 * HH), 0, st, dy, gates, w_hh, lengths, T, dg)
#endif
    switch (H) {
        case 32: AS_LSTM_BWD(32); break;
        case 64: AS_LSTM_BWD(64); break;
        case 128: AS_LSTM_BWD(128); break;
        default: {   // any other hidden size: the plain kernel
            const size_t gshm = (size_t)10 * H * sizeof(float);
            AS_REQUIRE(H > 0 && H % 4 == 0 && gshm <= 64 * 1024, AS_ERR_UNSUPPORTED, "as_lstm_bidir_bwd: hidden size %d (a multiple of 4 up to 1636)", H);
            hipLaunchKernelGGL(lstm_bwd_generic_kernel, grid, dim3(GEN_THREADS), gshm, st, dy, gates, w_hh, lengths, T, H, dg);
        }
    }
#undef AS_LSTM_BWD
    AS_LAUNCH_CHECK("as_lstm_bidir_bwd");
    return 0;
}
