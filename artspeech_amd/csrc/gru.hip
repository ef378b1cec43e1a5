// Bidirectional GRU recurrence (nn.GRU semantics, encoder_decoder/models.py:111,137) as persistent
// kernels: one workgroup per (utterance, direction) walks its own sequence, so packed-sequence
// semantics (per-utterance length, reverse direction starting at len-1, zero padded outputs) cost nothing.
//
// The step is a dependent chain, so the design minimises per-step latency rather than bytes:
//   * W_hh (3H x H fp32, 192 KB at H=128: more than the 160 KB LDS) is held in REGISTERS for the whole
//     sequence, spread over the workgroup: 4 lanes per hidden unit, each lane owns a quarter of the
//     reduction index of that unit's three gate rows (96 VGPRs at H=128);
//   * h_{t-1} lives in LDS (double buffered, ONE barrier per step); a lane's quarter is interleaved in
//     16-byte pieces (k = 16c + 4q + i) so the four lanes of a quad read four consecutive 16-B slots:
//     every ds_read_b128 is a conflict-free broadcast;
//   * the three gate dot products are finished with two quad shuffles (DPP), not LDS;
//   * the input projections (time-parallel, W_ih x + b_ih) come precomputed; for layer 0 they are a
//     [V] row table (embedding folded into W_ih), gathered by token id, and are prefetched one step ahead.
// The backward kernel mirrors this with W_hh^T in registers (lane owns a quarter of the 3H gate rows of
// one hidden unit's column) and emits the pre-activation gradients; weight gradients are time-batched
// GEMMs over them (artspeech.hip).
#include "as_common.h"

namespace {

__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}

template <int H, bool TRAIN>
__global__ __launch_bounds__(4 * H) void gru_fwd_kernel(const float* __restrict__ gi, const int64_t* __restrict__ tokens,
                                                        long tok_stride, const float* __restrict__ w_hh,
                                                        const float* __restrict__ b_hh, const int* __restrict__ lengths,
                                                        int T, float* __restrict__ y, float* __restrict__ gates) {
    constexpr int NC = H / 16;  // 16-float chunks of the reduction index; a lane owns 4 floats of each
    __shared__ __attribute__((aligned(16))) float hbuf[2][H];
    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, j = tid >> 2, q = tid & 3;
    const int len = lengths[b];

    float w[3][NC * 4];
    {
        const float* wd = w_hh + (long)dir * 3 * H * H;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(wd + (long)(g * H + j) * H + 16 * c + 4 * q);
                w[g][4 * c] = v.x; w[g][4 * c + 1] = v.y; w[g][4 * c + 2] = v.z; w[g][4 * c + 3] = v.w;
            }
    }
    const float bh_r = b_hh[dir * 3 * H + j], bh_z = b_hh[dir * 3 * H + H + j], bh_n = b_hh[dir * 3 * H + 2 * H + j];

    // pad_packed_sequence: outputs of padded frames are exact zeros
    for (int t = len + (tid / H); t < T; t += 4) y[((long)b * T + t) * 2 * H + dir * H + (tid % H)] = 0.f;
    if (tid < H) hbuf[0][tid] = 0.f;
    __syncthreads();

    auto gi_row = [&](int t) -> const float* {
        const long row = tokens ? (long)tokens[(long)b * tok_stride + t] : (long)b * T + t;
        return gi + (row * 2 + dir) * 3 * H;
    };
    float h = 0.f;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (len > 0) {
        const float* p = gi_row(dir ? len - 1 : 0);
        gr = p[j]; gz = p[H + j]; gn = p[2 * H + j];
    }
    for (int s = 0; s < len; ++s) {
        const int t = dir ? len - 1 - s : s;
        const int cur = s & 1;
        // prefetch next step's input projection (independent of the recurrence)
        float ngr = 0.f, ngz = 0.f, ngn = 0.f;
        if (s + 1 < len) {
            const float* p = gi_row(dir ? t - 1 : t + 1);
            ngr = p[j]; ngz = p[H + j]; ngn = p[2 * H + j];
        }
        const float4* hp = reinterpret_cast<const float4*>(hbuf[cur]);
        float ar = 0.f, az = 0.f, an = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 hv = hp[4 * c + q];
            ar = fmaf(w[0][4 * c], hv.x, ar); ar = fmaf(w[0][4 * c + 1], hv.y, ar);
            ar = fmaf(w[0][4 * c + 2], hv.z, ar); ar = fmaf(w[0][4 * c + 3], hv.w, ar);
            az = fmaf(w[1][4 * c], hv.x, az); az = fmaf(w[1][4 * c + 1], hv.y, az);
            az = fmaf(w[1][4 * c + 2], hv.z, az); az = fmaf(w[1][4 * c + 3], hv.w, az);
            an = fmaf(w[2][4 * c], hv.x, an); an = fmaf(w[2][4 * c + 1], hv.y, an);
            an = fmaf(w[2][4 * c + 2], hv.z, an); an = fmaf(w[2][4 * c + 3], hv.w, an);
        }
        ar = quad_sum(ar); az = quad_sum(az); an = quad_sum(an);
        const float r = as_sigmoid(gr + (ar + bh_r));
        const float z = as_sigmoid(gz + (az + bh_z));
        const float hn = an + bh_n;
        const float n = as_tanh(gn + r * hn);
        const float hnew = (1.f - z) * n + z * h;
        h = hnew;
        const long fr = (long)b * T + t;
        if (q == 0) {
            hbuf[cur ^ 1][j] = hnew;
            y[fr * 2 * H + dir * H + j] = hnew;
        }
        if (TRAIN) {
            const float gv = q == 0 ? r : (q == 1 ? z : (q == 2 ? n : hn));
            gates[((fr * 2 + dir) * 4 + q) * H + j] = gv;
        }
        gr = ngr; gz = ngz; gn = ngn;
        __syncthreads();
    }
}

template <int H>
__global__ __launch_bounds__(4 * H) void gru_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        const float* __restrict__ gates, const float* __restrict__ w_hh,
                                                        const int* __restrict__ lengths, int T, float* __restrict__ dgi,
                                                        float* __restrict__ dgh) {
    constexpr int NC = 3 * H / 16;
    __shared__ __attribute__((aligned(16))) float gbuf[2][3 * H];
    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, k = tid >> 2, q = tid & 3;
    const int len = lengths[b];

    // W_hh^T: this lane owns rows i = 16c + 4q + ii of column k
    float wt[NC * 4];
    {
        const float* wd = w_hh + (long)dir * 3 * H * H;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) wt[4 * c + ii] = wd[(long)(16 * c + 4 * q + ii) * H + k];
    }
    // zero the gradients of padded frames (rows feed time-batched GEMMs)
    for (long i = (long)len * 3 * H + tid; i < (long)T * 3 * H; i += 4 * H) {
        const long t = i / (3 * H), c = i % (3 * H);
        const long o = (((long)b * T + t) * 2 + dir) * 3 * H + c;
        dgi[o] = 0.f;
        dgh[o] = 0.f;
    }
    float dh = 0.f;
    for (int s = 0; s < len; ++s) {
        // walk opposite to the forward: forward dir t = len-1..0, reverse dir t = 0..len-1
        const int t = dir ? s : len - 1 - s;
        const int cur = s & 1;
        const long fr = (long)b * T + t;
        const float* gp = gates + (fr * 2 + dir) * 4 * H;
        const float r = gp[k], z = gp[H + k], n = gp[2 * H + k], hn = gp[3 * H + k];
        const int tp = dir ? t + 1 : t - 1;  // frame whose output was h_{prev} of this step
        const float hprev = (tp >= 0 && tp < len) ? y[((long)b * T + tp) * 2 * H + dir * H + k] : 0.f;
        const float dht = dh + dy[fr * 2 * H + dir * H + k];
        const float dn = dht * (1.f - z);
        const float dz = dht * (hprev - n);
        const float dnt = dn * (1.f - n * n);
        const float g_r = dnt * hn * r * (1.f - r);
        const float g_z = dz * z * (1.f - z);
        const float g_hn = dnt * r;
        const long o = (fr * 2 + dir) * 3 * H;
        if (q == 0) { gbuf[cur][k] = g_r; dgi[o + k] = g_r; dgh[o + k] = g_r; }
        else if (q == 1) { gbuf[cur][H + k] = g_z; dgi[o + H + k] = g_z; dgh[o + H + k] = g_z; }
        else if (q == 2) { gbuf[cur][2 * H + k] = g_hn; dgh[o + 2 * H + k] = g_hn; }
        else { dgi[o + 2 * H + k] = dnt; }
        __syncthreads();
        const float4* gq = reinterpret_cast<const float4*>(gbuf[cur]);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 gv = gq[4 * c + q];
            acc = fmaf(wt[4 * c], gv.x, acc); acc = fmaf(wt[4 * c + 1], gv.y, acc);
            acc = fmaf(wt[4 * c + 2], gv.z, acc); acc = fmaf(wt[4 * c + 3], gv.w, acc);
        }
        acc = quad_sum(acc);
        dh = dht * z + acc;
        // gbuf is double buffered: the next step writes gbuf[cur^1]; all reads of it (two steps ago)
        // precede the barrier above, so one barrier per step suffices.
    }
}

}  // namespace

extern "C" int as_gru_bidir_fwd(const float* gi, const int64_t* tokens, int64_t tok_stride, const float* w_hh,
                                const float* b_hh, const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* y,
                                float* gates, void* stream) {
    AS_REQUIRE(gi && w_hh && b_hh && lengths && y, AS_ERR_BAD_ARG, "as_gru_bidir_fwd: null pointer");
    AS_REQUIRE(B > 0 && T > 0, AS_ERR_BAD_ARG, "as_gru_bidir_fwd: B=%d T=%d", B, T);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B, 2);
#define AS_GRU_FWD(HH)                                                                                              \
    if (gates)                                                                                                      \
        hipLaunchKernelGGL((gru_fwd_kernel<HH, true>), grid, dim3(4 * HH), 0, st, gi, tokens, (long)tok_stride, w_hh, \
                           b_hh, lengths, T, y, gates);                                                             \
    else                                                                                                            \
        hipLaunchKernelGGL((gru_fwd_kernel<HH, false>), grid, dim3(4 * HH), 0, st, gi, tokens, (long)tok_stride, w_hh, \
                           b_hh, lengths, T, y, gates);
    switch (H) {
        case 32: AS_GRU_FWD(32) break;
        case 64: AS_GRU_FWD(64) break;
        case 128: AS_GRU_FWD(128) break;
        default:
            as_set_error("as_gru_bidir_fwd: hidden size %d not in {32, 64, 128}", H);
            return AS_ERR_UNSUPPORTED;
    }
#undef AS_GRU_FWD
    AS_LAUNCH_CHECK("as_gru_bidir_fwd");
    return 0;
}

extern "C" int as_gru_bidir_bwd(const float* dy, const float* y, const float* gates, const float* w_hh,
                                const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* dgi, float* dgh,
                                void* stream) {
    AS_REQUIRE(dy && y && gates && w_hh && lengths && dgi && dgh, AS_ERR_BAD_ARG, "as_gru_bidir_bwd: null pointer");
    AS_REQUIRE(B > 0 && T > 0, AS_ERR_BAD_ARG, "as_gru_bidir_bwd: B=%d T=%d", B, T);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B, 2);
    switch (H) {
        case 32: hipLaunchKernelGGL((gru_bwd_kernel<32>), grid, dim3(128), 0, st, dy, y, gates, w_hh, lengths, T, dgi, dgh); break;
        case 64: hipLaunchKernelGGL((gru_bwd_kernel<64>), grid, dim3(256), 0, st, dy, y, gates, w_hh, lengths, T, dgi, dgh); break;
        case 128: hipLaunchKernelGGL((gru_bwd_kernel<128>), grid, dim3(512), 0, st, dy, y, gates, w_hh, lengths, T, dgi, dgh); break;
        default:
            as_set_error("as_gru_bidir_bwd: hidden size %d not in {32, 64, 128}", H);
            return AS_ERR_UNSUPPORTED;
    }
    AS_LAUNCH_CHECK("as_gru_bidir_bwd");
    return 0;
}
