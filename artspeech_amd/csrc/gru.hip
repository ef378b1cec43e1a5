// Bidirectional GRU recurrence (nn.GRU semantics, encoder_decoder/models.py:111,137) as persistent
// kernels: one workgroup per (utterance, direction) walks its own sequence, so packed-sequence
// semantics (per-utterance length, reverse direction starting at len-1, zero padded outputs) cost nothing.
//
// The step is a dependent chain, so the design minimises per-step latency rather than bytes:
//   * W_hh (3H x H fp32, 192 KB at H=128: more than the 160 KB LDS) is held in REGISTERS for the whole
//     sequence, spread over the workgroup: LPU lanes per hidden unit, each lane owns 1/LPU of the
//     reduction index of that unit's three gate rows (LPU = 4: 96 weight VGPRs per lane at H = 128, two
//     waves per SIMD).  In-kernel cycle stamps (diagnostic build) put a step at ~1300 cycles: LDS read 130,
//     FMAs + reduction ~430 (VALU issue shared by the SIMD's two waves), gate math ~200, LDS write + stores
//     ~130 and ~430 waiting at the barrier for the partner wave -- VALU-issue bound, not memory bound;
//   * h_{t-1} lives in LDS (double buffered, ONE barrier per step); a lane's share is interleaved in
//     16-byte pieces (k = 4*LPU*c + 4q + i) so the lanes of a unit read consecutive 16-B slots:
//     every ds_read_b128 is a conflict-free broadcast; reads are issued ahead of the FMAs in batches of 8;
//   * the three gate dot products are finished with DPP quad permutes (no LDS round trip);
//   * everything that does not depend on the recurrence (input projections, saved gates, upstream
//     gradients) is loaded TWO STEPS AHEAD into three operand sets whose roles rotate by NAME through a
//     loop unrolled by three: nothing may consume a loaded value in the step that issued the load -- not
//     even a register-to-register rotation, which made every load return within one step (0.55 us, about
//     one HBM round trip): +30-45 ns per step.  The loop body is branch-free (redundant quad lanes store
//     the same word) so that hipcc can count its loads and never waits for the step's own stores;
//   * the input projections (time-parallel, W_ih x + b_ih) come precomputed; for layer 0 they are a
//     [V] row table (embedding folded into W_ih) gathered by token id through an LDS copy of the ids.
// The backward kernel mirrors this with W_hh^T in registers (lane owns a quarter of the 3H gate rows of
// one hidden unit's column) and emits the pre-activation gradients; weight gradients are time-batched
// GEMMs over them (artspeech.hip).  (Tried and dropped: s_setprio(3) for these waves while weight-gradient GEMMs
// share the CU from the side stream -- no measurable change, the interference is not VALU issue arbitration.)
#include <cstdlib>

#include <type_traits>

#include <hip/hip_ext.h>

#include "as_common.h"
#include "gemm_internal.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// sum over the LPU (2 or 4) adjacent lanes that share a hidden unit: quad_perm [1,0,3,2] (then [2,3,0,1])
template <int LPU>
__device__ __forceinline__ float unit_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    if (LPU == 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    return v;
}

#ifdef AS_GRU_NO_GATE_SHARE
constexpr bool GATE_SHARE = false;
#else
constexpr bool GATE_SHARE = true;
#endif
template <int H, int LPU, bool TRAIN, bool TOK, int AHEAD>
__global__ __launch_bounds__(LPU * H) void gru_fwd_kernel(const float* __restrict__ gi, const int64_t* __restrict__ tokens,
                                                          long tok_stride, const float* __restrict__ w_hh,
                                                          const float* __restrict__ b_hh, const int* __restrict__ lengths,
                                                          int T, float* __restrict__ y, float* __restrict__ gates, int nd, int V) {
    constexpr int CW = 4 * LPU;   // floats of the reduction index covered by one ds_read_b128 of every lane of a unit
    constexpr int NC = H / CW;    // such chunks; a lane owns 4 floats of each
    constexpr int NT = LPU * H;   // threads
    // THREE buffers for h, indexed by the step's position in the loop that is unrolled by three: the LDS addresses of the
    // read (buffer d) and of the write (buffer d + 1) are then constants of each unrolled copy
    __shared__ __attribute__((aligned(16))) float hbuf[3][H];
    extern __shared__ int tok_s[];  // TOK: BYTE offset of every frame's row in the token table (token id x row stride)
    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, j = tid / LPU, q = tid % LPU;
    const int len = lengths[b];

    f32x2 w[3][NC * 2];  // packed pairs: v_pk_fma_f32 does two FMAs per lane per issue slot
    {
        const float* wd = w_hh + (long)dir * 3 * H * H;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(wd + (long)(g * H + j) * H + CW * c + 4 * q);
                w[g][2 * c] = f32x2{v.x, v.y};
                w[g][2 * c + 1] = f32x2{v.z, v.w};
            }
    }
    const float bq_r = b_hh[dir * 3 * H + j] * (1.0f / LPU), bq_z = b_hh[dir * 3 * H + H + j] * (1.0f / LPU),
                bq_n = b_hh[dir * 3 * H + 2 * H + j] * (1.0f / LPU);

    // pad_packed_sequence: outputs of padded frames are exact zeros
    for (long i = (long)len * H + tid; i < (long)T * H; i += NT)
        y[((long)b * T + i / H) * nd * H + dir * H + (i % H)] = 0.f;
    if (tid < H) hbuf[0][tid] = 0.f;
    if (TOK)
        for (int t = tid; t < len; t += NT) {   // V > 0: ids clamped into the table (counted for the host by as_artspeech_fwd)
            const int64_t v = tokens[(long)b * tok_stride + t];
            tok_s[t] = (int)(V > 0 ? min(max(v, (int64_t)0), (int64_t)V - 1) : v) * (nd * 3 * H * 4);   // bytes
        }
    __syncthreads();
    if (len <= 0) return;

    // Per-step addresses advance by constants: keep running (wave-uniform) element offsets instead of
    // re-deriving them from t (the address arithmetic otherwise rivals the FMAs in issue slots).
    const int t0 = dir ? len - 1 : 0;
    const int dt = dir ? -1 : 1;
    // Addresses as UNIFORM base pointer (SGPR pair) + 32-bit BYTE offset (one VGPR): the loads / stores then take the
    // scalar-base form and a step's address arithmetic is one scalar multiply and one vector add per access, instead of
    // 64-bit vector shifts and adds.  (Every array here is far below 4 GB: checked by the launcher.)
    const unsigned ys = (unsigned)nd * H * 4u, gs = (unsigned)nd * 4u * H * 4u, is = (unsigned)nd * 3u * H * 4u;   // bytes per frame
    const unsigned yo = (unsigned)(dir * H + j) * 4u;                       // + frame * ys
    const unsigned go = (unsigned)(dir * 4 * H + j + (LPU == 4 ? q * H : q * H)) * 4u;   // + frame * gs: this lane's plane
    const unsigned io = (unsigned)(dir * 3 * H + j) * 4u;                   // + row offset (bytes)
    const char* gi_c = reinterpret_cast<const char*>(gi);
    char* y_c = reinterpret_cast<char*>(y);
    char* g_c = reinterpret_cast<char*>(gates);
    // the LPU lanes of a unit hold identical gate values: lane q stores planes q, q + LPU, ... (branch-free selects)
    const int m0 = q == 0 ? -1 : 0, m1 = q == 1 ? -1 : 0, m2 = q == 2 ? -1 : 0, m3 = q == 3 ? -1 : 0;

    float h = 0.f;
    struct Gi { float r, z, n; };
    // input projection of step u (clamped to the last step: the look-ahead stays inside the sequence)
    auto load_step = [&](int u) {
        const int uc = u < len ? u : len - 1;
        const int tu = t0 + uc * dt;
        const unsigned ro = (TOK ? (unsigned)tok_s[tu] : (unsigned)(b * T + tu) * is) + io;
        const float* p = reinterpret_cast<const float*>(gi_c + ro);
        return Gi{p[0], p[H], p[2 * H]};
    };
    // One recurrent step: consumes `ci` (loaded AHEAD steps ago) and starts the loads of step s + AHEAD into `fill`.
    auto step = [&](auto curc, int s, const Gi& ci, Gi& fill) {
        const int cur = curc, nxt = (cur + 1) % 3;             // buffer read / written by this step
        const unsigned fr = (unsigned)(b * T + t0 + s * dt);   // frame of this step (wave-uniform)
        fill = load_step(s + AHEAD);
        const float4* hp = reinterpret_cast<const float4*>(hbuf[cur]);
        // the recurrent biases ride in as the accumulators' start value (1 / LPU of it in each lane of the unit: exact, LPU
        // is a power of two), which costs nothing -- the first FMA takes a register pair instead of the literal 0
        f32x2 ar = {bq_r, 0.f}, az = {bq_z, 0.f}, an = {bq_n, 0.f};
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += 8) {  // 8 LDS reads in flight, then their FMAs (hipcc otherwise pairs them 2 by 2)
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
                    ar = __builtin_elementwise_fma(w[0][2 * c], lo, ar);
                    az = __builtin_elementwise_fma(w[1][2 * c], lo, az);
                    an = __builtin_elementwise_fma(w[2][2 * c], lo, an);
                    ar = __builtin_elementwise_fma(w[0][2 * c + 1], hi, ar);
                    az = __builtin_elementwise_fma(w[1][2 * c + 1], hi, az);
                    an = __builtin_elementwise_fma(w[2][2 * c + 1], hi, an);
                }
        }
        const float sr = unit_sum<LPU>(ar.x + ar.y), sz = unit_sum<LPU>(az.x + az.y), sn = unit_sum<LPU>(an.x + an.y);
        float r, z;
        if constexpr (LPU == 4 && GATE_SHARE) {
            // the four lanes of a unit hold the same sums: lane 0 takes r's sigmoid, lanes 1..3 z's -- ONE exp + rcp sequence per
            // lane instead of two (quarter-rate instructions: 32 of the step's ~410 issue cycles per wave) -- and two quad
            // broadcasts hand both to every lane.  Same operations on the same values: bit-identical.
            const float pre = __int_as_float((__float_as_int(ci.r + sr) & m0) | (__float_as_int(ci.z + sz) & ~m0));
            const float sg = as_sigmoid(pre);
            r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0x00, 0xF, 0xF, true));   // quad_perm [0,0,0,0]
            z = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0x55, 0xF, 0xF, true));   // quad_perm [1,1,1,1]
        } else {
            r = as_sigmoid(ci.r + sr);
            z = as_sigmoid(ci.z + sz);
        }
        const float hn = sn;
        const float n = as_tanh(ci.n + r * hn);
        const float hnew = (1.f - z) * n + z * h;
        h = hnew;
        // the lanes of a unit hold identical values: all of them store (same word) -> no divergence
        hbuf[nxt][j] = hnew;
        *reinterpret_cast<float*>(y_c + (fr * ys + yo)) = hnew;
        if (TRAIN) {
            float* gp = reinterpret_cast<float*>(g_c + (fr * gs + go));
            if (LPU == 4) {
                // lane q stores plane q: three bitfield inserts ((a & m) | (b & ~m) is v_bfi_b32) on loop-invariant lane masks
                const int m01 = m0 | m1;
                const int t1 = (__float_as_int(r) & m0) | (__float_as_int(z) & ~m0);
                const int t2 = (__float_as_int(n) & m2) | (__float_as_int(hn) & ~m2);
                gp[0] = __int_as_float((t1 & m01) | (t2 & ~m01));
            } else {  // two lanes per unit: lane 0 stores r and n, lane 1 stores z and hn
                const int ga = (__float_as_int(r) & m0) | (__float_as_int(z) & m1);
                const int gc = (__float_as_int(n) & m0) | (__float_as_int(hn) & m1);
                gp[0] = __int_as_float(ga);
                gp[2 * H] = __int_as_float(gc);
            }
        }
        __syncthreads();
    };
    constexpr std::integral_constant<int, 0> c0{};
    constexpr std::integral_constant<int, 1> c1{};
    constexpr std::integral_constant<int, 2> c2{};
    if constexpr (AHEAD == 1) {
        Gi a = load_step(0), bn;
        for (int s = 0; s < len; ++s) {
            step(s % 3, s, a, bn);
            a = bn;
        }
    } else {
        // Two steps of look-ahead: three operand sets whose roles rotate by NAME through a loop unrolled by three.  A
        // register-to-register rotation is a CONSUMER of the newest loads: the wave would wait for them at the end of the
        // step that issued them, and the memory round trip (not the arithmetic) would set the step time.
        Gi a = load_step(0), bq = load_step(1), c;
        int s = 0;
        for (; s + 2 < len; s += 3) {
            step(c0, s, a, c);
            step(c1, s + 1, bq, a);
            step(c2, s + 2, c, bq);
        }
        if (s < len) step(c0, s, a, c);
        if (s + 1 < len) step(c1, s + 1, bq, a);
    }
}

template <int H, int LPU>
__global__ __launch_bounds__(LPU * H) void gru_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ gates, const float* __restrict__ w_hh,
                                                          const int* __restrict__ lengths, int T, float* __restrict__ dgi,
                                                          float* __restrict__ dgh) {
    constexpr int CW = 4 * LPU;
    constexpr int NC = 3 * H / CW;
    constexpr int NT = LPU * H;
    __shared__ __attribute__((aligned(16))) float gbuf[2][3 * H];
    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, k = tid / LPU, q = tid % LPU;
    const int len = lengths[b];

    // W_hh^T: this lane owns rows i = CW*c + 4q + ii of column k (packed pairs for v_pk_fma_f32)
    f32x2 wt[NC * 2];
    {
        const float* wd = w_hh + (long)dir * 3 * H * H;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            wt[2 * c] = f32x2{wd[(long)(CW * c + 4 * q) * H + k], wd[(long)(CW * c + 4 * q + 1) * H + k]};
            wt[2 * c + 1] = f32x2{wd[(long)(CW * c + 4 * q + 2) * H + k], wd[(long)(CW * c + 4 * q + 3) * H + k]};
        }
    }
    // zero the gradients of padded frames (rows feed time-batched GEMMs)
    for (long i = (long)len * 3 * H + tid; i < (long)T * 3 * H; i += NT) {
        const long t = i / (3 * H), c = i % (3 * H);
        const long o = (((long)b * T + t) * 2 + dir) * 3 * H + c;
        dgi[o] = 0.f;
        dgh[o] = 0.f;
    }
    if (len <= 0) return;

    // walk opposite to the forward: forward dir t = len-1..0, reverse dir t = 0..len-1; running offsets
    const int t0 = dir ? 0 : len - 1;
    const int dt = dir ? 1 : -1;
    const float* gtb = gates + (long)dir * 4 * H + k;   // + frame * 8H, planes at +0, +H, +2H, +3H
    const float* yb = y + dir * H + k;                   // + frame * 2H
    const float* dyb = dy + dir * H + k;                 // + frame * 2H
    float* dgib = dgi + (long)dir * 3 * H + k;           // + frame * 6H + plane * H
    float* dghb = dgh + (long)dir * 3 * H + k;
    // Plane(s) stored by this lane.  LPU = 4: lane q stores plane min(q, 2) of both arrays (lanes 2, 3 the same
    // words).  LPU = 2: lane 0 stores planes r (both arrays) and n of dgi; lane 1 planes z (both) and n of dgh.
    const int sel = LPU == 4 ? (q < 2 ? q : 2) : q;
    const int m0 = q == 0 ? -1 : 0, m1 = q == 1 ? -1 : 0, m2 = q >= 2 ? -1 : 0;
    struct In { float r, z, n, hn, hprev, dyv; };
    // h_{prev} of frame t is the output of the frame this backward walk visits NEXT (t + dt): forward dir t-1,
    // reverse dir t+1; zero beyond the sequence ends.
    auto load = [&](long fr, bool has_prev) {
        In v;
        const float* gp = gtb + fr * 8 * H;
        v.r = gp[0]; v.z = gp[H]; v.n = gp[2 * H]; v.hn = gp[3 * H];
        const float hp = yb[(fr + (has_prev ? dt : 0)) * 2 * H];
        v.hprev = has_prev ? hp : 0.f;
        v.dyv = dyb[fr * 2 * H];
        return v;
    };
    long fr = (long)b * T + t0;
    float dh = 0.f;
    In cur_in = load(fr, len > 1);
    for (int s = 0; s < len; ++s) {
        const int cur = s & 1;
        // next step's operands: in flight during this step (last step re-reads its own frame)
        const int adv = s + 1 < len ? dt : 0;
        const In nxt = load(fr + adv, s + 2 < len);
        const float r = cur_in.r, z = cur_in.z, n = cur_in.n, hn = cur_in.hn;
        const float dht = dh + cur_in.dyv;
        const float dn = dht * (1.f - z);
        const float dz = dht * (cur_in.hprev - n);
        const float dnt = dn * (1.f - n * n);
        const float g_r = dnt * hn * r * (1.f - r);
        const float g_z = dz * z * (1.f - z);
        const float g_hn = dnt * r;
        // planes r, z, n of d/d(W_ih x + b_ih) and d/d(W_hh h + b_hh): they differ in the n plane only
        if (LPU == 4) {
            const int rz = (__float_as_int(g_r) & m0) | (__float_as_int(g_z) & m1);
            const float vi = __int_as_float(rz | (__float_as_int(dnt) & m2));
            const float vh = __int_as_float(rz | (__float_as_int(g_hn) & m2));
            gbuf[cur][sel * H + k] = vh;
            dgib[fr * 6 * H + sel * H] = vi;
            dghb[fr * 6 * H + sel * H] = vh;
        } else {
            const float rz = __int_as_float((__float_as_int(g_r) & m0) | (__float_as_int(g_z) & m1));
            const float nn = __int_as_float((__float_as_int(dnt) & m0) | (__float_as_int(g_hn) & m1));
            gbuf[cur][sel * H + k] = rz;
            gbuf[cur][2 * H + k] = g_hn;  // both lanes write the same word
            dgib[fr * 6 * H + sel * H] = rz;
            dghb[fr * 6 * H + sel * H] = rz;
            (q == 0 ? dgib : dghb)[fr * 6 * H + 2 * H] = nn;
        }
        __syncthreads();
        const float4* gq = reinterpret_cast<const float4*>(gbuf[cur]);
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += 8) {  // 8 LDS reads in flight, then their FMAs
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
        const float acc = unit_sum<LPU>((a0.x + a0.y) + (a1.x + a1.y));
        dh = dht * z + acc;
        cur_in = nxt;
        fr += dt;
        // gbuf is double buffered: the next step writes gbuf[cur^1]; all reads of it (two steps ago)
        // precede the barrier above, so one barrier per step suffices.
    }
}

// Backward recurrence, second layout: the LDS read of the step's 3H gate gradients is what bounds the layout above (every
// lane reads 3H/4 floats: 24 ds_read_b128 per lane, 8 waves x 24 x 8 cycles = 1536 LDS cycles per step at H = 128, about
// the whole measured step).  Here a ROW of 16 lanes owns 4 hidden units: each lane holds W_hh^T for those 4 columns over
// 1/16 of the gate rows (the same 96 weight VGPRs), reads only 3H/16 gate gradients per step (6 ds_read_b128 at H = 128)
// and the four partial sums are all-reduced across the row with 4 DPP steps each (quad xor 1, xor 2, half-row mirror,
// row mirror).  Lane r of a row then plays the old role for unit 4*row + (r & 3), plane r >> 2.
// TOK (layer 0 under a token table): the input-side gate gradients are only ever summed per token (the embedding and the
// input projection see them through the table, rowops.hip emb_grads_kernel), so instead of writing dgi [B][T][2][3H] for a
// later segmented-sum pass, every workgroup keeps the sums of ITS utterance and direction in an LDS table [V][3H] (one
// read-modify-write per lane and step, off the recurrence's dependency chain; each word has one owner lane, so the order
// of additions is the frame order: reproducible) and stores it once at the end into part [B][V][2 * 3H]; dgi is not
// touched.  A fixed-order reduction over the B tables follows (token_segsum_reduce_kernel).
template <int H, bool TOK, int AHEAD>
__global__ __launch_bounds__(4 * H) void gru_bwd_row_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ gates, const float* __restrict__ w_hh,
                                                            const int* __restrict__ lengths, int T, float* __restrict__ dgi,
                                                            float* __restrict__ dgh, unsigned long long* __restrict__ dbg,
                                                            const int64_t* __restrict__ tokens, long tok_stride, int V,
                                                            float* __restrict__ part) {
    constexpr int NT = 4 * H;
    extern __shared__ float tab[];            // TOK: [V][3H] + a dummy word per lane + T token offsets
    constexpr int VL = 3 * H / 16;            // gate rows per lane
    // diagnostic (as_gru_debug_stamps): shader-clock and 100 MHz wall-clock stamps around the recurrence of this workgroup;
    // dbg is null in every product launch and no stamp executes then
    unsigned long long t_start = 0, r_start = 0;
    if (dbg != nullptr && threadIdx.x == 0) {
        t_start = __builtin_amdgcn_s_memtime();
        r_start = __builtin_amdgcn_s_memrealtime();
    }
    constexpr int VW = VL % 4 == 0 ? 4 : 2;   // floats per LDS read
    constexpr int NCH = VL / VW;              // LDS reads per lane per step
    __shared__ __attribute__((aligned(16))) float gbuf[3][3 * H];   // three buffers: index = position in the loop unrolled by three
    const int b = blockIdx.x, dir = blockIdx.y;
    const int tid = threadIdx.x, row = tid >> 4, r = tid & 15;
    const int k0 = row * 4, k = k0 + (r & 3), pl = r >> 2;
    const int len = lengths[b];

    // wt[kk][p]: W_hh^T rows i = (c*16 + r)*VW + 2*pp (+1), p = c*VW/2 + pp, of column k0 + kk
    f32x2 wt[4][VL / 2];
    {
        const float* wd = w_hh + (long)dir * 3 * H * H;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int pp = 0; pp < VW / 2; ++pp) {
                const int i = (c * 16 + r) * VW + 2 * pp;
                const float4 lo = *reinterpret_cast<const float4*>(wd + (long)i * H + k0);
                const float4 hi = *reinterpret_cast<const float4*>(wd + (long)(i + 1) * H + k0);
                // accumulator slot s of the lane at quad position qp works for unit (s + qp) & 3: see the reduction below
                auto pick = [](const float4& v, int u) { return u == 0 ? v.x : u == 1 ? v.y : u == 2 ? v.z : v.w; };
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) wt[sl][c * (VW / 2) + pp] = f32x2{pick(lo, (sl + (r & 3)) & 3), pick(hi, (sl + (r & 3)) & 3)};
            }
    }
    for (long i = (long)len * 3 * H + tid; i < (long)T * 3 * H; i += NT) {  // padded frames feed the time-batched GEMMs as zeros
        const long t = i / (3 * H), c = i % (3 * H);
        const long o = (((long)b * T + t) * 2 + dir) * 3 * H + c;
        if constexpr (!TOK) dgi[o] = 0.f;
        dgh[o] = 0.f;
    }
    float* part_wg = nullptr;
    if constexpr (TOK) {
        for (int i = tid; i < V * 3 * H; i += NT) tab[i] = 0.f;
        // the utterance's tokens as table offsets, staged once: a per-step global (or scalar) load of the token put a
        // memory round trip on the step (+0.3 us, measured); an LDS broadcast read one step ahead costs nothing
        int* toff = reinterpret_cast<int*>(tab + (long)V * 3 * H + NT);
        // ids are clamped into [0, V): the table below is read-modify-written every step, so an id from a mismatched
        // vocabulary must not be able to address LDS outside it (as_artspeech_fwd counts such ids for the host: ws token flag)
        for (int t = tid; t < len; t += NT) toff[t] = (int)min(max(tokens[(long)b * tok_stride + t], (int64_t)0), (int64_t)V - 1) * (3 * H);
        part_wg = part + (long)b * V * 6 * H + (long)dir * 3 * H;   // + v * 6H + column
        if (len <= 0) {
            for (int i = tid; i < V * 3 * H; i += NT) part_wg[(long)(i / (3 * H)) * 6 * H + i % (3 * H)] = 0.f;
            return;
        }
        // the table is first touched after the first barrier of the loop below
    } else {
        if (len <= 0) return;
    }

    const int t0 = dir ? 0 : len - 1;  // opposite to the forward walk
    const int dt = dir ? 1 : -1;
    // uniform base pointer + 32-bit BYTE offset per access (gru_fwd_kernel): frames count < 2^32 / (8H * 4), checked by the launcher
    const char* g_c = reinterpret_cast<const char*>(gates);
    const char* y_c = reinterpret_cast<const char*>(y);
    const char* dy_c = reinterpret_cast<const char*>(dy);
    char* dgi_c = reinterpret_cast<char*>(dgi);
    char* dgh_c = reinterpret_cast<char*>(dgh);
    const unsigned gto = (unsigned)(dir * 4 * H + k) * 4u;   // + frame * 8H * 4; planes at +0, +H, +2H, +3H floats
    const unsigned yo = (unsigned)(dir * H + k) * 4u;        // + frame * 2H * 4   (y and dy)
    const int sel = pl < 2 ? pl : 2;                     // planes r, z, n; rows 2 and 3 of the quad store the same n word
    const unsigned dgo = (unsigned)(dir * 3 * H + sel * H + k) * 4u;   // + frame * 6H * 4   (dgi and dgh)
    struct In { float r, z, n, hn, hprev, dyv; bool has_prev; };
    auto load = [&](long fr, bool has_prev) {
        In v;
        const unsigned f = (unsigned)fr;
        const float* gp = reinterpret_cast<const float*>(g_c + (f * (8u * H * 4u) + gto));
        v.r = gp[0]; v.z = gp[H]; v.n = gp[2 * H]; v.hn = gp[3 * H];
        v.hprev = *reinterpret_cast<const float*>(y_c + ((f + (unsigned)(has_prev ? dt : 0)) * (2u * H * 4u) + yo));
        v.has_prev = has_prev;
        v.dyv = *reinterpret_cast<const float*>(dy_c + (f * (2u * H * 4u) + yo));
        return v;
    };
    // Reduce-scatter of the four partial sums over the 16 lanes of a DPP row: the lane at quad position qp needs only the
    // total of unit qp.  Its slot s holds unit (s + qp) & 3, so the partial for unit qp of the lane t positions further in the
    // quad sits in that lane's slot (4 - t) & 3: three rotating quad_perm reads, then row_ror 4 and 8 (which keep the quad
    // position) -- 5 DPP adds instead of 16 moves + 16 adds + the select that an all-reduce of all four sums needs.
    auto dpp_add = [](float acc, float v, auto ctrl) {
        return acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    const long fbase = (long)b * T + t0;   // frame of step u: fbase + u * dt
    float dh = 0.f;
    const int tmask = pl < 3 ? -1 : 0;                                                      // dummy word: same for every token
    float* tabk = pl < 3 ? tab + sel * H + k : tab + (long)V * 3 * H + tid;                 // + (token offset & tmask)
    const int* toff = reinterpret_cast<const int*>(tab + (long)V * 3 * H + NT);
    int toff_cur = 0;
    if constexpr (TOK) {
        __syncthreads();   // offsets staged, table zeroed
        toff_cur = toff[t0];
    }
    // Operands of step u (clamped to the last step: the look-ahead stays inside the sequence); h_prev exists for all but
    // the last step of the backward walk.
    auto load_step = [&](int u) {
        const int uc = u < len ? u : len - 1;
        return load(fbase + (long)uc * dt, uc + 1 < len);
    };
    // One recurrent step: consumes `ci` (loaded AHEAD steps ago), starts the loads of step s + AHEAD into `fill`.
    // bulk: step s + AHEAD + 1 exists, so neither the look-ahead's clamp nor the h_prev selects are needed (scalar compares,
    // selects and multiplies that the general form pays on every step)
    auto step = [&](auto bulk, auto curc, int s, const In& ci, In& fill) {
        constexpr bool BULK = decltype(bulk)::value;
        const int cur = curc;
        const long fr = fbase + (long)s * dt;
        if constexpr (BULK) fill = load(fbase + (long)(s + AHEAD) * dt, true);
        else fill = load_step(s + AHEAD);

        // Every output of the step is dht times a COEFFICIENT that depends on the saved gates only (known two steps ago):
        //   dnt = dht (1-z)(1-n^2),  g_hn = dnt r,  g_r = dnt hn r (1-r),  g_z = dht (h_prev - n) z (1-z).
        // The coefficients (and the choice of this lane's plane among them) are off the recurrence's dependency chain; what
        // follows the arrival of dh is one add and two multiplies instead of a six-deep product chain.
        const float rg = ci.r, z = ci.z, n = ci.n, hn = ci.hn;
        const float omz = 1.f - z;
        const float c_n = omz * (1.f - n * n);
        const float c_hn = c_n * rg;
        const float c_r = c_hn * hn * (1.f - rg);
        const float c_z = (((BULK || ci.has_prev) ? ci.hprev : 0.f) - n) * z * omz;
        const float c_rz = pl == 0 ? c_r : c_z;     // lanes of planes r, z, n (two of them): selects on loop-invariant masks
        const float c_vi = pl >= 2 ? c_n : c_rz;
        const float c_vh = pl >= 2 ? c_hn : c_rz;
        const float dht = dh + ci.dyv;
        const float vi = dht * c_vi;
        const float vh = dht * c_vh;
        gbuf[cur][sel * H + k] = vh;
        const unsigned dfo = (unsigned)fr * (6u * H * 4u) + dgo;
        if constexpr (!TOK) *reinterpret_cast<float*>(dgi_c + dfo) = vi;
        *reinterpret_cast<float*>(dgh_c + dfo) = vh;
        __syncthreads();

        f32x2 a[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        if constexpr (VW == 4) {
            const float4* gq = reinterpret_cast<const float4*>(gbuf[cur]);
            float4 gv[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) gv[c] = gq[c * 16 + r];
            // (Round 3 tried ONE no-return LDS atomic, ds_add_f32, issued as soon as vi exists, in place of the read + write:
            // 2508 instead of 1415 cycles per step -- a 64-lane float atomic occupies the LDS for about a thousand cycles
            // and the step's barrier waits for it.  Not kept.)
            // TOK: the table word of this step's token is read BEHIND the gate reads (LDS answers in order) and written
            // back after the FMAs below: its round trip hides under them instead of standing before them.  The fourth lane
            // of a unit (a duplicate of the n plane) works on a private dummy word, so no lane is masked off.
            float* tw = nullptr;
            float told = 0.f;
            int toff_nxt = 0;
            if constexpr (TOK) {
                tw = tabk + (toff_cur & tmask);
                told = *tw;
                toff_nxt = toff[t0 + (s + 1 < len ? s + 1 : s) * dt];   // broadcast read, used one step later
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f32x2 lo = {gv[c].x, gv[c].y}, hi = {gv[c].z, gv[c].w};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    a[kk] = __builtin_elementwise_fma(wt[kk][2 * c], lo, a[kk]);
                    a[kk] = __builtin_elementwise_fma(wt[kk][2 * c + 1], hi, a[kk]);
                }
            }
            if constexpr (TOK) {
                __builtin_amdgcn_sched_barrier(0);
                *tw = told + vi;
                toff_cur = toff_nxt;
            }
        } else {
            const float2* gq = reinterpret_cast<const float2*>(gbuf[cur]);
            float2 gv[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) gv[c] = gq[c * 16 + r];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f32x2 g2 = {gv[c].x, gv[c].y};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) a[kk] = __builtin_elementwise_fma(wt[kk][c], g2, a[kk]);
            }
            if constexpr (TOK) {
                tabk[toff_cur & tmask] += vi;
                toff_cur = toff[t0 + (s + 1 < len ? s + 1 : s) * dt];
            }
        }
        float acc = a[0].x + a[0].y;
        acc = dpp_add(acc, a[3].x + a[3].y, std::integral_constant<int, 0x39>{});   // quad_perm [1,2,3,0]
        acc = dpp_add(acc, a[2].x + a[2].y, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
        acc = dpp_add(acc, a[1].x + a[1].y, std::integral_constant<int, 0x93>{});   // quad_perm [3,0,1,2]
        acc = dpp_add(acc, acc, std::integral_constant<int, 0x124>{});              // row_ror 4
        acc = dpp_add(acc, acc, std::integral_constant<int, 0x128>{});              // row_ror 8
        dh = dht * z + acc;
        // the next step writes another buffer of gbuf, whose readers all passed the barrier above: one barrier per step
    };
    constexpr std::integral_constant<int, 0> c0{};
    constexpr std::integral_constant<int, 1> c1{};
    constexpr std::integral_constant<int, 2> c2{};
    if constexpr (AHEAD == 1) {
        In a = load_step(0), bnx;
        for (int s = 0; s < len; ++s) {
            step(std::false_type{}, s % 3, s, a, bnx);
            a = bnx;
        }
    } else if constexpr (AHEAD == 3) {
        In a = load_step(0), bq = load_step(1), c = load_step(2), d;
        int s = 0;
        for (; s + 3 < len; s += 4) {
            step(std::false_type{}, s % 3, s, a, d);
            step(std::false_type{}, (s + 1) % 3, s + 1, bq, a);
            step(std::false_type{}, (s + 2) % 3, s + 2, c, bq);
            step(std::false_type{}, (s + 3) % 3, s + 3, d, c);
        }
        if (s < len) step(std::false_type{}, s % 3, s, a, d);
        if (s + 1 < len) step(std::false_type{}, (s + 1) % 3, s + 1, bq, a);
        if (s + 2 < len) step(std::false_type{}, (s + 2) % 3, s + 2, c, bq);
    } else {
        // Two steps of look-ahead: three operand sets whose roles rotate by NAME through a loop unrolled by three (a
        // register-to-register rotation would be a consumer of the newest loads and bring their wait back to this step)
        In a = load_step(0), bq = load_step(1), c;
        int s = 0;
        constexpr std::true_type yes{};
        constexpr std::false_type no{};
        for (; s + 5 < len; s += 3) {   // every step of the group has s + AHEAD + 1 < len
            step(yes, c0, s, a, c);
            step(yes, c1, s + 1, bq, a);
            step(yes, c2, s + 2, c, bq);
        }
        for (; s + 2 < len; s += 3) {
            step(no, c0, s, a, c);
            step(no, c1, s + 1, bq, a);
            step(no, c2, s + 2, c, bq);
        }
        if (s < len) step(no, c0, s, a, c);
        if (s + 1 < len) step(no, c1, s + 1, bq, a);
    }
    if constexpr (TOK) {
        __syncthreads();
        for (int i = tid; i < V * 3 * H; i += NT) part_wg[(long)(i / (3 * H)) * 6 * H + i % (3 * H)] = tab[i];
    }
    if (dbg != nullptr && threadIdx.x == 0) {
        unsigned long long* d = dbg + 4L * (blockIdx.y * gridDim.x + blockIdx.x);
        d[0] = __builtin_amdgcn_s_memtime() - t_start;
        d[1] = __builtin_amdgcn_s_memrealtime() - r_start;
        d[2] = (unsigned long long)len;
        d[3] = r_start;
    }
}


// ---- any hidden size (nn.GRU takes any, encoder_decoder/models.py:100-111): plain kernels for the sizes the register-
// resident ones above are not built for (they hold W_hh in 96 VGPRs per lane at H = 128; at H = 256 it would be 192).  One
// workgroup of 1024 threads per (utterance, direction) as above, h in LDS, W_hh streamed from L2 every step (768 KB per step
// and workgroup at H = 256).  Same gates / y / dgi / dgh layouts and the same packed-sequence semantics; the reduction order
// over k differs from the kernels above in the last bits.  Several times slower per step -- a correct fallback, not a tuned
// path (measured: DESIGN.md 8).
constexpr int GEN_THREADS = 1024;

// Forward: four adjacent lanes share a hidden unit (j = tid / 4); lane q takes the 16-byte chunks q, q + 4, ... of the unit's
// three W_hh rows -- the four lanes read 64 consecutive bytes of a row -- four chunks (12 global loads) in flight per pass,
// and the three dot products are finished with the quad DPP sum of the kernels above.  H % 4 == 0.
template <bool TRAIN, bool TOK>
__global__ __launch_bounds__(GEN_THREADS) void gru_fwd_generic_kernel(const float* __restrict__ gi, const int64_t* __restrict__ tokens,
                                                                     long tok_stride, const float* __restrict__ w_hh,
                                                                     const float* __restrict__ b_hh, const int* __restrict__ lengths,
                                                                     int T, int H, float* __restrict__ y, float* __restrict__ gates,
                                                                     int nd, int V) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];   // h double buffer [2][H]
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int q = tid & 3;
    const int len = lengths[b];
    for (long i = (long)len * H + tid; i < (long)T * H; i += GEN_THREADS)   // pad_packed_sequence: exact zeros
        y[((long)b * T + i / H) * nd * H + dir * H + (i % H)] = 0.f;
    for (int j = tid; j < 2 * H; j += GEN_THREADS) gsm[j] = 0.f;
    __syncthreads();
    if (len <= 0) return;
    const float* wd = w_hh + (long)dir * 3 * H * H;
    const float* bd = b_hh + (long)dir * 3 * H;
    const int nch = H >> 2;   // 16-byte chunks per row
    for (int s = 0; s < len; ++s) {
        const int t = dir ? len - 1 - s : s;
        const long frame = (long)b * T + t;
        const float* hc = gsm + (s & 1) * H;
        float* hn_ = gsm + ((s & 1) ^ 1) * H;
        const float* gr;
        if (TOK) {
            int64_t v = tokens[(long)b * tok_stride + t];
            if (V > 0) v = v < 0 ? 0 : (v >= V ? V - 1 : v);
            gr = gi + (v * nd + dir) * 3L * H;
        } else {
            gr = gi + (frame * nd + dir) * 3L * H;
        }
        for (int j = tid >> 2; j < H; j += GEN_THREADS / 4) {   // (the four lanes of a quad share j: the DPP sums are whole)
            const float4* wr = reinterpret_cast<const float4*>(wd + (long)j * H);
            const float4* wz = reinterpret_cast<const float4*>(wd + (long)(H + j) * H);
            const float4* wn = reinterpret_cast<const float4*>(wd + (long)(2 * H + j) * H);
            const float4* h4 = reinterpret_cast<const float4*>(hc);
            float sr = 0.f, sz = 0.f, sn = 0.f;
            for (int c0 = q; c0 < nch; c0 += 16) {
                float4 a[4], c[4], e[4], hv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {      // branch-free: chunks beyond the row re-read its last one against a zero h
                    const int cc = c0 + 4 * u;
                    const int ci = cc < nch ? cc : nch - 1;
                    a[u] = wr[ci]; c[u] = wz[ci]; e[u] = wn[ci];
                    hv[u] = h4[ci];
                    if (cc >= nch) hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sr += a[u].x * hv[u].x + a[u].y * hv[u].y + a[u].z * hv[u].z + a[u].w * hv[u].w;
                    sz += c[u].x * hv[u].x + c[u].y * hv[u].y + c[u].z * hv[u].z + c[u].w * hv[u].w;
                    sn += e[u].x * hv[u].x + e[u].y * hv[u].y + e[u].z * hv[u].z + e[u].w * hv[u].w;
                }
            }
            sr = unit_sum<4>(sr) + bd[j];
            sz = unit_sum<4>(sz) + bd[H + j];
            sn = unit_sum<4>(sn) + bd[2 * H + j];
            const float r = as_sigmoid(gr[j] + sr);
            const float z = as_sigmoid(gr[H + j] + sz);
            const float n = as_tanh(gr[2 * H + j] + r * sn);
            const float hnew = (1.f - z) * n + z * hc[j];
            if (q == 0) {
                hn_[j] = hnew;
                y[(frame * nd + dir) * H + j] = hnew;
                if (TRAIN) {
                    float* gp = gates + (frame * nd + dir) * 4L * H + j;
                    gp[0] = r; gp[H] = z; gp[2 * H] = n; gp[3 * H] = sn;
                }
            }
        }
        __syncthreads();
    }
}

// Backward: the gate gradients of a step by one thread per hidden unit; then dh = dht * z + W_hh^T g with the 3H gate rows
// dealt over four thread groups (lanes = consecutive hidden columns: a wave reads 256 consecutive bytes of a row, eight rows
// in flight), the four partial sums meeting in LDS in a fixed order.
__global__ __launch_bounds__(GEN_THREADS) void gru_bwd_generic_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                     const float* __restrict__ gates, const float* __restrict__ w_hh,
                                                                     const int* __restrict__ lengths, int T, int H,
                                                                     float* __restrict__ dgi, float* __restrict__ dgh) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];   // g [3H], dh carried [H], dht * z [H], partial sums [4][H]
    float* gb = gsm;
    float* dhb = gsm + 3 * H;
    float* dhz = gsm + 4 * H;
    float* part = gsm + 5 * H;
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int len = lengths[b];
    for (long i = (long)len * 3 * H + tid; i < (long)T * 3 * H; i += GEN_THREADS) {   // padded frames feed time-batched GEMMs
        const long t = i / (3 * H), c = i % (3 * H);
        const long o = (((long)b * T + t) * 2 + dir) * 3 * H + c;
        dgi[o] = 0.f;
        dgh[o] = 0.f;
    }
    for (int j = tid; j < H; j += GEN_THREADS) dhb[j] = 0.f;
    __syncthreads();
    if (len <= 0) return;
    const float* wd = w_hh + (long)dir * 3 * H * H;
    // opposite to the forward walk; h_prev of frame t is the output of the frame visited NEXT (zero beyond the sequence)
    const int dt = dir ? 1 : -1;
    const int kq = tid >> 8, kl = tid & 255;   // row group (0..3), column within a block of 256
    for (int s = 0; s < len; ++s) {
        const int t = dir ? s : len - 1 - s;
        const long frame = (long)b * T + t;
        const bool has_prev = s + 1 < len;
        for (int j = tid; j < H; j += GEN_THREADS) {
            const float* gp = gates + (frame * 2 + dir) * 4L * H + j;
            const float r = gp[0], z = gp[H], n = gp[2 * H], hn = gp[3 * H];
            const float hprev = has_prev ? y[((frame + dt) * 2 + dir) * H + j] : 0.f;
            const float dht = dhb[j] + dy[(frame * 2 + dir) * H + j];
            const float dn = dht * (1.f - z);
            const float dz = dht * (hprev - n);
            const float dnt = dn * (1.f - n * n);
            const float g_r = dnt * hn * r * (1.f - r);
            const float g_z = dz * z * (1.f - z);
            const float g_hn = dnt * r;
            float* di = dgi + (frame * 2 + dir) * 3L * H + j;
            float* dh = dgh + (frame * 2 + dir) * 3L * H + j;
            di[0] = g_r; di[H] = g_z; di[2 * H] = dnt;
            dh[0] = g_r; dh[H] = g_z; dh[2 * H] = g_hn;
            gb[j] = g_r; gb[H + j] = g_z; gb[2 * H + j] = g_hn;
            dhz[j] = dht * z;
        }
        __syncthreads();
        for (int k0 = 0; k0 < H; k0 += 256) {
            const int k = k0 + kl;
            const int kc = k < H ? k : H - 1;
            float acc = 0.f;
            for (int i0 = kq; i0 < 3 * H; i0 += 32) {   // rows kq, kq + 4, ...: eight of them in flight
                float wv[8], gv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 4 * u;
                    const int ic = i < 3 * H ? i : 3 * H - 1;
                    wv[u] = wd[(long)ic * H + kc];
                    gv[u] = i < 3 * H ? gb[ic] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += gv[u] * wv[u];
            }
            if (k < H) part[kq * H + k] = acc;
        }
        __syncthreads();
        for (int k = tid; k < H; k += GEN_THREADS) dhb[k] = dhz[k] + ((part[k] + part[H + k]) + (part[2 * H + k] + part[3 * H + k]));
        __syncthreads();
    }
}

// Lanes per hidden unit.  Both layouts are built; measured at H = 128, B = 32, T = 200 (tools/bench_gru.py):
// LPU = 4 (512 threads, two waves per SIMD): forward 0.55 us/step, backward 0.70; LPU = 2 (256 threads, one
// wave per SIMD, 252 VGPRs): 0.56 / 0.77 -- the second wave hides the first one's LDS / DPP / transcendental
// latencies about as well as halving the redundant gate math helps, so the 4-lane layout stays.
constexpr int lpu_of(int) { return 4; }

}  // namespace

// Exclusive CUs for the recurrences.  Their 64 long-lived workgroups share the chip with throughput-bound side work (weight
// gradients on other streams); side workgroups that land on a recurrence's CU take issue slots and LDS bandwidth from its
// dependent chain (layer-0 backward beside two 64 x 64-tile GEMMs: 1387 -> 1296 cycles per step without them, 124 -> 117 us).
// A CU mask on the side streams keeps them away and doubles the step (masked queues launch slowly); instead every
// recurrence workgroup asks for 144 KB of dynamic LDS it does not touch, so that nothing that needs LDS fits beside it and
// the dispatcher places the side work on the other CUs.  Only while the grid leaves at least half of the CUs free;
// ARTSPEECH_GRU_SHARED_CUS=1 turns it off.
constexpr size_t GRU_LDS_PAD = 144 * 1024, GRU_LDS_ATTR = 152 * 1024;
static size_t gru_lds_pad(int workgroups) {
    static const bool off = getenv("ARTSPEECH_GRU_SHARED_CUS") != nullptr;
    if (off) return 0;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n;
    }();
    return workgroups * 2 <= cus ? GRU_LDS_PAD : 0;
}
// dynamic LDS beyond 64 KB needs the attribute, once per kernel; false: the kernel keeps the 64 KB limit (no padding then)
template <typename K>
static bool gru_lds_attr(K kernel) {
    static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU_LDS_ATTR);
    return e == hipSuccess;
}

static unsigned long long* g_gru_dbg = nullptr;
static unsigned g_gru_dbg_launch = 0;   // stamps of consecutive backward launches alternate between two halves of the buffer
constexpr int AS_GRU_TOK_LDS_MAX = 128 * 1024;   // token-sum table of the layer-0 backward recurrence (one workgroup per CU)
extern "C" void as_gru_debug_stamps(uint64_t* buf) { g_gru_dbg = (unsigned long long*)buf; g_gru_dbg_launch = 0; }

static int gru_fwd_launch(const float* gi, const int64_t* tokens, int64_t tok_stride, const float* w_hh, const float* b_hh,
                          const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* y, float* gates, int nd, void* stream,
                          int V = 0) {
    AS_REQUIRE(gi && w_hh && b_hh && lengths && y, AS_ERR_BAD_ARG, "as_gru_fwd: null pointer");
    AS_REQUIRE(B > 0 && T > 0, AS_ERR_BAD_ARG, "as_gru_fwd: B=%d T=%d", B, T);
    AS_REQUIRE(!tokens || T <= 32768, AS_ERR_UNSUPPORTED, "as_gru_fwd: T=%d > 32768 with a token table", T);
    AS_REQUIRE((long)B * T * nd * 4 * H * 4 < (1L << 32), AS_ERR_UNSUPPORTED, "as_gru_fwd: B*T=%ld frames exceed the 32-bit offsets", (long)B * T);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B, nd);
    const size_t pad = gru_lds_pad(B * nd);
    const size_t need = tokens ? (size_t)T * sizeof(int) : 0;
#ifdef AS_DIAG
    static const int ahead = AS_DIAG_INT("AS_GRU_AHEAD", 2);   // look-ahead of the operand loads (steps)
#endif
#ifdef AS_DIAG   // the one-step look-ahead instantiation exists in the diagnostic build only
#define AS_GRU_LAUNCH(HH, TR, TK)                                                                                         \
    do {                                                                                                                  \
        const size_t shm = pad > need && gru_lds_attr(gru_fwd_kernel<HH, lpu_of(HH), TR, TK, 1>) &&                       \
                           gru_lds_attr(gru_fwd_kernel<HH, lpu_of(HH), TR, TK, 2>) ? pad : need;                          \
        if (ahead == 1)                                                                                                   \
            hipLaunchKernelGGL((gru_fwd_kernel<HH, lpu_of(HH), TR, TK, 1>), grid, dim3(lpu_of(HH) * HH), shm, st, gi, tokens, \
                               (long)tok_stride, w_hh, b_hh, lengths, T, y, gates, nd, V);                                   \
        else                                                                                                              \
            hipLaunchKernelGGL((gru_fwd_kernel<HH, lpu_of(HH), TR, TK, 2>), grid, dim3(lpu_of(HH) * HH), shm, st, gi, tokens, \
                               (long)tok_stride, w_hh, b_hh, lengths, T, y, gates, nd, V);                                   \
    } while (0)
#else
#define AS_GRU_LAUNCH(HH, TR, TK)                                                                                         \
    do {                                                                                                                  \
        const size_t shm = pad > need && gru_lds_attr(gru_fwd_kernel<HH, lpu_of(HH), TR, TK, 2>) ? pad : need;            \
        hipLaunchKernelGGL((gru_fwd_kernel<HH, lpu_of(HH), TR, TK, 2>), grid, dim3(lpu_of(HH) * HH), shm, st, gi, tokens, \
                           (long)tok_stride, w_hh, b_hh, lengths, T, y, gates, nd, V);                                    \
    } while (0)
#endif
#define AS_GRU_FWD(HH)                                      \
    if (gates && tokens) AS_GRU_LAUNCH(HH, true, true);     \
    else if (gates) AS_GRU_LAUNCH(HH, true, false);         \
    else if (tokens) AS_GRU_LAUNCH(HH, false, true);        \
    else AS_GRU_LAUNCH(HH, false, false)
    switch (H) {
        case 32: AS_GRU_FWD(32); break;
        case 64: AS_GRU_FWD(64); break;
        case 128: AS_GRU_FWD(128); break;
        default: {   // any other hidden size: the plain kernels
            const size_t gshm = (size_t)2 * H * sizeof(float);
            AS_REQUIRE(H > 0 && H % 4 == 0 && gshm <= 64 * 1024, AS_ERR_UNSUPPORTED, "as_gru_fwd: hidden size %d (a multiple of 4 up to 8192)", H);
#define AS_GRU_GEN(TR, TK) \
    hipLaunchKernelGGL((gru_fwd_generic_kernel<TR, TK>), grid, dim3(GEN_THREADS), gshm, st, gi, tokens, (long)tok_stride, w_hh, b_hh, lengths, T, H, y, gates, nd, V)
            if (gates && tokens) AS_GRU_GEN(true, true);
            else if (gates) AS_GRU_GEN(true, false);
            else if (tokens) AS_GRU_GEN(false, true);
            else AS_GRU_GEN(false, false);
#undef AS_GRU_GEN
        }
    }
#undef AS_GRU_FWD
#undef AS_GRU_LAUNCH
    AS_LAUNCH_CHECK("as_gru_fwd");
    return 0;
}

extern "C" int as_gru_bidir_fwd(const float* gi, const int64_t* tokens, int64_t tok_stride, const float* w_hh,
                                const float* b_hh, const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* y,
                                float* gates, void* stream) {
    return gru_fwd_launch(gi, tokens, tok_stride, w_hh, b_hh, lengths, B, T, H, y, gates, 2, stream);
}

// internal (gemm_internal.h): the token-table form with the vocabulary size, so that ids are clamped into the table
int as_gru_bidir_fwd_tokens(const float* table, const int64_t* tokens, int64_t tok_stride, int32_t V, const float* w_hh,
                            const float* b_hh, const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* y, float* gates,
                            hipStream_t st) {
    return gru_fwd_launch(table, tokens, tok_stride, w_hh, b_hh, lengths, B, T, H, y, gates, 2, st, V);
}

extern "C" int as_gru_unidir_fwd(const float* gi, const float* w_hh, const float* b_hh, const int32_t* lengths, int32_t B,
                                 int32_t T, int32_t H, float* y, void* stream) {
    return gru_fwd_launch(gi, nullptr, 0, w_hh, b_hh, lengths, B, T, H, y, nullptr, 1, stream);
}

static int gru_bwd_launch(const float* dy, const float* y, const float* gates, const float* w_hh, const int32_t* lengths,
                          int32_t B, int32_t T, int32_t H, float* dgi, float* dgh, const int64_t* tokens, int64_t tok_stride,
                          int32_t V, float* part, void* stream) {
    AS_REQUIRE(dy && y && gates && w_hh && lengths && (dgi || tokens) && dgh, AS_ERR_BAD_ARG, "as_gru_bidir_bwd: null pointer");
    AS_REQUIRE(B > 0 && T > 0, AS_ERR_BAD_ARG, "as_gru_bidir_bwd: B=%d T=%d", B, T);
    AS_REQUIRE(!tokens || (part && V > 0 && tok_stride >= T), AS_ERR_BAD_ARG, "as_gru_bidir_bwd: token table arguments");
    AS_REQUIRE((long)B * T * 8 * H * 4 < (1L << 32), AS_ERR_UNSUPPORTED, "as_gru_bidir_bwd: B*T=%ld frames exceed the 32-bit offsets", (long)B * T);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(B, 2);
    // diagnostic stamps (as_gru_debug_stamps): 2 x (2 B workgroups x 4 words); launch k writes half k % 2, so that the two
    // backward recurrences of one training step (layer 1, then layer 0) can both be read afterwards
    unsigned long long* dbg_now = g_gru_dbg ? g_gru_dbg + (size_t)(g_gru_dbg_launch++ & 1u) * 8u * (size_t)B : nullptr;
#ifdef AS_DIAG
    static const bool unit_layout = AS_DIAG_SET("AS_GRU_BWD_UNIT");  // ablation: the 4-lanes-per-unit layout
#endif
    hipEvent_t stop_ev = (H == 32 || H == 64 || H == 128) ? as_stop_event_take() : nullptr;
    const size_t pad = gru_lds_pad(B * 2);
    const size_t need = tokens ? ((size_t)V * 3 * H + 4 * H + T) * sizeof(float) : 0;   // + one dummy word per lane + T offsets
#ifdef AS_DIAG
    static const int ahead = AS_DIAG_INT("AS_GRU_AHEAD", 2);   // look-ahead of the operand loads (steps)
#endif
#define AS_GRU_BWD_ROW(HH, TK, AH)                                                                                            \
    do {                                                                                                                      \
        const bool big = gru_lds_attr(gru_bwd_row_kernel<HH, TK, AH>);                                                        \
        AS_REQUIRE(big || need <= 64 * 1024, AS_ERR_UNSUPPORTED, "as_gru_bidir_bwd: cannot reserve %zu bytes of LDS", need);   \
        const size_t shm = big && pad > need ? pad : need;                                                                    \
        if (stop_ev)   /* a fork event rides on this dispatch (gemm_internal.h, as_stop_event_set) */                         \
            hipExtLaunchKernelGGL((gru_bwd_row_kernel<HH, TK, AH>), grid, dim3(4 * HH), (unsigned)shm, st, nullptr, stop_ev, 0, dy, y, \
                                  gates, w_hh, lengths, T, dgi, dgh, dbg_now, tokens, (long)tok_stride, V, part);             \
        else                                                                                                                  \
            hipLaunchKernelGGL((gru_bwd_row_kernel<HH, TK, AH>), grid, dim3(4 * HH), shm, st, dy, y, gates, w_hh,             \
                               lengths, T, dgi, dgh, dbg_now, tokens, (long)tok_stride, V, part);                             \
    } while (0)
#ifdef AS_DIAG   // other look-aheads and the 4-lanes-per-unit layout exist in the diagnostic build only
#define AS_GRU_BWD(HH)                                                                                                        \
    if (tokens) {                                                                                                             \
        if (ahead == 3) AS_GRU_BWD_ROW(HH, true, 3); else if (ahead == 2) AS_GRU_BWD_ROW(HH, true, 2); else AS_GRU_BWD_ROW(HH, true, 1); \
    } else if (unit_layout) {                                                                                                 \
        hipLaunchKernelGGL((gru_bwd_kernel<HH, lpu_of(HH)>), grid, dim3(lpu_of(HH) * HH), 0, st, dy, y, gates, w_hh, lengths, \
                           T, dgi, dgh);                                                                                      \
        if (stop_ev) (void)hipEventRecord(stop_ev, st);                                                                       \
    }                                                                                                                         \
    else if (ahead == 3) AS_GRU_BWD_ROW(HH, false, 3); else if (ahead == 2) AS_GRU_BWD_ROW(HH, false, 2); else AS_GRU_BWD_ROW(HH, false, 1)
#else
#define AS_GRU_BWD(HH) \
    if (tokens) AS_GRU_BWD_ROW(HH, true, 2); else AS_GRU_BWD_ROW(HH, false, 2)
#endif
    switch (H) {
        case 32: AS_GRU_BWD(32); break;
        case 64: AS_GRU_BWD(64); break;
        case 128: AS_GRU_BWD(128); break;
        default: {   // any other hidden size: the plain kernel (never with a token table: as_gru_bwd_tokens_fits)
            AS_REQUIRE(!tokens && dgi, AS_ERR_UNSUPPORTED, "as_gru_bidir_bwd: token sums need a hidden size in {32, 64, 128}");
            const size_t gshm = (size_t)9 * H * sizeof(float);
            AS_REQUIRE(H > 0 && H % 4 == 0 && gshm <= 64 * 1024, AS_ERR_UNSUPPORTED, "as_gru_bidir_bwd: hidden size %d (a multiple of 4 up to 1820)", H);
            hipLaunchKernelGGL(gru_bwd_generic_kernel, grid, dim3(GEN_THREADS), gshm, st, dy, y, gates, w_hh, lengths, T, H, dgi, dgh);
        }
    }
#undef AS_GRU_BWD
#undef AS_GRU_BWD_ROW
    AS_LAUNCH_CHECK("as_gru_bidir_bwd");
    return 0;
}

extern "C" int as_gru_bidir_bwd(const float* dy, const float* y, const float* gates, const float* w_hh,
                                const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* dgi, float* dgh,
                                void* stream) {
    AS_REQUIRE(dgi, AS_ERR_BAD_ARG, "as_gru_bidir_bwd: null pointer");
    return gru_bwd_launch(dy, y, gates, w_hh, lengths, B, T, H, dgi, dgh, nullptr, 0, 0, nullptr, stream);
}

// Layer 0 under a token table (internal, gemm_internal.h): per-utterance token sums of the input-side gate gradients,
// part [B][V][6H], instead of dgi.  0 = not a case for it (V * 3H floats must fit the LDS budget): the caller takes
// as_gru_bidir_bwd + as_token_segsum.
bool as_gru_bwd_tokens_fits(int32_t V, int32_t H, int32_t T) {
    static const bool off = AS_DIAG_SET("AS_NO_GRU_TOKSUM");   // ablation: dgi + the segmented-sum kernel
    return !off && (H == 32 || H == 64 || H == 128) && ((long)V * 3 * H + 4 * H + T) * (long)sizeof(float) <= AS_GRU_TOK_LDS_MAX;
}
int as_gru_bidir_bwd_tokens(const float* dy, const float* y, const float* gates, const float* w_hh, const int32_t* lengths,
                            int32_t B, int32_t T, int32_t H, float* dgh, const int64_t* tokens, int64_t tok_stride, int32_t V,
                            float* part, hipStream_t st) {
    if (!tokens || !as_gru_bwd_tokens_fits(V, H, T)) return 0;
    const int rc = gru_bwd_launch(dy, y, gates, w_hh, lengths, B, T, H, nullptr, dgh, tokens, tok_stride, V, part, st);
    return rc == 0 ? 1 : rc;
}
