// Fused multi-head attention core for the transformer variant (nn.MultiheadAttention inside ChannelProcessingLayer,
// transformer/models.py:37-100): ctx = softmax(Q_h K_h^T / sqrt(dh) + attn_mask[b] + key_padding_mask[b]) V_h per
// (channel pair g, utterance b, head h), exact fp32 on the f32 MFMA, scores never leave the registers.
//
// The unfused path (two grouped GEMMs around as_attn_softmax) runs at ~30 TFLOP/s at T = 200, dh = 64: 200 x 200 x 64
// problems make 4 ragged 128 x 128 tiles with a 2-step reduction, and the 2.25 GB score tensor of one interaction layer
// goes to HBM and back twice.  Here one workgroup owns one (g, b, h): K_h and V_h are staged once in LDS and every wave
// takes a strip of 32 queries.
//
// Everything is computed TRANSPOSED so that no operand ever has to change layout between the two matrix products:
//   S^T[key][q]  = sum_c K[key][c] Q[q][c]      A = K rows (LDS), B = the lane's own query row (registers)
//   O^T[c][q]    = sum_key V[key][c] P^T[key][q] A = V^T (LDS),    B = P^T -- exactly the accumulator registers of S^T
// The 32x32x2 MFMA returns D[i][j] with j = lane & 31 and i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) for register r, and
// takes B[k][j] from lane (j, k = lane >> 5).  Pairing the two reduction indices of one MFMA step as (key, key + 4) makes
// register r of an S^T accumulator -- key_r in the lower half-wave, key_r + 4 in the upper -- the B operand of the second
// product as it stands.  A query is a lane column, so the softmax statistics are in-lane reductions over the accumulator
// registers plus ONE exchange between the half-waves.
#include <cmath>

#include "as_common.h"
#include "artspeech_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct AttnK {
    const float* Q; const float* K; const float* V; float* O;
    const float* mask_t;  // [B][Tk rounded up to 32][T] additive, KEY-major (transposed and padded by the caller) or nullptr
    const float* kpm;     // [B][Tk] additive or nullptr
    float* lse;           // [Z][T] row maximum + log of the row sum (for a backward that recomputes P) or nullptr
    float* probs_t;       // [Z][Tk][tp] the probabilities, KEY-major (what the unfused backward consumes), or nullptr
    int B, heads, T, Tk, d;
    int causal;           // the mask is -inf above the diagonal for every utterance: key blocks beyond a strip's own are skipped
    int tp;               // row pitch of probs_t: T rounded up to 32 floats, so that a strip's 128-byte row segment is one cache line
                          // (at the natural pitch of T = 200 floats = 800 B every segment straddles two lines)
    float scale;
};

constexpr int ATT_THREADS = 512;

// DH: head width (16 / 32 / 64).  NB: 32-key blocks held in registers (Tk <= 32 NB).
template <int DH, int NB>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(AttnK a) {
    constexpr int LD = DH + 4;                 // LDS row stride: 16-byte aligned rows, conflict-free b128 row reads
    constexpr int HS = DH / 2;                 // reduction steps of S^T: step s pairs columns (s, s + HS)
    constexpr int OB = (DH + 31) / 32;         // 32-row blocks of O^T
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                          // [32 NB][LD]
    float* Vs = smem + 32 * NB * LD;           // [32 NB][LD]
    float* kp = Vs + 32 * NB * LD;             // [32 NB] key padding mask (0 when absent, -inf beyond Tk)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const long z = blockIdx.x;
    const int h = (int)(z % a.heads);
    const int b = (int)((z / a.heads) % a.B);
    const long g = z / ((long)a.heads * a.B);
    const long qbase = ((g * a.B + b) * (long)a.T) * a.d + (long)h * DH;
    const long kbase = ((g * a.B + b) * (long)a.Tk) * a.d + (long)h * DH;

    // stage K_h, V_h (zero rows beyond Tk) and the key padding mask
    constexpr int V4 = DH / 4;
    for (int i = tid; i < 32 * NB * V4; i += ATT_THREADS) {
        const int row = i / V4, c4 = i - row * V4;
        const bool ok = row < a.Tk;
        const long off = kbase + (long)(ok ? row : 0) * a.d + c4 * 4;
        float4 kv = *reinterpret_cast<const float4*>(a.K + off);
        float4 vv = *reinterpret_cast<const float4*>(a.V + off);
        if (!ok) kv = vv = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Ks + row * LD + c4 * 4) = kv;
        *reinterpret_cast<float4*>(Vs + row * LD + c4 * 4) = vv;
    }
    for (int i = tid; i < 32 * NB; i += ATT_THREADS)
        kp[i] = i < a.Tk ? (a.kpm ? a.kpm[(long)b * a.Tk + i] : 0.f) : -INFINITY;
    __syncthreads();

    const float* vl = Vs + 4 * lh * LD + l31;  // this lane's corner of V^T: row 4 lh (+ key_r), column l31 (+ 32 ob)
    const float* kpl = kp + 4 * lh;
    const int strips = (a.T + 31) / 32;
    // Causal mask: strip s only has the key blocks 0 .. s (the others hold exact zeros: no MFMAs, no mask loads for them), so
    // the strips cost 1 .. strips blocks.  Waves w and w + 4 share a SIMD: the strips are dealt so that every SIMD gets about
    // the same number of blocks (at 7 strips: {6,0} {5,1} {4,2} {3} = 8, 8, 8, 4 blocks instead of 14 each).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool balanced = a.causal && strips <= ATT_THREADS / 64;
    for (int it = wave_u; it < (balanced ? ATT_THREADS / 64 : strips); it += ATT_THREADS / 64) {
        int strip = it;
        if (balanced) {   // waves 0-3: the four longest strips, descending; waves 4-7: the short ones, ascending; one strip per wave
            strip = it < 4 ? strips - 1 - it : (it - 4 <= strips - 5 ? it - 4 : -1);
            if (strip < 0) break;
        }
        const int q = strip * 32 + l31;
        const int qc = q < a.T ? q : a.T - 1;
        const int last_blk = a.causal ? min(strip, NB - 1) : NB - 1;   // (wave-uniform) the last key block that holds anything
        // the lane's query row, the half of it this half-wave feeds to the MFMA: columns lh*HS + s
        float qv[HS];
        {
            const float* qp = a.Q + qbase + (long)qc * a.d + lh * HS;
#pragma unroll
            for (int s = 0; s < HS; s += 4) {
                const float4 t = *reinterpret_cast<const float4*>(qp + s);
                qv[s] = t.x; qv[s + 1] = t.y; qv[s + 2] = t.z; qv[s + 3] = t.w;
            }
        }
        f32x16 sacc[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            if (blk > last_blk) {   // masked for every query of the strip: the scores are -inf, the probabilities exact zeros
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[blk][r] = -INFINITY;
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[blk][r] = 0.f;
            float kv[HS];
            const float* kr = Ks + (blk * 32 + l31) * LD + lh * HS;
#pragma unroll
            for (int s = 0; s < HS; s += 4) {
                const float4 t = *reinterpret_cast<const float4*>(kr + s);
                kv[s] = t.x; kv[s + 1] = t.y; kv[s + 2] = t.z; kv[s + 3] = t.w;
            }
#pragma unroll
            for (int s = 0; s < HS; ++s) sacc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[s], qv[s], sacc[blk], 0, 0, 0);
            // keep the blocks apart: left alone, the scheduler hoists every block's LDS reads to the top and spills
            __builtin_amdgcn_sched_barrier(0);
        }
        // scores -> scale + masks; column (= query) maximum
        float m = -INFINITY;
        if (a.mask_t) {  // uniform branch around the whole unrolled block, not one per element
            // uniform row pointer + one per-lane 32-bit offset: per-element 64-bit addresses would be hoisted out of the strip
            // loop and cost 2 VGPRs each (the mask is padded to 32 NB key rows, so no clamping either)
            const float* mt = a.mask_t + (long)b * (32 * NB) * a.T;
            const int moff = qc + 4 * lh * a.T;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {  // 16 loads in flight per block (all NB x 16 at once would not fit the registers)
                if (blk > last_blk) continue;
                float mv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int ku = blk * 32 + (r & 3) + 8 * (r >> 2);
                    asm volatile("" : "+s"(ku));   // (row pointers recomputed on the scalar unit, not hoisted into ~200 SGPRs)
                    mv[r] = (mt + (long)ku * a.T)[moff];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[blk][r] = sacc[blk][r] * a.scale + mv[r];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[blk][r] *= a.scale;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = sacc[blk][r] + kpl[blk * 32 + (r & 3) + 8 * (r >> 2)];  // -inf beyond Tk
                sacc[blk][r] = x;
                m = fmaxf(m, x);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(sacc[blk][r] - m);  // a fully masked query: -inf - -inf = NaN, like PyTorch
                sacc[blk][r] = e;
                l += e;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        // O^T = V^T P^T: register r of block blk is the B operand for the key pair (key_r, key_r + 4)
        f32x16 oacc[OB];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[ob][r] = 0.f;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            if (blk > last_blk) {   // zeros (NaN for a query whose whole row is masked, as everywhere else in that row)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[blk][r] *= inv;
                continue;
            }
            float vv[16][OB];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int ob = 0; ob < OB; ++ob) {  // lane base + compile-time offset: an immediate of the ds_read
                    const float t = vl[(blk * 32 + (r & 3) + 8 * (r >> 2)) * LD + ob * 32];
                    vv[r][ob] = ob * 32 + l31 < DH ? t : 0.f;
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = sacc[blk][r] * inv;
                sacc[blk][r] = p;
#pragma unroll
                for (int ob = 0; ob < OB; ++ob) oacc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r][ob], p, oacc[ob], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (a.probs_t) {  // training: P^T rows are contiguous in q, i.e. across the lanes (uniform row pointer + lane offset)
            float* pz = a.probs_t + z * (long)a.Tk * a.tp;
            const int poff = q + 4 * lh * a.tp;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int ku = blk * 32 + (r & 3) + 8 * (r >> 2);
                    asm volatile("" : "+s"(ku));
                    if (q < a.T && ku + 4 * lh < a.Tk) (pz + (long)ku * a.tp)[poff] = sacc[blk][r];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (q < a.T) {
            float* op = a.O + qbase + (long)q * a.d;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {  // rows c = 32 ob + 8 rr + 4 lh + (0..3): one 16-byte store
                    const int c = ob * 32 + 8 * rr + 4 * lh;
                    if (c < DH)
                        *reinterpret_cast<float4*>(op + c) =
                            make_float4(oacc[ob][4 * rr], oacc[ob][4 * rr + 1], oacc[ob][4 * rr + 2], oacc[ob][4 * rr + 3]);
                }
            if (a.lse && lh == 0) a.lse[z * a.T + q] = m + __logf(l);
        }
    }
}

// Backward, first half: dS^T = P^T o (V dctx^T - D) * scale, key-major, without materialising dP.
// Same transposed product as S^T in the forward (A = V rows from LDS, B = the lane's own dctx row), one 32-key block at a
// time: nothing has to be kept across blocks here, so the kernel needs ~100 VGPRs and 61 KB of LDS -- two 8-wave workgroups
// per CU.  D[q] = sum_c dctx[q][c] ctx[q][c] is a per-lane scalar (each half-wave sums its half of the row, one exchange).
// Replaces a 200 x 200 x 64 grouped GEMM (4 ragged tiles, 2-step reduction) + the softmax-backward sweep: the dP tensor
// (2.25 GB per interaction group) is neither written nor read.
struct AttnDsK {
    const float* V; const float* dctx; const float* ctx; const float* probs_t; float* ds_t;   // probs_t, ds_t: [Z][Tk][tp]
    int B, heads, T, Tk, d, tp;
    float scale;
};

template <int DH, int NB>
__global__ __launch_bounds__(ATT_THREADS, 4) void attn_ds_kernel(AttnDsK a) {
    constexpr int LD = DH + 4;
    constexpr int HS = DH / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vs = smem;  // [32 NB][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const long z = blockIdx.x;
    const int h = (int)(z % a.heads);
    const int b = (int)((z / a.heads) % a.B);
    const long g = z / ((long)a.heads * a.B);
    const long qbase = ((g * a.B + b) * (long)a.T) * a.d + (long)h * DH;
    const long kbase = ((g * a.B + b) * (long)a.Tk) * a.d + (long)h * DH;
    constexpr int V4 = DH / 4;
    for (int i = tid; i < 32 * NB * V4; i += ATT_THREADS) {
        const int row = i / V4, c4 = i - row * V4;
        const bool ok = row < a.Tk;
        float4 v = *reinterpret_cast<const float4*>(a.V + kbase + (long)(ok ? row : 0) * a.d + c4 * 4);
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Vs + row * LD + c4 * 4) = v;
    }
    __syncthreads();
    const float* pz = a.probs_t + z * (long)a.Tk * a.tp;
    float* dz = a.ds_t + z * (long)a.Tk * a.tp;
    const int strips = (a.T + 31) / 32;
    for (int strip = wave; strip < strips; strip += ATT_THREADS / 64) {
        const int q = strip * 32 + l31;
        const int qc = q < a.T ? q : a.T - 1;
        float gv[HS];
        float D = 0.f;
        {
            const float* gp = a.dctx + qbase + (long)qc * a.d + lh * HS;
            const float* op = a.ctx + qbase + (long)qc * a.d + lh * HS;
#pragma unroll
            for (int s = 0; s < HS; s += 4) {
                const float4 t = *reinterpret_cast<const float4*>(gp + s);
                const float4 o = *reinterpret_cast<const float4*>(op + s);
                gv[s] = t.x; gv[s + 1] = t.y; gv[s + 2] = t.z; gv[s + 3] = t.w;
                D += t.x * o.x + t.y * o.y + t.z * o.z + t.w * o.w;
            }
        }
        D += __shfl_xor(D, 32, 64);
        // uniform row pointer + one of two per-lane 32-bit offsets (as in the forward: per-element 64-bit addresses would be
        // hoisted out of the strip loop, two VGPRs each).  Rows are clamped on the uniform side, so every load is in bounds.
        const int off0 = qc, off4 = qc + 4 * lh * a.tp;
        // the probabilities of block blk + 1 go in flight BEFORE the matrix work of block blk: a block's loads queue behind the
        // previous block's stores (vector memory operations retire in order), so one block of look-ahead is what lets a
        // wave's memory pipe work while its 32 MFMAs run (before: load -> MFMAs -> store, block after block: 2.2 ms per
        // 110-block group for 4.5 GB = 2 TB/s)
        auto fetch = [&](int blk, float (&pv)[16]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int ku = blk * 32 + (r & 3) + 8 * (r >> 2);
                // (recomputed per use on the scalar unit: hoisted out of the strip loop, the 2 x 16 NB row pointers are
                // ~450 SGPRs that end up in VGPR lanes and scratch)
                asm volatile("" : "+s"(ku));
                const float* row = pz + (long)(ku < a.Tk ? ku : a.Tk - 1) * a.tp;
                pv[r] = row[ku + 4 < a.Tk ? off4 : off0];
            }
        };
        auto block = [&](int blk, const float (&pv)[16]) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* vrow = Vs + (blk * 32 + l31) * LD + lh * HS;
            constexpr int VC = HS < 8 ? HS : 8;   // the V row in chunks of 8 reduction steps: 8 live registers instead of HS
#pragma unroll
            for (int s0 = 0; s0 < HS; s0 += VC) {
                float vr[VC];
#pragma unroll
                for (int s = 0; s < VC; s += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(vrow + s0 + s);
                    vr[s] = t.x; vr[s + 1] = t.y; vr[s + 2] = t.z; vr[s + 3] = t.w;
                }
#pragma unroll
                for (int s = 0; s < VC; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[s], gv[s0 + s], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int ku = blk * 32 + (r & 3) + 8 * (r >> 2);
                asm volatile("" : "+s"(ku));
                const bool both = ku + 4 < a.Tk;                    // uniform: rows ku and ku + 4 both exist
                const bool mine = both || (lh == 0 && ku < a.Tk);   // this half-wave's key row exists
                float* row = dz + (long)(ku < a.Tk ? ku : a.Tk - 1) * a.tp;
                if (mine && q < a.T) row[both ? off4 : off0] = pv[r] * (acc[r] - D) * a.scale;
            }
        };
        float pv0[16], pv1[16];
        fetch(0, pv0);
#pragma unroll
        for (int blk = 0; blk < NB; blk += 2) {
            if (blk + 1 < NB) fetch(blk + 1, pv1);
            __builtin_amdgcn_sched_barrier(0);
            block(blk, pv0);
            __builtin_amdgcn_sched_barrier(0);
            if (blk + 1 < NB) {
                if (blk + 2 < NB) fetch(blk + 2, pv0);
                __builtin_amdgcn_sched_barrier(0);
                block(blk + 1, pv1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int DH, int NB>
int launch_attn_ds(const AttnDsK& k, long Z, hipStream_t st) {
    constexpr size_t shm = (size_t)(32 * NB * (DH + 4)) * sizeof(float);
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_ds_kernel<DH, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    AS_REQUIRE(attr == hipSuccess, (int)attr, "as_attention_bwd_ds: cannot reserve %zu bytes of LDS: %s", shm, hipGetErrorString(attr));
    hipLaunchKernelGGL((attn_ds_kernel<DH, NB>), dim3((unsigned)Z), dim3(ATT_THREADS), shm, st, k);
    return 0;
}


// The same under a CAUSAL mask (P^T[key][q] == 0 for q < key): only the (query strip s, key block kb) pairs with kb <= s hold
// anything, 28 of 49 at T = Tk = 200.  With a strip per wave the wave of the last strip would still walk all its blocks and set
// the workgroup's time; a block of dS^T depends on nothing but its own strip's dctx rows and D, so the PAIRS are the work items
// here: the non-empty ones are dealt evenly over the eight waves (3.5 each), the empty ones are zero-filled (the GEMMs that
// follow read parts of them).  D[q] of all queries is computed once per workgroup into LDS.  T, Tk <= 256.
template <int DH>
__global__ __launch_bounds__(ATT_THREADS, 4) void attn_ds_causal_kernel(AttnDsK a) {
    constexpr int LD = DH + 4;
    constexpr int HS = DH / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nb = (a.Tk + 31) / 32, strips = (a.T + 31) / 32;   // <= 8 each
    float* Vs = smem;                       // [32 nb][LD]
    float* Dq = smem + 32 * nb * LD;        // [32 strips]
    int* items = reinterpret_cast<int*>(Dq + 32 * strips);   // [strips * nb] pairs s * 8 + kb, the non-empty ones first; [64] = their count
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const long z = blockIdx.x;
    const int h = (int)(z % a.heads);
    const int b = (int)((z / a.heads) % a.B);
    const long g = z / ((long)a.heads * a.B);
    const long qbase = ((g * a.B + b) * (long)a.T) * a.d + (long)h * DH;
    const long kbase = ((g * a.B + b) * (long)a.Tk) * a.d + (long)h * DH;
    constexpr int V4 = DH / 4;
    for (int i = tid; i < 32 * nb * V4; i += ATT_THREADS) {
        const int row = i / V4, c4 = i - row * V4;
        const bool ok = row < a.Tk;
        float4 v = *reinterpret_cast<const float4*>(a.V + kbase + (long)(ok ? row : 0) * a.d + c4 * 4);
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Vs + row * LD + c4 * 4) = v;
    }
    if (wave < strips) {   // D[q] = sum_c dctx[q][c] ctx[q][c] of strip `wave` (each half-wave sums its half of the row)
        const int q = wave * 32 + l31;
        const int qc = q < a.T ? q : a.T - 1;
        const float* gp = a.dctx + qbase + (long)qc * a.d + lh * HS;
        const float* op = a.ctx + qbase + (long)qc * a.d + lh * HS;
        float D = 0.f;
#pragma unroll
        for (int s = 0; s < HS; s += 4) {
            const float4 t = *reinterpret_cast<const float4*>(gp + s);
            const float4 o = *reinterpret_cast<const float4*>(op + s);
            D += t.x * o.x + t.y * o.y + t.z * o.z + t.w * o.w;
        }
        D += __shfl_xor(D, 32, 64);
        if (lh == 0) Dq[q] = D;
    }
    if (tid == 0) {
        int n = 0;
        for (int s = 0; s < strips; ++s)
            for (int kb = 0; kb < nb && kb <= s; ++kb) items[n++] = s * 8 + kb;
        items[64] = n;
        for (int s = 0; s < strips; ++s)
            for (int kb = s + 1; kb < nb; ++kb) items[n++] = s * 8 + kb;
    }
    __syncthreads();
    const int n_comp = items[64], n_total = strips * nb;
    const float* pz = a.probs_t + z * (long)a.Tk * a.tp;
    float* dz = a.ds_t + z * (long)a.Tk * a.tp;
    // A wave takes a CONTIGUOUS run of the non-empty pairs (strip-major order: at most two strips per run, so the strip's dctx rows
    // are loaded once or twice per wave) and keeps the next pair's probabilities in flight under the current pair's MFMAs: a
    // pair's loads queue behind the previous pair's stores (vector memory operations retire in order), one pair of look-ahead is
    // what keeps the memory pipe busy.  Its share of the empty pairs is zero-filled at the end.
    const int c0 = wave * n_comp / 8, c1 = (wave + 1) * n_comp / 8;
    const int nz = n_total - n_comp;
    const int z0 = n_comp + wave * nz / 8, z1 = n_comp + (wave + 1) * nz / 8;
    auto fetch = [&](int it, float (&pv)[16]) {
        const int pair = __builtin_amdgcn_readfirstlane(items[it]);
        const int s_ = pair >> 3, kb = pair & 7;
        const int q = s_ * 32 + l31;
        const int qc = q < a.T ? q : a.T - 1;
        const int off0 = qc, off4 = qc + 4 * lh * a.tp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ku = kb * 32 + (r & 3) + 8 * (r >> 2);
            const float* row = pz + (long)(ku < a.Tk ? ku : a.Tk - 1) * a.tp;
            pv[r] = row[ku + 4 < a.Tk ? off4 : off0];
        }
    };
    int cur_s = -1;
    float gv[HS];
    float D = 0.f;
    auto compute = [&](int it, const float (&pv)[16]) {
        const int pair = __builtin_amdgcn_readfirstlane(items[it]);
        const int s_ = pair >> 3, kb = pair & 7;
        const int q = s_ * 32 + l31;
        const int qc = q < a.T ? q : a.T - 1;
        const int off0 = qc, off4 = qc + 4 * lh * a.tp;
        if (s_ != cur_s) {   // (wave-uniform) a new strip: its dctx rows and D
            cur_s = s_;
            const float* gp = a.dctx + qbase + (long)qc * a.d + lh * HS;
#pragma unroll
            for (int s = 0; s < HS; s += 4) {
                const float4 t = *reinterpret_cast<const float4*>(gp + s);
                gv[s] = t.x; gv[s + 1] = t.y; gv[s + 2] = t.z; gv[s + 3] = t.w;
            }
            D = Dq[qc];
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* vrow = Vs + (kb * 32 + l31) * LD + lh * HS;
        constexpr int VC = HS < 8 ? HS : 8;
#pragma unroll
        for (int s0 = 0; s0 < HS; s0 += VC) {
            float vr[VC];
#pragma unroll
            for (int s = 0; s < VC; s += 4) {
                const float4 t = *reinterpret_cast<const float4*>(vrow + s0 + s);
                vr[s] = t.x; vr[s + 1] = t.y; vr[s + 2] = t.z; vr[s + 3] = t.w;
            }
#pragma unroll
            for (int s = 0; s < VC; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[s], gv[s0 + s], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ku = kb * 32 + (r & 3) + 8 * (r >> 2);
            const bool both = ku + 4 < a.Tk;
            const bool mine = both || (lh == 0 && ku < a.Tk);
            float* row = dz + (long)(ku < a.Tk ? ku : a.Tk - 1) * a.tp;
            if (mine && q < a.T) row[both ? off4 : off0] = pv[r] * (acc[r] - D) * a.scale;
        }
    };
    float pvA[16], pvB[16];
    if (c0 < c1) fetch(c0, pvA);
    for (int it = c0; it < c1; it += 2) {
        if (it + 1 < c1) fetch(it + 1, pvB);
        __builtin_amdgcn_sched_barrier(0);
        compute(it, pvA);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < c1) {
            if (it + 2 < c1) fetch(it + 2, pvA);
            __builtin_amdgcn_sched_barrier(0);
            compute(it + 1, pvB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int it = z0; it < z1; ++it) {   // nothing but zeros above the diagonal
        const int pair = __builtin_amdgcn_readfirstlane(items[it]);
        const int s_ = pair >> 3, kb = pair & 7;
        const int q = s_ * 32 + l31;
        const int qc = q < a.T ? q : a.T - 1;
        const int off0 = qc, off4 = qc + 4 * lh * a.tp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ku = kb * 32 + (r & 3) + 8 * (r >> 2);
            const bool both = ku + 4 < a.Tk;
            const bool mine = both || (lh == 0 && ku < a.Tk);
            float* row = dz + (long)(ku < a.Tk ? ku : a.Tk - 1) * a.tp;
            if (mine && q < a.T) row[both ? off4 : off0] = 0.f;
        }
    }
}

template <int DH>
int launch_attn_ds_causal(const AttnDsK& k, long Z, hipStream_t st) {
    const int nb = (k.Tk + 31) / 32, strips = (k.T + 31) / 32;
    const size_t shm = (size_t)(32 * nb * (DH + 4) + 32 * strips) * sizeof(float) + 80 * sizeof(int);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_ds_causal_kernel<DH>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    AS_REQUIRE(attr == hipSuccess, (int)attr, "as_attention_bwd_ds: cannot reserve LDS: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL((attn_ds_causal_kernel<DH>), dim3((unsigned)Z), dim3(ATT_THREADS), shm, st, k);
    return 0;
}

template <int DH>
int launch_attn_ds_nb(const AttnDsK& k, long Z, hipStream_t st) {
    const int nb = (k.Tk + 31) / 32;
    if (nb <= 1) return launch_attn_ds<DH, 1>(k, Z, st);
    if (nb <= 2) return launch_attn_ds<DH, 2>(k, Z, st);
    if (nb <= 4) return launch_attn_ds<DH, 4>(k, Z, st);
    if (nb <= 6) return launch_attn_ds<DH, 6>(k, Z, st);
    if (nb <= 7) return launch_attn_ds<DH, 7>(k, Z, st);
    return launch_attn_ds<DH, 8>(k, Z, st);
}

// softmax backward on key-major tensors, two launches:
//  (1) D[z][q] = sum_c dctx[q][c] ctx[q][c] over the head's columns: a quarter-wave per query row (16 lanes x float4 = 64
//      floats), consecutive quarter-waves on consecutive heads of the same row -> a wave reads one contiguous 1 KB row;
//  (2) the [Z][Tk][T] tensors are walked as ONE flat array in float4s (z, q from the index): full aligned 16-byte
//      accesses per lane whatever T is, no per-slice barrier.  (A thread per query walking 800-byte rows measured 2.07 ms,
//      one workgroup per slice with D in LDS 2.25 ms, against 1.3 ms for the row-major kernel this replaces.)
__global__ __launch_bounds__(256) void attn_dsum_kernel(const float* __restrict__ ctx, const float* __restrict__ dctx, long rows,
                                                        int heads, int T, int B, int d, float* __restrict__ D) {
    // item = (row r = (g*B + b)*T + q, head h); D index = ((g*B + b)*heads + h)*T + q
    const int dh = d / heads;
    const int per = dh / 4;                      // lanes per item (float4 each): 4, 8 or 16
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const long item = gt / per;
    const int part = (int)(gt - item * per);
    const bool ok = item < rows * heads;
    const long r = ok ? item / heads : 0;
    const int h = ok ? (int)(item - r * heads) : 0;
    const float4 a = *reinterpret_cast<const float4*>(ctx + r * d + (long)h * dh + part * 4);
    const float4 g = *reinterpret_cast<const float4*>(dctx + r * d + (long)h * dh + part * 4);
    float s = a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
    for (int o = per / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (ok && part == 0) {
        const long gb = r / T;
        const int q = (int)(r - gb * T);
        D[(gb * heads + h) * T + q] = s;
    }
}

__global__ __launch_bounds__(256) void attn_softmax_bwd_t_kernel(const float* __restrict__ pt, float* __restrict__ dpt,
                                                                 const float* __restrict__ D, long total4, int T, int Tp, long plane,
                                                                 float scale) {
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < total4; i0 += 4 * stride) {  // four float4 pairs in flight
        float4 pv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * stride < total4 ? i0 + u * stride : total4 - 1;
            pv[u] = reinterpret_cast<const float4*>(pt)[i];
            dv[u] = reinterpret_cast<const float4*>(dpt)[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * stride;
            if (i < total4) {
                const long e = 4 * i;                     // T % 4 == 0: the float4 is q .. q + 3 of one key row (row pitch Tp)
                const long z = e / plane;
                const int q = (int)((e - z * plane) % Tp);
                if (q >= T) continue;                     // the row's padding up to the pitch
                const float4 dq = *reinterpret_cast<const float4*>(D + z * T + q);
                float4 r;
                r.x = pv[u].x * (dv[u].x - dq.x) * scale;
                r.y = pv[u].y * (dv[u].y - dq.y) * scale;
                r.z = pv[u].z * (dv[u].z - dq.z) * scale;
                r.w = pv[u].w * (dv[u].w - dq.w) * scale;
                reinterpret_cast<float4*>(dpt)[i] = r;
            }
        }
    }
}

__global__ __launch_bounds__(256) void attn_softmax_bwd_t_scalar_kernel(const float* __restrict__ pt, float* __restrict__ dpt,
                                                                        const float* __restrict__ D, long total, int T, int Tp, long plane,
                                                                        float scale) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const long z = e / plane;
        const int q = (int)((e - z * plane) % Tp);
        if (q < T) dpt[e] = pt[e] * (dpt[e] - D[z * T + q]) * scale;
    }
}

template <int DH, int NB>
int launch_attn(const AttnK& k, long Z, hipStream_t st) {
    constexpr size_t shm = (size_t)(2 * 32 * NB * (DH + 4) + 32 * NB) * sizeof(float);
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<DH, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    AS_REQUIRE(attr == hipSuccess, (int)attr, "as_attention_fwd: cannot reserve %zu bytes of LDS: %s", shm, hipGetErrorString(attr));
    hipLaunchKernelGGL((attn_fwd_kernel<DH, NB>), dim3((unsigned)Z), dim3(ATT_THREADS), shm, st, k);
    return 0;
}

template <int DH>
int launch_attn_nb(const AttnK& k, long Z, hipStream_t st) {
    const int nb = (k.Tk + 31) / 32;
    if (nb <= 1) return launch_attn<DH, 1>(k, Z, st);
    if (nb <= 2) return launch_attn<DH, 2>(k, Z, st);
    if (nb <= 4) return launch_attn<DH, 4>(k, Z, st);
    if (nb <= 6) return launch_attn<DH, 6>(k, Z, st);
    if (nb <= 7) return launch_attn<DH, 7>(k, Z, st);
    return launch_attn<DH, 8>(k, Z, st);
}

}  // namespace

extern "C" int as_attention_supported(int32_t T, int32_t Tk, int32_t d, int32_t heads) {
    if (T <= 0 || Tk <= 0 || heads <= 0 || d <= 0 || d % heads != 0) return 0;
    const int dh = d / heads;
    return (dh == 16 || dh == 32 || dh == 64) && Tk <= 256;
}

static int attention_fwd(const float* Q, const float* K, const float* V, const float* attn_mask_t, const float* key_padding_mask,
                         float* out, float* lse, float* probs_t, int32_t G, int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d,
                         float scale, int causal, void* stream) {
    AS_REQUIRE(Q && K && V && out && G > 0 && B > 0 && heads > 0 && T > 0 && Tk > 0 && d > 0, AS_ERR_BAD_ARG, "as_attention_fwd: bad argument");
    AS_REQUIRE(as_attention_supported(T, Tk, d, heads), AS_ERR_UNSUPPORTED,
               "as_attention_fwd: head width %d / %d not in {16, 32, 64} or Tk=%d > 256 (use the unfused path)", d, heads, Tk);
    AS_REQUIRE(((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) |
                 reinterpret_cast<uintptr_t>(out)) & 15) == 0, AS_ERR_BAD_ARG, "as_attention_fwd: operands must be 16-byte aligned");
    AttnK k;
    k.Q = Q; k.K = K; k.V = V; k.O = out; k.mask_t = attn_mask_t; k.kpm = key_padding_mask; k.lse = lse; k.probs_t = probs_t;
    k.B = B; k.heads = heads; k.T = T; k.Tk = Tk; k.d = d; k.scale = scale;
    k.tp = (T + 31) / 32 * 32;
    k.causal = causal && attn_mask_t != nullptr;
    const long Z = (long)G * B * heads;
    hipStream_t st = (hipStream_t)stream;
    const int dh = d / heads;
    if (dh == 64) AS_TRY(launch_attn_nb<64>(k, Z, st));
    else if (dh == 32) AS_TRY(launch_attn_nb<32>(k, Z, st));
    else AS_TRY(launch_attn_nb<16>(k, Z, st));
    AS_LAUNCH_CHECK("as_attention_fwd");
    return 0;
}

extern "C" int as_attention_fwd(const float* Q, const float* K, const float* V, const float* attn_mask_t, const float* key_padding_mask,
                                float* out, float* lse, float* probs_t, int32_t G, int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d,
                                float scale, void* stream) {
    return attention_fwd(Q, K, V, attn_mask_t, key_padding_mask, out, lse, probs_t, G, B, heads, T, Tk, d, scale, 0, stream);
}

extern "C" int as_attention_fwd_causal(const float* Q, const float* K, const float* V, const float* attn_mask_t,
                                       const float* key_padding_mask, float* out, float* lse, float* probs_t, int32_t G, int32_t B,
                                       int32_t heads, int32_t T, int32_t Tk, int32_t d, float scale, void* stream) {
    return attention_fwd(Q, K, V, attn_mask_t, key_padding_mask, out, lse, probs_t, G, B, heads, T, Tk, d, scale, 1, stream);
}

extern "C" int as_attn_softmax_bwd_t(const float* probs_t, float* dprobs_t, const float* ctx, const float* dctx, float* dsum, int32_t G,
                                     int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d, float scale, void* stream) {
    AS_REQUIRE(probs_t && dprobs_t && ctx && dctx && dsum && G > 0 && B > 0 && heads > 0 && T > 0 && Tk > 0 && d > 0 && d % heads == 0,
               AS_ERR_BAD_ARG, "as_attn_softmax_bwd_t: bad argument");
    const int dh = d / heads;
    AS_REQUIRE(dh == 16 || dh == 32 || dh == 64, AS_ERR_UNSUPPORTED, "as_attn_softmax_bwd_t: head width %d not in {16, 32, 64}", dh);
    AS_REQUIRE(((reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(dctx)) & 15) == 0, AS_ERR_BAD_ARG,
               "as_attn_softmax_bwd_t: ctx / dctx must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)G * B * T, Z = (long)G * B * heads;
    const long lanes = rows * heads * (dh / 4);
    hipLaunchKernelGGL(attn_dsum_kernel, dim3((unsigned)as_cdiv(lanes, 256)), dim3(256), 0, st, ctx, dctx, rows, heads, T, B, d, dsum);
    AS_LAUNCH_CHECK("as_attn_softmax_bwd_t(dsum)");
    const int Tp = (T + 31) / 32 * 32;   // row pitch of the key-major tensors (as_attention_fwd)
    const long plane = (long)Tk * Tp, total = Z * plane;
    const bool vec = T % 4 == 0 && ((reinterpret_cast<uintptr_t>(probs_t) | reinterpret_cast<uintptr_t>(dprobs_t) |
                                     reinterpret_cast<uintptr_t>(dsum)) & 15) == 0;
    if (vec) {
        long blocks = as_cdiv(total / 4, 4 * 256);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(attn_softmax_bwd_t_kernel, dim3((unsigned)blocks), dim3(256), 0, st, probs_t, dprobs_t, dsum, total / 4, T, Tp, plane, scale);
    } else {
        long blocks = as_cdiv(total, 256);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(attn_softmax_bwd_t_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, st, probs_t, dprobs_t, dsum, total, T, Tp, plane, scale);
    }
    AS_LAUNCH_CHECK("as_attn_softmax_bwd_t");
    return 0;
}

static int attention_bwd_ds(const float* V, const float* dctx, const float* ctx, const float* probs_t, float* ds_t, int32_t G,
                            int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d, float scale, int causal, void* stream) {
    AS_REQUIRE(V && dctx && ctx && probs_t && ds_t && G > 0 && B > 0 && heads > 0 && T > 0 && Tk > 0 && d > 0, AS_ERR_BAD_ARG,
               "as_attention_bwd_ds: bad argument");
    AS_REQUIRE(as_attention_supported(T, Tk, d, heads), AS_ERR_UNSUPPORTED,
               "as_attention_bwd_ds: head width %d / %d not in {16, 32, 64} or Tk=%d > 256", d, heads, Tk);
    AS_REQUIRE(((reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(dctx) | reinterpret_cast<uintptr_t>(ctx)) & 15) == 0,
               AS_ERR_BAD_ARG, "as_attention_bwd_ds: operands must be 16-byte aligned");
    AttnDsK k;
    k.V = V; k.dctx = dctx; k.ctx = ctx; k.probs_t = probs_t; k.ds_t = ds_t;
    k.B = B; k.heads = heads; k.T = T; k.Tk = Tk; k.d = d; k.scale = scale;
    k.tp = (T + 31) / 32 * 32;
    const long Z = (long)G * B * heads;
    hipStream_t st = (hipStream_t)stream;
    const int dh = d / heads;
    if (causal && T <= 256 && Tk <= 256) {
        if (dh == 64) AS_TRY(launch_attn_ds_causal<64>(k, Z, st));
        else if (dh == 32) AS_TRY(launch_attn_ds_causal<32>(k, Z, st));
        else AS_TRY(launch_attn_ds_causal<16>(k, Z, st));
    } else if (dh == 64) AS_TRY(launch_attn_ds_nb<64>(k, Z, st));
    else if (dh == 32) AS_TRY(launch_attn_ds_nb<32>(k, Z, st));
    else AS_TRY(launch_attn_ds_nb<16>(k, Z, st));
    AS_LAUNCH_CHECK("as_attention_bwd_ds");
    return 0;
}

extern "C" int as_attention_bwd_ds(const float* V, const float* dctx, const float* ctx, const float* probs_t, float* ds_t, int32_t G,
                                   int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d, float scale, void* stream) {
    return attention_bwd_ds(V, dctx, ctx, probs_t, ds_t, G, B, heads, T, Tk, d, scale, 0, stream);
}

extern "C" int as_attention_bwd_ds_causal(const float* V, const float* dctx, const float* ctx, const float* probs_t, float* ds_t,
                                          int32_t G, int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d, float scale,
                                          void* stream) {
    return attention_bwd_ds(V, dctx, ctx, probs_t, ds_t, G, B, heads, T, Tk, d, scale, 1, stream);
}
