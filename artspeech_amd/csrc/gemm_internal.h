// Internal (non-ABI) hand-over between the GEMM translation units.
#pragma once
#include "as_common.h"

// wgrad_f32.hip: weight-gradient shapes (both operands reduction-strided, long reduction).  1 = taken and launched,
// 0 = not a shape for this kernel (the caller continues with the general kernel), < 0 = error.
int as_wgrad_try(const as_gemm* g, hipStream_t st);
// Several weight-gradient problems of the same reduction length (a_i == b_j == 1, linear batch strides, N > 128) as ONE
// launch of 128 x 256 tiles + one reduce launch.  g.splitk_ws / cu_budget of the jobs are ignored (slab, cu_budget here).
// colsum_b (optional): column sums of the B operand [batch][N] (the bias gradient when the problem is posed transposed);
// c_trans: the result is stored transposed, C[batch][n * ldc + m].  1 = launched, 0 = not a case (caller falls back), < 0 = error.
struct as_wgrad_job {
    as_gemm g;
    float* colsum_b; long colsum_b_batch;
    int c_trans;
};
// exact: the fp32 matrix instruction whatever as_matrix_arith() says -- for launches that run BESIDE a latency-bound kernel: the
// bf16 instruction draws more power, the chip clocks ~10 % lower under it, and a recurrence on the other CUs pays that on
// every dependent step (measured in the BiGRU step: weight gradients 161 -> 114 us, the recurrence beside them 115 -> 131 us).
int as_wgrad_multi(const as_wgrad_job* jobs, int n, float* slab, long slab_floats, int cu_budget, hipStream_t st, bool exact = false);

// How fp32 matrix products are formed (as_set_matrix_arith, include/artspeech_hip.h):
//   AS_ARITH_FP32    v_mfma_f32_32x32x2_f32 on the fp32 operands (every kernel has this path)
//   AS_ARITH_BF16X6  operands split exactly into three bfloat16 planes, six plane products on v_mfma_f32_32x32x16_bf16,
//                    fp32 accumulation -- where a kernel has the path (the default)
enum { AS_ARITH_FP32 = 0, AS_ARITH_BF16X6 = 1 };
int as_matrix_arith();

// rowops.hip: B[batch][n][k] (element strides n_stride, k_stride, batch_stride) as three bfloat16 planes
// out[plane][batch][Kpad / 16][rows_pad][16] (x = hi + mid + lo exactly; n >= N or k >= K: zeros).  Up to 8 jobs, one launch.
struct as_planes_job {
    const float* B; long n_stride, k_stride, batch_stride;
    int batch, N, K, rows_pad, Kpad;
    uint16_t* out;      // plane stride = batch * (Kpad / 16) * rows_pad * 16 elements, batch stride = (Kpad / 16) * rows_pad * 16
};
int as_emit_planes(const as_planes_job* jobs, int n, hipStream_t st);
static inline long as_planes_batch_stride(int rows_pad, int Kpad) { return (long)(Kpad / 16) * rows_pad * 16; }
static inline long as_planes_floats(int batch, int rows_pad, int Kpad) { return 3 * batch * as_planes_batch_stride(rows_pad, Kpad) / 2; }

// gemm_s6.hip: C[g] = act(A[g][M][K] . B[g][N][K]^T + bias[g]) with both operands fp32 and reduction-contiguous, in the split
// arithmetic (both split inside the kernel: no plane copies).  K % 16 == 0, lda / ldb / batch strides % 4 == 0, 16-byte aligned
// operands.  1 = launched, 0 = not a case (mode fp32, shape, alignment: take as_gemm_f32), < 0 = error.
int as_gemm_s6_nt(const float* A, long lda, long a_batch, const float* B, long ldb, long b_batch, const float* bias, long bias_batch, float* C,
                  long ldc, long c_batch, int M, int N, int K, int batch, int act, hipStream_t st);
// the same from an as_gemm descriptor: forward shapes (a_k == b_k == 1) incl. grouped offsets (a_off ...) and relu_bits; everything
// else the descriptor may ask for (res, mask_bits, k_seg, k_tri, colsum, split K, accumulate, shifts) -> 0
int as_gemm_s6_nt_ext(const as_gemm* g, hipStream_t st);
template <int N> struct IC2 { static constexpr int value = N; };

// A cross-stream fork without a barrier packet on the producing stream: the event is bound to the completion of the NEXT kernel
// this thread launches through a launch site that knows the mechanism (hipExtLaunchKernelGGL's stopEvent: the dispatch packet's
// own completion signal), instead of hipEventRecord's marker packet behind it (~7 us on the stream that records it, measured on
// the step timeline).  as_stop_event_take(): the event if no launch has consumed it (the caller then records it the ordinary way).
void as_stop_event_set_raw(void* ev);
void* as_stop_event_take_raw();
inline void as_stop_event_set(hipEvent_t ev) { as_stop_event_set_raw((void*)ev); }
inline hipEvent_t as_stop_event_take() { return (hipEvent_t)as_stop_event_take_raw(); }

// lin_f32.hip: one Linear of the ArticulatorPredictor heads with the adjoining LayerNorm fused in (batched over heads):
//   C[bz] = epilogue(A[bz] [M][K] . B[bz]),  B[bz] = [N][K] (b_kc: forward) or [K][N] (backward); N <= 256, K % 32 == 0.
//   epi 0: act(. + bias) (act as as_gemm: 0 none, 1 ReLU, 2 sigmoid)
//   epi 1: x = relu(. + bias); C = (x - mean) * rstd over the 256 features; rstd [M][batch]; bits [M][batch][4] = x > 0
//   epi 2: g = .; C = relu'(bits_in) * rstd_in * (g - mean g - xhat * mean(g xhat))   (LayerNorm + ReLU backward)
// ka_valid (0 = K): k >= ka_valid of A is not read (the caller's B has zero rows there).  Strides in floats.
// 1 = launched, 0 = not a case for this kernel (take the general GEMM + row kernels), < 0 = error.
struct as_lin {
    const float* A; long lda, a_batch;
    const float* B; long ldb, b_batch; int b_kc;
    // optional: the same B as three bfloat16 planes [plane][batch][K / 16][bp_rows][16] (as_emit_planes; rows >= N zero,
    // strides in bf16 elements).  With as_matrix_arith() == AS_ARITH_BF16X6 and K % 32 == 0 the layer then runs on the bf16
    // matrix instruction (six plane products per fp32 product, fp32 accumulate: lin_s6_kernel); B itself is not read.
    const uint16_t* Bp; long bp_plane, bp_batch; int bp_rows;
    float* C; long ldc, c_batch;
    const float* bias; long bias_batch;
    int M, N, K, ka_valid, batch, act, epi;
    float* rstd; unsigned long long* bits;
    const float* xhat; long ldx, x_batch; const float* rstd_in; const unsigned long long* bits_in;
    int tile_rows;   // 0 = the mixed tile list (64-row tiles that fill whole rounds, then 32-row tiles); 64 / 32 = that size only
};
int as_lin_try(const as_lin* a, hipStream_t st);
// lin_f32.hip: epi 0 on the bf16 matrix instruction (a->Bp required; a->B unused): C = act(A . B^T + bias), N <= 256, K % 32 == 0.
// ksplit > 1: the reduction is cut into that many chunks, chunk y writes its partial sums (no bias / activation allowed) to
// C + y * c_split floats; the CONSUMER adds the slabs.  1 = launched, 0 = not a case (mode fp32, shapes), < 0 = error.
int as_lin_plain_s6(const as_lin* a, int ksplit, long c_split, hipStream_t st);
// the number of k-chunks (slabs) as_lin_plain_s6 really uses for a requested ksplit
static inline int as_lin_plain_s6_slabs(int K, int ksplit) {
    if (ksplit < 1) ksplit = 1;
    const int kchunk = (int)as_round_up(as_cdiv(K, ksplit), 32);
    return as_cdiv(K, kchunk);
}

// lin_f32.hip: the heads' output layer out[M][batch][N] = sigmoid(A[M][batch][K] . B[batch][>= 128 rows][K]^T + bias), N <= 128,
// optionally with the masked Euclidean criterion fused (tgt != NULL): loss partials (one per workgroup, `partial`, at most
// partial_capacity) and d loss / d(pre-sigmoid) in `dout` (layout of out).  1 = launched, 0 = not a case, < 0 = error.
struct as_lin_out {
    const float* A; long lda, a_batch;
    const float* B; long ldb, b_batch; int b_rows;
    const float* bias; long bias_batch;
    float* out; long ldo, o_batch;
    int M, N, K, batch;
    const float* tgt; long tgt_T; const int* lengths; int T; float scale;
    float* dout; float* partial; long partial_capacity;
    const uint16_t* Bp; long bp_plane, bp_batch; int bp_rows;   // optional: B as bfloat16 planes, bp_rows >= 128 (see as_lin.Bp)
    // optional with tgt: the device scalar that receives scale * sum of the partials.  The workgroup that delivers the LAST partial
    // adds them all itself, in the order of as_loss_final (same value): no second launch on the critical stream.  Needs an arrival
    // counter (as_arrival_counter); without one the partials are left for as_loss_final as before (*n_partials > 0).
    float* loss;
};
// 1 = launched (fused criterion: *n_partials workgroup sums were written to `partial` and await as_loss_final; 0 of them when the
// kernel has summed them into a->loss itself), 0 = not a case, < 0 = error
int as_lin_out_try(const as_lin_out* a, int* n_partials, hipStream_t st);
// gemm_f32.hip: one of the stream's arrival counters (a device word that every launch leaves zero), or nullptr
int* as_arrival_counter(hipStream_t st);
// metrics.hip: *loss = scale * sum of the first n partials (fixed order)
int as_loss_final(const float* partial, int n, float scale, float* loss, hipStream_t st);

// gru.hip: backward recurrence of layer 0 under a token table; see the kernel.  1 = launched, 0 = not a case, < 0 = error.
bool as_gru_bwd_tokens_fits(int32_t V, int32_t H, int32_t T);   // the [V][3H] table + T offsets fit the kernel's LDS budget
int as_gru_bidir_bwd_tokens(const float* dy, const float* y, const float* gates, const float* w_hh, const int32_t* lengths,
                            int32_t B, int32_t T, int32_t H, float* dgh, const int64_t* tokens, int64_t tok_stride, int32_t V,
                            float* part, hipStream_t st);
// gru.hip: as_gru_bidir_fwd over a token table [V][2][3H] with the ids clamped into [0, V) (memory safety; the composite
// entry point counts out-of-range ids for the host)
int as_gru_bidir_fwd_tokens(const float* table, const int64_t* tokens, int64_t tok_stride, int32_t V, const float* w_hh,
                            const float* b_hh, const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* y, float* gates,
                            hipStream_t st);
// rowops.hip: *count = number of ids outside [0, V) among tokens[b][t], b < rows / T, t < T (one small workgroup)
int as_count_bad_tokens(const int64_t* tokens, long tok_stride, int T, long rows, int V, int* count, hipStream_t st);
// rowops.hip: out[i] = sum over chunks (fixed order) of part[chunk][i], i < n
int as_sum_partials(const float* part, long n, int chunks, float* out, hipStream_t st);
