// Composite entry points: the flat parameter layout, the workspace carve and the launch sequences of
// ArtSpeech / SimpleArtSpeech forward + backward and of the stacked ArticulatorPredictor heads.
// This is the native "runtime" of the path: Python hands over pointers once per step; every kernel of
// the step is enqueued from here on one HIP stream (graph-capturable: no allocation, no sync).
//
// Design notes (reference lines: encoder_decoder/models.py):
//  * embedding (:135) + layer-0 input projection are folded into a [V][2][3H] row table (one tiny
//    GEMM per step); the GRU kernel gathers table rows by token id, so neither the embedded sequence
//    nor its projection is ever materialised.  Backward: per-token segment sum -> two tiny GEMMs.
//  * every LayerNorm of the heads (:11,14,17) is applied affine-free (x_hat) and its gamma/beta are
//    folded into the following Linear (W' = W.diag(gamma), b' = b + W.beta).  The first LayerNorm's
//    x_hat is then shared by all A heads (same input row), and head GEMM 1 becomes ONE GEMM with
//    N = A*256 columns.  The backward unfolds dW', db' into dW, dgamma, dbeta, db exactly.
//  * heads are batched over A with strided-batched GEMMs on [rows][A][256] activations; the last GEMM
//    writes sigmoid(.) straight into the (B, T, A, 2, N) output (:141-145).
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "gemm_internal.h"
#include "rowops.h"

namespace {

constexpr int D = AS_HEAD_HIDDEN;
constexpr int64_t SLAB_FLOATS = 16LL << 20;  // 64 MB of split-K partial tiles per stream
constexpr int64_t SLAB2_FLOATS = 28LL << 20; // first side stream: the partial tiles of ALL head weight gradients in one launch

struct Carve {
    int64_t off = 0;
    int64_t take(int64_t n) {
        const int64_t o = off;
        off += as_round_up(n, 64);
        return o;
    }
};

struct HeadWs {  // offsets in floats relative to the head workspace base
    int64_t xhat, rstd0, w1f, b1f, w2f, b2f, w3f, b3f, r1, r1hat, rstd1, r2, r2hat, rstd2;
    int64_t dpre3, dz2, dz1, dxhat, dw1f, dw2f, dw3f, slab, slab2, slab3, bits1, bits2, lpart, lpart_n, total;
    // the folded weights as bfloat16 planes (as_emit_planes): forward orientation [rows = out features][k = in features] and,
    // for the input-gradient chain, transposed [rows = in features][k = out features]
    int64_t w1p, w2p, w3p, w2tp, w3tp, w1tp;
};
constexpr int OUT_BN = 128;   // column tile of the output layer's kernel (lin_out): rows of the W3' planes per head

HeadWs head_ws(const as_dims& d, int64_t rows) {
    const int64_t A = d.n_art, H = d.hidden, O = 2 * d.n_samp;
    Carve c;
    HeadWs w;
    w.xhat = c.take(rows * H);
    w.rstd0 = c.take(rows);
    w.w1f = c.take(A * D * H);
    w.b1f = c.take(A * D);
    w.w2f = c.take(A * D * D);
    w.b2f = c.take(A * D);
    w.w3f = c.take(A * as_round_up(O, 32) * D);  // rows O .. Opad-1 of every head are zeros (head_fold)
    w.b3f = c.take(A * O);
    w.r1hat = c.take(rows * A * D);
    w.rstd1 = c.take(rows * A);
    w.r2hat = c.take(rows * A * D);
    w.rstd2 = c.take(rows * A);
    w.dpre3 = c.take(rows * A * O);
    w.dz2 = c.take(rows * A * D);
    w.dz1 = c.take(rows * A * D);
    // pre-normalisation activations of the UNFUSED forward path only (as_lin_try declined): they live where the backward's
    // dz1 / dz2 will go, which nothing touches before the backward (the fused path never materialises them)
    w.r1 = w.dz1;
    w.r2 = w.dz2;
    w.dxhat = c.take(rows * H);
    w.dw1f = c.take(A * D * H);
    w.dw2f = c.take(A * D * D);
    w.dw3f = c.take(A * O * D);
    w.slab = c.take(SLAB_FLOATS);   // split-K partial tiles of the weight-gradient GEMMs (main stream)
    w.slab2 = c.take(SLAB2_FLOATS); // same, for GEMMs issued on the side stream
    w.slab3 = c.take(SLAB_FLOATS);  // same, second side stream
    w.bits1 = c.take(rows * A * (D / 64) * 2);  // ReLU masks of r1 / r2: one bit per element (64-bit words, 16-byte aligned)
    w.bits2 = c.take(rows * A * (D / 64) * 2);
    // workgroup partial sums of the fused criterion (one per output-layer tile), or of the separate criterion kernel
    w.lpart_n = std::max<int64_t>((rows / 32 + 2) * A, as_euclid_masked_partials());
    w.lpart = c.take(w.lpart_n);
    const int Opad = (int)as_round_up(O, 32);
    w.w1p = c.take(as_planes_floats((int)A, D, (int)as_round_up(H, 32)));
    w.w2p = c.take(as_planes_floats((int)A, D, D));
    w.w3p = c.take(as_planes_floats((int)A, (int)as_round_up(O, OUT_BN), D));
    w.w2tp = c.take(as_planes_floats((int)A, D, D));
    w.w3tp = c.take(as_planes_floats((int)A, D, Opad));
    w.w1tp = c.take(as_planes_floats(1, (int)as_round_up(H, 128), (int)(A * D)));   // d(x_hat) = dz1 . W1': rows = the H inputs, k = (head, feature)
    w.total = c.off;
    return w;
}

struct ModelWs {
    int64_t tokflag, tab0, y0, y0d, g0, xp1, y1, g1, lin, head, dy1, dgi1, dgh1, dy0, dgi0, dgh0, dtab0, total;
};

ModelWs model_ws(const as_dims& d, int64_t B, int64_t T) {
    const int64_t R = B * T, H = d.hidden, V = d.vocab;
    Carve c;
    ModelWs w{};
    w.tokflag = c.take(64);   // first word (int32): ids outside [0, V) seen by the last as_artspeech_fwd on this workspace
    if (!d.simple) {
        w.tab0 = c.take(V * 6 * H);
        w.y0 = c.take(R * 2 * H);
        w.y0d = c.take(R * 2 * H);  // layer-0 output after inter-layer dropout (training with p > 0)
        w.g0 = c.take(R * 8 * H);
        w.xp1 = c.take(R * 6 * H);
        w.y1 = c.take(R * 2 * H);
        w.g1 = c.take(R * 8 * H);
        w.dy1 = c.take(R * 2 * H);
        w.dgi1 = c.take(R * 6 * H);
        w.dgh1 = c.take(R * 6 * H);
        w.dy0 = c.take(R * 2 * H);
        w.dgi0 = c.take(R * 6 * H);
        w.dgh0 = c.take(R * 6 * H);
        w.dtab0 = c.take(V * 6 * H);
    } else {
        w.tab0 = c.take(V * H);   // relu(Emb Wl^T + bl) per token
        w.dtab0 = c.take(V * H);
        w.y0 = c.take(R * d.embed);   // training with dropout p > 0: the embedded frames after dropout ...
        w.dy0 = c.take(R * d.embed);  // ... and the gradient that reaches them
    }
    w.lin = c.take(R * H);
    w.head = c.take(head_ws(d, R).total);
    w.total = c.off;
    return w;
}

int check_dims(const as_dims* d, const char* who) {
    AS_REQUIRE(d, AS_ERR_BAD_ARG, "%s: null dims", who);
    AS_REQUIRE(d->vocab > 0 && d->n_art > 0 && d->embed > 0 && d->hidden > 0 && d->n_samp > 0, AS_ERR_BAD_ARG,
               "%s: non-positive dimension (V=%d A=%d E=%d H=%d N=%d)", who, d->vocab, d->n_art, d->embed, d->hidden, d->n_samp);
    AS_REQUIRE(d->hidden <= 512, AS_ERR_UNSUPPORTED, "%s: hidden size %d > 512", who, d->hidden);
    // hidden sizes 32 / 64 / 128 run the register-resident recurrence kernels, any other multiple of 4 (16-byte rows) the plain
    // ones of gru.hip (several times slower per step)
    if (!d->simple)
        AS_REQUIRE(d->hidden % 4 == 0, AS_ERR_UNSUPPORTED, "%s: GRU hidden size %d is not a multiple of 4", who, d->hidden);
    return 0;
}

// C = act(A . B^T + bias): both operands reduction-contiguous
int gemm_nt(const float* A, long lda, const float* Bw, long ldb, float* C, long ldc, const float* bias, int M, int N, int K,
            int act, hipStream_t st, int batch = 1, long ab = 0, long bb = 0, long cb = 0, long biasb = 0) {
    as_gemm g{};
    g.A = A; g.B = Bw; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.a_i = lda; g.a_k = 1; g.b_j = ldb; g.b_k = 1; g.ldc = ldc;
    g.batch = batch; g.a_batch = ab; g.b_batch = bb; g.c_batch = cb; g.bias_batch = biasb; g.act = act;
    return as_gemm_f32(&g, st);
}
// C[M][N] = A[M][K] . B[K][N]   (input gradient: reduction over the rows of B)
int gemm_nn(const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int M, int N, int K, hipStream_t st,
            int batch = 1, long ab = 0, long bb = 0, long cb = 0, float* slab = nullptr) {
    as_gemm g{};
    g.A = A; g.B = Bm; g.C = C; g.M = M; g.N = N; g.K = K;
    g.a_i = lda; g.a_k = 1; g.b_j = 1; g.b_k = ldb; g.ldc = ldc;
    g.batch = batch; g.a_batch = ab; g.b_batch = bb; g.c_batch = cb;
    g.splitk_ws = slab; g.splitk_ws_floats = slab ? SLAB_FLOATS : 0;  // few output tiles + long reduction: split K
    return as_gemm_f32(&g, st);
}
// C[M][N] = A[K][M]^T . B[K][N]   (weight gradient: reduction over the rows of both)
// colsum (optional): column sums of A, i.e. the bias gradient that goes with this weight gradient, [batch][M]
// cu_budget: CUs the launch can count on (0 = all): 192 for the side stream's launches beside a 64-workgroup recurrence
int gemm_tn(const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, int M, int N, int K, hipStream_t st,
            float* slab, float* colsum = nullptr, long csb = 0, int batch = 1, long ab = 0, long bb = 0, long cb = 0,
            int kshift = 0, int kT = 0, int kshift_batch = 0, int cu_budget = 0) {
    as_gemm g{};
    g.A = A; g.B = Bm; g.C = C; g.M = M; g.N = N; g.K = K;
    g.a_i = 1; g.a_k = lda; g.b_j = 1; g.b_k = ldb; g.ldc = ldc;
    g.batch = batch; g.a_batch = ab; g.b_batch = bb; g.c_batch = cb; g.b_kshift = kshift; g.b_kT = kT; g.b_kshift_batch = kshift_batch;
    g.splitk_ws = slab; g.splitk_ws_floats = SLAB_FLOATS; g.colsum = colsum; g.colsum_batch = csb; g.cu_budget = cu_budget;
    return as_gemm_f32(&g, st);
}

// fold the LayerNorm affines into the head weights (depends on the parameters only: can run early / elsewhere)
int head_fold(const as_dims& d, const as_layout& L, const float* P, int64_t rows, float* ws, hipStream_t st) {
    const int A = d.n_art, H = d.hidden, O = 2 * d.n_samp;
    const HeadWs w = head_ws(d, rows);
    AS_STEP("head.fold", st, as_fold(P + L.w1, P + L.ln1_g, P + L.ln1_b, P + L.b1, ws + w.w1f, ws + w.b1f, A, D, H, st));
    AS_STEP("head.fold", st, as_fold(P + L.w2, P + L.ln2_g, P + L.ln2_b, P + L.b2, ws + w.w2f, ws + w.b2f, A, D, D, st));
    AS_STEP("head.fold", st, as_fold(P + L.w3, P + L.ln3_g, P + L.ln3_b, P + L.b3, ws + w.w3f, ws + w.b3f, A, O, D, st, (int)as_round_up(O, 32)));
    if (as_matrix_arith() == AS_ARITH_BF16X6 && H % 4 == 0) {
        // the folded weights as bfloat16 planes for the split-arithmetic kernels: forward orientation, and transposed for
        // the input-gradient chain (d(x_hat) = dz . W': the reduction runs over W' rows)
        const int Opad = (int)as_round_up(O, 32), Hp = (int)as_round_up(H, 32);
        auto up = [&](int64_t off) { return reinterpret_cast<uint16_t*>(ws + off); };
        as_planes_job j[6] = {
            {ws + w.w1f, H, 1, (long)D * H, A, D, H, D, Hp, up(w.w1p)},
            {ws + w.w2f, D, 1, (long)D * D, A, D, D, D, D, up(w.w2p)},
            {ws + w.w3f, D, 1, (long)Opad * D, A, O, D, (int)as_round_up(O, OUT_BN), D, up(w.w3p)},
            {ws + w.w2f, 1, D, (long)D * D, A, D, D, D, D, up(w.w2tp)},
            {ws + w.w3f, 1, D, (long)Opad * D, A, D, O, D, Opad, up(w.w3tp)},
            {ws + w.w1f, 1, H, 0, 1, H, A * D, (int)as_round_up(H, 128), A * D, up(w.w1tp)},
        };
        AS_STEP("head.planes", st, as_emit_planes(j, 6, st));
    }
    return 0;
}

// crit (optional): the training criterion fused into the output layer (as_opts.loss_*; lengths / T of the batch)
struct Criterion { const float* tgt; long tgt_T; const int* lengths; int T; float scale; float* loss; float* dout; };

int head_fwd_impl(const as_dims& d, const as_layout& L, const float* P, const float* x, int64_t rows, float* out, float* ws,
                  hipStream_t st, const Criterion* crit = nullptr) {
    (void)P;
    const int A = d.n_art, H = d.hidden, O = 2 * d.n_samp;
    const HeadWs w = head_ws(d, rows);
    const int R = (int)rows;
    AS_STEP("head.norm0", st, as_normalize_fwd(x, ws + w.xhat, ws + w.rstd0, rows, H, st));
    const long AD = (long)A * D;
    const int Opad = (int)as_round_up(O, 32);
    unsigned long long* bits1 = reinterpret_cast<unsigned long long*>(ws + w.bits1);
    unsigned long long* bits2 = reinterpret_cast<unsigned long long*>(ws + w.bits2);
    // Linear 1 + ReLU + LayerNorm 2 in one kernel per (96 frames, head): r1hat, rstd1, ReLU bits (lin_f32.hip); else the
    // general GEMM (all heads at once: shared x_hat, N = A*256 columns) followed by the row kernel
    as_lin l1{};
    l1.A = ws + w.xhat; l1.lda = H; l1.a_batch = 0;
    l1.B = ws + w.w1f; l1.ldb = H; l1.b_batch = (long)D * H; l1.b_kc = 1;
    l1.C = ws + w.r1hat; l1.ldc = AD; l1.c_batch = D;
    l1.bias = ws + w.b1f; l1.bias_batch = D;
    l1.M = R; l1.N = D; l1.K = H; l1.batch = A; l1.epi = 1;
    l1.rstd = ws + w.rstd1; l1.bits = bits1;
    const uint16_t* w1p = reinterpret_cast<const uint16_t*>(ws + w.w1p);
    const uint16_t* w2p = reinterpret_cast<const uint16_t*>(ws + w.w2p);
    if (H % 32 == 0) {
        l1.Bp = w1p; l1.bp_rows = D; l1.bp_batch = as_planes_batch_stride(D, H); l1.bp_plane = A * l1.bp_batch;
    }
    int took;
    bool crit_done = false;
    {
        AS_PROF("head.gemm1", st);
        took = as_lin_try(&l1, st);
        AS_REQUIRE(took >= 0, took, "head gemm1: launch failed");
    }
    if (!took) {
        AS_STEP("head.gemm1", st, gemm_nt(ws + w.xhat, H, ws + w.w1f, H, ws + w.r1, AD, ws + w.b1f, R, A * D, H, 1, st));
        AS_STEP("head.norm1", st, as_normalize_fwd(ws + w.r1, ws + w.r1hat, ws + w.rstd1, rows * A, D, st, bits1));
    }
    // Linear 2 + ReLU + LayerNorm 3, batched over heads on [rows][A][D]
    as_lin l2 = l1;
    l2.A = ws + w.r1hat; l2.lda = AD; l2.a_batch = D;
    l2.B = ws + w.w2f; l2.ldb = D; l2.b_batch = (long)D * D;
    l2.C = ws + w.r2hat; l2.bias = ws + w.b2f;
    l2.K = D; l2.rstd = ws + w.rstd2; l2.bits = bits2;
    l2.Bp = w2p; l2.bp_rows = D; l2.bp_batch = as_planes_batch_stride(D, D); l2.bp_plane = A * l2.bp_batch;
    {
        AS_PROF("head.gemm2", st);
        took = as_lin_try(&l2, st);
        AS_REQUIRE(took >= 0, took, "head gemm2: launch failed");
    }
    if (!took) {
        AS_STEP("head.gemm2", st, gemm_nt(ws + w.r1hat, AD, ws + w.w2f, D, ws + w.r2, AD, ws + w.b2f, R, D, D, 1, st, A, D,
                       (long)D * D, D, D));
        AS_STEP("head.norm2", st, as_normalize_fwd(ws + w.r2, ws + w.r2hat, ws + w.rstd2, rows * A, D, st, bits2));
    }
    // Linear 3 (x_coords | y_coords) with the sigmoid epilogue writes out[rows][A][2][N]
    as_lin l3{};
    l3.A = ws + w.r2hat; l3.lda = AD; l3.a_batch = D;
    l3.B = ws + w.w3f; l3.ldb = D; l3.b_batch = (long)Opad * D; l3.b_kc = 1;
    l3.C = out; l3.ldc = (long)A * O; l3.c_batch = O;
    l3.bias = ws + w.b3f; l3.bias_batch = O;
    l3.M = R; l3.N = O; l3.K = D; l3.batch = A; l3.act = 2; l3.epi = 0;
    {
        AS_PROF("head.gemm3", st);
        // 64 x 128-column tiles of one head each, four workgroups per CU (lin_out_kernel); with `crit` the masked Euclidean
        // criterion and its gradient ride in the epilogue
        as_lin_out lo{};
        lo.A = ws + w.r2hat; lo.lda = AD; lo.a_batch = D;
        lo.B = ws + w.w3f; lo.ldb = D; lo.b_batch = (long)Opad * D; lo.b_rows = Opad;
        lo.bias = ws + w.b3f; lo.bias_batch = O;
        lo.out = out; lo.ldo = (long)A * O; lo.o_batch = O;
        lo.M = R; lo.N = O; lo.K = D; lo.batch = A;
        lo.Bp = reinterpret_cast<const uint16_t*>(ws + w.w3p); lo.bp_rows = (int)as_round_up(O, OUT_BN);
        lo.bp_batch = as_planes_batch_stride(lo.bp_rows, D); lo.bp_plane = A * lo.bp_batch;
        int n_part = 0;
        if (crit) {
            lo.tgt = crit->tgt; lo.tgt_T = crit->tgt_T; lo.lengths = crit->lengths; lo.T = crit->T; lo.scale = crit->scale;
            lo.dout = crit->dout; lo.partial = ws + w.lpart; lo.partial_capacity = w.lpart_n; lo.loss = crit->loss;
        }
        took = as_lin_out_try(&lo, &n_part, st);
        AS_REQUIRE(took >= 0, took, "head gemm3: launch failed");
        if (took && crit && n_part > 0) AS_TRY(as_loss_final(ws + w.lpart, n_part, crit->scale, crit->loss, st));   // (0: summed in the kernel)
        crit_done = took && crit;
        if (!took) {
            took = as_lin_try(&l3, st);
            AS_REQUIRE(took >= 0, took, "head gemm3: launch failed");
        }
    }
    if (!took)
        AS_STEP("head.gemm3", st, gemm_nt(ws + w.r2hat, AD, ws + w.w3f, D, out, (long)A * O, ws + w.b3f, R, O, D, 2, st, A, D,
                       (long)Opad * D, O, O));
    if (crit && !crit_done) {
        // the output-layer kernel declined (more than 128 outputs per head, fewer than 4, odd strides): the criterion as its own
        // kernel on the stored contours -- same loss, same d loss / d(pre-sigmoid), one pass more
        AS_STEP("loss", st, as_euclid_masked_fwd_bwd_presigmoid(out, crit->tgt, crit->tgt_T, crit->lengths, (int32_t)(rows / crit->T), crit->T, A,
                                                               d.n_samp, crit->scale, crit->loss, crit->dout, ws + w.lpart, st));
    }
    return 0;
}

// Backward of the heads in two parts so that the caller can put them on different streams:
//  head_bwd_dx : the input-gradient chain (critical path towards the GRU backward)
//  head_bwd_dw : weight/bias gradients + LayerNorm-affine unfold (only consumes what the chain left in ws)
// relu_src: optional [rows][H] activation whose ReLU produced x (its mask is fused into the last step)
// presig: dout already is the gradient w.r.t. the pre-sigmoid activations (as_opts.dout_presigmoid): read in place
int head_bwd_dx(const as_dims& d, const as_layout& L, const float* P, const float* out, const float* dout, int64_t rows,
                float* dx, const float* relu_src, float* ws, hipStream_t st, bool presig = false) {
    (void)L; (void)P;
    const int A = d.n_art, H = d.hidden, O = 2 * d.n_samp;
    const HeadWs w = head_ws(d, rows);
    const int R = (int)rows;
    const long AD = (long)A * D, AO = (long)A * O;
    const float* dpre3 = presig ? dout : ws + w.dpre3;
    if (!presig) AS_STEP("headb.sigmoid", st, as_sigmoid_bwd(out, dout, ws + w.dpre3, rows * AO, st));
    const int Opad = (int)as_round_up(O, 32);
    const unsigned long long* bits1 = reinterpret_cast<const unsigned long long*>(ws + w.bits1);
    const unsigned long long* bits2 = reinterpret_cast<const unsigned long long*>(ws + w.bits2);
    // d(r2hat) = dpre3 . W3' through LayerNorm 3 and ReLU 2 in one kernel (the reduction runs over Opad: the folded
    // weights' rows O .. Opad-1 are zeros); else GEMM + the row kernel.  The ReLU masks come as bit words (32 B per row).
    as_lin b3{};
    b3.A = dpre3; b3.lda = AO; b3.a_batch = O;
    b3.B = ws + w.w3f; b3.ldb = D; b3.b_batch = (long)Opad * D; b3.b_kc = 0;
    b3.C = ws + w.dz2; b3.ldc = AD; b3.c_batch = D;
    b3.M = R; b3.N = D; b3.K = Opad; b3.ka_valid = O; b3.batch = A; b3.epi = 2;
    b3.xhat = ws + w.r2hat; b3.ldx = AD; b3.x_batch = D; b3.rstd_in = ws + w.rstd2; b3.bits_in = bits2;
    b3.Bp = reinterpret_cast<const uint16_t*>(ws + w.w3tp); b3.bp_rows = D; b3.bp_batch = as_planes_batch_stride(D, Opad);
    b3.bp_plane = A * b3.bp_batch;
    int took;
    {
        AS_PROF("headb.dx3", st);
        took = as_lin_try(&b3, st);
        AS_REQUIRE(took >= 0, took, "head dx3: launch failed");
    }
    if (!took) {
        AS_STEP("headb.dx3", st, gemm_nn(dpre3, AO, ws + w.w3f, D, ws + w.dz2, AD, R, D, O, st, A, O, (long)Opad * D, D));
        AS_STEP("headb.norm2", st, as_normalize_bwd(ws + w.dz2, ws + w.r2hat, ws + w.rstd2, nullptr, ws + w.dz2, rows * A, D, st, bits2));
    }
    as_lin b2 = b3;
    b2.A = ws + w.dz2; b2.lda = AD; b2.a_batch = D;
    b2.B = ws + w.w2f; b2.b_batch = (long)D * D;
    b2.C = ws + w.dz1; b2.K = D; b2.ka_valid = D;
    b2.xhat = ws + w.r1hat; b2.rstd_in = ws + w.rstd1; b2.bits_in = bits1;
    b2.Bp = reinterpret_cast<const uint16_t*>(ws + w.w2tp); b2.bp_batch = as_planes_batch_stride(D, D); b2.bp_plane = A * b2.bp_batch;
    {
        AS_PROF("headb.dx2", st);
        took = as_lin_try(&b2, st);
        AS_REQUIRE(took >= 0, took, "head dx2: launch failed");
    }
    if (!took) {
        AS_STEP("headb.dx2", st, gemm_nn(ws + w.dz2, AD, ws + w.w2f, D, ws + w.dz1, AD, R, D, D, st, A, D, (long)D * D, D));
        AS_STEP("headb.norm1", st, as_normalize_bwd(ws + w.dz1, ws + w.r1hat, ws + w.rstd1, nullptr, ws + w.dz1, rows * A, D, st, bits1));
    }
    // d(x_hat) = dz1 . W1' summed over the heads: N = H columns under an A * 256-long reduction.  Split arithmetic: 64 x 128
    // tiles, the reduction cut into chunks whose partial sums the LayerNorm backward below adds (as_lin_plain_s6); else the
    // general kernel (100 x 2 tiles of 64 x 64, split K over the main-stream slab)
    int dx1_slabs = 0;
    if (H <= 256 && H % 4 == 0) {
        as_lin l{};
        l.A = ws + w.dz1; l.lda = AD;
        l.Bp = reinterpret_cast<const uint16_t*>(ws + w.w1tp); l.bp_rows = (int)as_round_up(H, 128);
        l.bp_batch = as_planes_batch_stride(l.bp_rows, (int)AD); l.bp_plane = l.bp_batch;
        l.C = ws + w.slab; l.ldc = H;
        l.M = R; l.N = H; l.K = (int)AD; l.batch = 1;
        const int want = (int)std::max<int64_t>(1, std::min<int64_t>(8, 512 / std::max(1, as_cdiv(R, 64))));
        const int slabs = as_lin_plain_s6_slabs((int)AD, want);
        if ((int64_t)slabs * R * H <= SLAB_FLOATS) {
            AS_PROF("headb.dx1", st);
            const int took1 = as_lin_plain_s6(&l, want, (long)R * H, st);
            AS_REQUIRE(took1 >= 0, took1, "head dx1: launch failed");
            if (took1) dx1_slabs = slabs;
        }
    }
    if (!dx1_slabs)
        AS_STEP("headb.dx1", st, gemm_nn(ws + w.dz1, AD, ws + w.w1f, H, ws + w.dxhat, H, R, H, (int)AD, st, 1, 0, 0, 0, ws + w.slab));
    AS_STEP("headb.norm0", st, as_normalize_bwd(dx1_slabs ? ws + w.slab : ws + w.dxhat, ws + w.xhat, ws + w.rstd0, relu_src, dx, rows, H, st, nullptr,
                                                 dx1_slabs ? dx1_slabs : 1, (long)R * H));
    return 0;
}

// trunk (optional): the trunk Linear's weight gradient dzlin^T . y1 rides in the same launch (ArtSpeech: N = 2H = 256)
struct TrunkJob { const float* dz; const float* y; float* dW; float* db; int H; };

// part: 0 = everything; 1 = layers 3 and 1 (+ the trunk's); 2 = layer 2 + the LayerNorm-affine unfold of all three.  The
// model's backward issues part 1 beside the layer-1 recurrence and part 2 beside the layer-0 recurrence: each is one round
// of <= 192 long-lived workgroups, so that the GRU input-gradient GEMM between the two recurrences finds the chip free
// (a single launch of all four problems kept 184 CUs for ~240 us and that GEMM waited for CUs: 113 us instead of 52).
int head_bwd_dw(const as_dims& d, const as_layout& L, const float* P, int64_t rows, float* G, float* ws, float* slab,
                hipStream_t st, const float* dpre3_in = nullptr, int cu_budget = 0, long slab_floats = SLAB_FLOATS,
                const TrunkJob* trunk = nullptr, int part = 0, int unfold_layers = -1) {
    const int A = d.n_art, H = d.hidden, O = 2 * d.n_samp;
    const HeadWs w = head_ws(d, rows);
    const int R = (int)rows;
    const long AD = (long)A * D, AO = (long)A * O;
    // each weight-gradient GEMM also emits the bias gradient = column sums of its A operand
    const float* dpre3 = dpre3_in ? dpre3_in : ws + w.dpre3;
    // ---- several of them as ONE launch of 128 x 256 tiles + one reduce (wgrad_f32.hip, as_wgrad_multi).  Layer 1 is posed
    // transposed (dW1'^T = xhat^T . dz1: 11 tiles of 128 x 256 like the others instead of 22 of 128 x 128): its result is
    // stored transposed and its bias gradient is the column sum of the B operand.
    as_wgrad_job jobs[4] = {};
    as_gemm& g3 = jobs[0].g;
    g3.A = dpre3; g3.a_i = 1; g3.a_k = AO; g3.a_batch = O; g3.M = O;
    g3.B = ws + w.r2hat; g3.b_j = 1; g3.b_k = AD; g3.b_batch = D; g3.N = D;
    g3.C = ws + w.dw3f; g3.ldc = D; g3.c_batch = (long)O * D; g3.K = R; g3.batch = A;
    g3.colsum = G + L.b3; g3.colsum_batch = O;
    as_gemm& g1 = jobs[1].g;
    g1 = g3;
    g1.A = ws + w.xhat; g1.a_k = H; g1.a_batch = 0; g1.M = H;
    g1.B = ws + w.dz1;
    g1.C = ws + w.dw1f; g1.ldc = H; g1.c_batch = (long)D * H;
    g1.colsum = nullptr; g1.colsum_batch = 0;
    jobs[1].colsum_b = G + L.b1; jobs[1].colsum_b_batch = D; jobs[1].c_trans = 1;
    as_wgrad_job job2{};
    as_gemm& g2 = job2.g;
    g2 = g3;
    g2.A = ws + w.dz2; g2.a_k = AD; g2.a_batch = D; g2.M = D;
    g2.B = ws + w.r1hat;
    g2.C = ws + w.dw2f; g2.c_batch = (long)D * D;
    g2.colsum = G + L.b2; g2.colsum_batch = D;
    const bool multi_ok = R % 32 == 0 && R >= 512;
    if (part == 0 || part == 1) {
        int n = 2;
        const bool trunk_rides = trunk && 2 * trunk->H == 256;
        if (trunk_rides) {
            as_gemm& gt = jobs[n].g;
            gt.A = trunk->dz; gt.a_i = 1; gt.a_k = trunk->H; gt.M = trunk->H;
            gt.B = trunk->y; gt.b_j = 1; gt.b_k = 2 * trunk->H; gt.N = 2 * trunk->H;
            gt.C = trunk->dW; gt.ldc = 2 * trunk->H; gt.K = R; gt.batch = 1;
            gt.colsum = trunk->db; gt.colsum_batch = 0;
            ++n;
        }
        if (part == 0) jobs[n++] = job2;
        int took = 0;
        if (multi_ok) {
            AS_PROF(part == 0 ? "headb.dw_fused" : "headb.dw31", st);
            took = as_wgrad_multi(jobs, n, slab, slab_floats, cu_budget, st, cu_budget > 0);   // (beside a recurrence: exact, see gemm_internal.h)
            AS_REQUIRE(took >= 0, took, "head weight gradients: launch failed");
        }
        if (!took) {
            AS_STEP("headb.dw3", st, gemm_tn(dpre3, AO, ws + w.r2hat, AD, ws + w.dw3f, D, O, D, R, st, slab, G + L.b3, O, A, O, D, (long)O * D, 0, 0, 0, cu_budget));
            AS_STEP("headb.dw1", st, gemm_tn(ws + w.dz1, AD, ws + w.xhat, H, ws + w.dw1f, H, (int)AD, H, R, st, slab, G + L.b1, 0, 1, 0, 0, 0, 0, 0, 0, cu_budget));
            if (part == 0)
                AS_STEP("headb.dw2", st, gemm_tn(ws + w.dz2, AD, ws + w.r1hat, AD, ws + w.dw2f, D, D, D, R, st, slab, G + L.b2, D, A, D, D, (long)D * D, 0, 0, 0, cu_budget));
        }
        if (trunk && !(took && trunk_rides))
            AS_STEP("trunkb.dw", st, gemm_tn(trunk->dz, trunk->H, trunk->y, 2 * trunk->H, trunk->dW, 2 * trunk->H, trunk->H, 2 * trunk->H, R, st, slab, trunk->db, 0));
    } else {
        int took = 0;
        if (multi_ok) {
            AS_PROF("headb.dw2", st);
            took = as_wgrad_multi(&job2, 1, slab, slab_floats, cu_budget, st, cu_budget > 0);
            AS_REQUIRE(took >= 0, took, "head weight gradients: launch failed");
        }
        if (!took)
            AS_STEP("headb.dw2", st, gemm_tn(ws + w.dz2, AD, ws + w.r1hat, AD, ws + w.dw2f, D, D, D, R, st, slab, G + L.b2, D, A, D, D, (long)D * D, 0, 0, 0, cu_budget));
    }
    // unfold the LayerNorm affines (bit 0: layer 3, bit 1: layer 2, bit 2: layer 1; default: all three once everything is
    // there, i.e. with part 0 or 2): one launch
    if (unfold_layers < 0) unfold_layers = part == 1 ? 0 : 7;
    if (unfold_layers == 0) return 0;
    const float* dWf[3]; const float* dbf[3]; const float* Wp[3]; const float* gm[3]; const float* bt[3];
    float* dWo[3]; float* dgm[3]; float* dbt[3];
    int Rs[3], Ks[3], cnt = 0;
    auto add = [&](int64_t dwf, int64_t b, int64_t wgt, int64_t g_, int64_t b_, int r, int k) {
        dWf[cnt] = ws + dwf; dbf[cnt] = G + b; Wp[cnt] = P + wgt; gm[cnt] = P + g_; bt[cnt] = P + b_;
        dWo[cnt] = G + wgt; dgm[cnt] = G + g_; dbt[cnt] = G + b_; Rs[cnt] = r; Ks[cnt] = k; ++cnt;
    };
    if (unfold_layers & 1) add(w.dw3f, L.b3, L.w3, L.ln3_g, L.ln3_b, O, D);
    if (unfold_layers & 2) add(w.dw2f, L.b2, L.w2, L.ln2_g, L.ln2_b, D, D);
    if (unfold_layers & 4) add(w.dw1f, L.b1, L.w1, L.ln1_g, L.ln1_b, D, H);
    AS_STEP("headb.unfold", st, as_unfold3(dWf, dbf, Wp, gm, bt, dWo, dgm, dbt, Rs, Ks, A, st, cnt));
    return 0;
}

// ---- per-(device, caller stream) state: a library-owned side stream + fork/join events, and the "head gradients are
// final" event.  Weight-gradient GEMMs (throughput bound, fill the chip) run on the side stream beside the GRU backward
// recurrences (latency bound, 2*B workgroups) instead of after them.  Fork/join are stream-ordered event waits: no host
// synchronisation, capturable in a HIP graph.  Keyed by the CALLER'S stream, so two models (or two host threads) that
// drive the library on different streams of one device never share a side stream or an event; calls on ONE stream are
// ordered by that stream anyway.
struct StreamState {
    int dev = -1;
    hipStream_t owner = nullptr;
    hipStream_t side = nullptr, side2 = nullptr;                  // created on first use with overlap on
    hipEvent_t fork[3] = {nullptr, nullptr, nullptr}, join = nullptr;
    hipEvent_t fork2[3] = {nullptr, nullptr, nullptr}, join2 = nullptr;   // second side stream
    hipEvent_t heads = nullptr;                                   // recorded by as_artspeech_bwd
};
int g_overlap = -1;  // -1: not decided yet (environment), 0: off, 1: on
// AS_ONE_SIDE_STREAM (ablation): the hidden-to-hidden weight gradients queue on the first side stream (round-2 start)
const bool g_one_side = AS_DIAG_SET("AS_ONE_SIDE_STREAM");
std::mutex g_state_mu;
std::vector<StreamState*> g_states;

StreamState* state_for(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_state_mu);
    for (StreamState* p : g_states)
        if (p->dev == dev && p->owner == st) return p;
    if (g_states.size() >= 256) return nullptr;
    // nothing may be created inside a stream capture: the caller warms the library up once before capturing
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    StreamState* p = new StreamState;
    p->dev = dev;
    p->owner = st;
    bool ok = hipEventCreateWithFlags(&p->heads, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&p->join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&p->join2, hipEventDisableTiming) == hipSuccess;
    for (auto& e : p->fork) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (auto& e : p->fork2) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        delete p;
        return nullptr;
    }
    g_states.push_back(p);
    return p;
}
// the side stream of the caller's stream, or nullptr when overlap is off / unavailable
StreamState* side_for(hipStream_t st) {
    {
        std::lock_guard<std::mutex> lock(g_state_mu);
        if (g_overlap < 0) g_overlap = getenv("ARTSPEECH_NO_OVERLAP") ? 0 : 1;
        if (!g_overlap) return nullptr;
    }
    StreamState* p = state_for(st);
    if (!p) return nullptr;   // no state for this stream (table full, or first call inside a capture): in order on `st`
    {
        std::lock_guard<std::mutex> lock(g_state_mu);
        if (p->side) return p;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
        if (!p->side && hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) {
            p->side = nullptr;
            return nullptr;
        }
        // optional: without it, its work goes to `side`
        if (!g_one_side && !p->side2 && hipStreamCreateWithFlags(&p->side2, hipStreamNonBlocking) != hipSuccess) p->side2 = nullptr;
    }
    return p;
}
// "gradients of the trunk Linear and of all heads are final" -- recorded by as_artspeech_bwd on the stream that produced
// them, so that a data-parallel host can start their all-reduce while the GRU backward is still running.
int record_heads_done(hipStream_t owner, hipStream_t s) {
    StreamState* p = state_for(owner);
    // no per-stream state (more than 256 distinct caller streams, or the first call on a stream inside a capture): the
    // backward itself does not need the event -- everything then runs in order on the caller's stream -- only
    // as_artspeech_wait_head_grads() does, and it reports the missing state itself
    if (!p) return 0;
    const hipError_t e = hipEventRecord(p->heads, s);
    AS_REQUIRE(e == hipSuccess, (int)e, "as_artspeech_bwd: hipEventRecord failed: %s", hipGetErrorString(e));
    return 0;
}

// fork whose event the producing kernel may already carry (as_stop_event_set before its launch): `left` = what
// as_stop_event_take() returned after the launch -- non-null: nobody consumed it, record it the ordinary way
int fork_after(hipStream_t from, hipStream_t to, hipEvent_t ev, hipEvent_t left) {
    hipError_t e = left ? hipEventRecord(ev, from) : hipSuccess;
    if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
    if (e != hipSuccess) {
        as_set_error("stream fork failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

int fork_to(hipStream_t from, hipStream_t to, hipEvent_t ev) {
    hipError_t e = hipEventRecord(ev, from);
    if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
    if (e != hipSuccess) {
        as_set_error("stream fork/join failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

}  // namespace

extern "C" void as_set_overlap(int32_t on) {
    std::lock_guard<std::mutex> lock(g_state_mu);
    g_overlap = on ? 1 : 0;
}

extern "C" int as_artspeech_wait_head_grads(void* compute_stream, void* waiting_stream) {
    StreamState* p = state_for((hipStream_t)compute_stream);
    AS_REQUIRE(p, AS_ERR_UNSUPPORTED, "as_artspeech_wait_head_grads: as_artspeech_bwd has not run on that stream");
    const hipError_t e = hipStreamWaitEvent((hipStream_t)waiting_stream, p->heads, 0);
    AS_REQUIRE(e == hipSuccess, (int)e, "as_artspeech_wait_head_grads: hipStreamWaitEvent failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int as_artspeech_layout(const as_dims* d, as_layout* out) {
    AS_TRY(check_dims(d, "as_artspeech_layout"));
    AS_REQUIRE(out, AS_ERR_BAD_ARG, "as_artspeech_layout: null out");
    const int64_t V = d->vocab, A = d->n_art, E = d->embed, H = d->hidden, N = d->n_samp;
    Carve c;
    as_layout L{};
    L.embedding = c.take(V * E);
    if (!d->simple) {
        const int64_t in[2] = {E, 2 * H};
        for (int l = 0; l < 2; ++l) {
            L.w_ih[l] = c.take(2 * 3 * H * in[l]);
            L.b_ih[l] = c.take(2 * 3 * H);
            L.w_hh[l] = c.take(2 * 3 * H * H);
            L.b_hh[l] = c.take(2 * 3 * H);
        }
        L.lin_w = c.take(H * 2 * H);
    } else {
        for (int l = 0; l < 2; ++l) L.w_ih[l] = L.b_ih[l] = L.w_hh[l] = L.b_hh[l] = -1;
        L.lin_w = c.take(H * E);
    }
    L.lin_b = c.take(H);
    L.ln1_g = c.take(A * H);
    L.ln1_b = c.take(A * H);
    L.w1 = c.take(A * D * H);
    L.b1 = c.take(A * D);
    L.ln3_g = c.take(A * D);
    L.ln3_b = c.take(A * D);
    L.w3 = c.take(A * 2 * N * D);
    L.b3 = c.take(A * 2 * N);
    // last: the group whose gradient a pipelined training loop computes one step late (as_opts.defer_dw2): one
    // contiguous slice [ln2_g, total) for its all-reduce and its Adam update
    L.ln2_g = c.take(A * D);
    L.ln2_b = c.take(A * D);
    L.w2 = c.take(A * D * D);
    L.b2 = c.take(A * D);
    L.total = c.off;
    *out = L;
    return 0;
}

extern "C" int64_t as_head_workspace_floats(const as_dims* d, int64_t rows) {
    if (check_dims(d, "as_head_workspace_floats") != 0 || rows <= 0) return -1;
    return head_ws(*d, rows).total;
}

extern "C" int64_t as_artspeech_workspace_floats(const as_dims* d, int32_t B, int32_t T) {
    if (check_dims(d, "as_artspeech_workspace_floats") != 0 || B <= 0 || T <= 0) return -1;
    return model_ws(*d, B, T).total;
}

extern "C" int as_head_fwd(const as_dims* d, const as_layout* lay, const float* params, const float* x, int64_t rows,
                           float* out, float* ws, int32_t train, void* stream) {
    (void)train;  // the forward keeps its activations in ws either way
    AS_TRY(check_dims(d, "as_head_fwd"));
    AS_REQUIRE(lay && params && x && out && ws && rows > 0 && rows < (1LL << 31), AS_ERR_BAD_ARG, "as_head_fwd: bad argument");
    AS_TRY(head_fold(*d, *lay, params, rows, ws, (hipStream_t)stream));
    return head_fwd_impl(*d, *lay, params, x, rows, out, ws, (hipStream_t)stream);
}

extern "C" int as_head_bwd(const as_dims* d, const as_layout* lay, const float* params, const float* out, const float* dout,
                           int64_t rows, float* dx, float* grads, float* ws, void* stream) {
    AS_TRY(check_dims(d, "as_head_bwd"));
    AS_REQUIRE(lay && params && out && dout && dx && grads && ws && rows > 0 && rows < (1LL << 31), AS_ERR_BAD_ARG,
               "as_head_bwd: bad argument");
    const HeadWs hw = head_ws(*d, rows);
    AS_TRY(head_bwd_dx(*d, *lay, params, out, dout, rows, dx, nullptr, ws, (hipStream_t)stream));
    return head_bwd_dw(*d, *lay, params, rows, grads, ws, ws + hw.slab, (hipStream_t)stream);
}

extern "C" int as_artspeech_fwd(const as_dims* d, const float* P, const int64_t* tokens, int64_t tok_stride,
                                const int32_t* lengths, int32_t B, int32_t T, float* out, float* ws, int32_t train,
                                const as_opts* opts, void* stream) {
    AS_TRY(check_dims(d, "as_artspeech_fwd"));
    const float pdrop = (opts && train) ? opts->gru_dropout : 0.f;
    AS_REQUIRE(pdrop >= 0.f && pdrop < 1.f, AS_ERR_BAD_ARG, "as_artspeech_fwd: dropout %g not in [0, 1)", (double)pdrop);
    AS_REQUIRE(P && tokens && out && ws && B > 0 && T > 0 && tok_stride >= T, AS_ERR_BAD_ARG, "as_artspeech_fwd: bad argument");
    AS_REQUIRE(d->simple || lengths, AS_ERR_BAD_ARG, "as_artspeech_fwd: lengths required for the GRU model");
    AS_REQUIRE((int64_t)B * T < (1LL << 31), AS_ERR_BAD_ARG, "as_artspeech_fwd: B*T too large");
    hipStream_t st = (hipStream_t)stream;
    as_layout L;
    AS_TRY(as_artspeech_layout(d, &L));
    const ModelWs w = model_ws(*d, B, T);
    const int V = d->vocab, E = d->embed, H = d->hidden, R = B * T;
    // the weight folds depend on the parameters only: run them on the side stream beside the recurrences
    StreamState* sd = d->simple ? nullptr : side_for(st);
    const bool late = opts && opts->fold_wait_event;
    if (late) {   // a deferred parameter update (as_opts) must land before the heads' weights are folded
        const hipError_t e = hipStreamWaitEvent(sd ? sd->side : st, (hipEvent_t)opts->fold_wait_event, 0);
        AS_REQUIRE(e == hipSuccess, (int)e, "as_artspeech_fwd: cannot wait for fold_wait_event: %s", hipGetErrorString(e));
    }
    if (sd) {
        // the folds read the parameters: behind everything `st` holds (the previous optimizer step).  With a deferred update
        // the event above already stands behind that (the update was enqueued after it), and an event record on `st` is a
        // barrier packet in front of the step's first kernel: skipped then
        if (!late) AS_TRY(fork_to(st, sd->side, sd->fork[0]));
        AS_TRY(head_fold(*d, L, P, R, ws + w.head, sd->side));
        AS_TRY(as_count_bad_tokens(tokens, tok_stride, T, R, V, reinterpret_cast<int*>(ws + w.tokflag), sd->side));
        if (hipEventRecord(sd->join, sd->side) != hipSuccess) {
            as_set_error("as_artspeech_fwd: event record failed");
            return AS_ERR_BAD_ARG;
        }
    } else {
        AS_TRY(head_fold(*d, L, P, R, ws + w.head, st));
        AS_TRY(as_count_bad_tokens(tokens, tok_stride, T, R, V, reinterpret_cast<int*>(ws + w.tokflag), st));
    }
    if (!d->simple) {
        // token table of layer-0 input projections, both directions: [V][2][3H]
        // the step's first kernel, on its critical path: a small dedicated kernel (rowops.hip) while its staging area --
        // (4 E + 64 (E + 1)) floats of dynamic LDS -- stays inside the 64 KB a launch gets without asking (E <= 240)
        if ((long)V * E <= 16384 && ((size_t)4 * E + 64 * ((size_t)E + 1)) * sizeof(float) <= 65536)
            AS_STEP("gru.table0", st, as_token_table(P + L.embedding, P + L.w_ih[0], P + L.b_ih[0], V, 6 * H, E, ws + w.tab0, st));
        else
            AS_STEP("gru.table0", st, gemm_nt(P + L.embedding, E, P + L.w_ih[0], E, ws + w.tab0, 6 * H, P + L.b_ih[0], V, 6 * H, E, 0, st));
        AS_STEP("gru.fwd_l0", st, as_gru_bidir_fwd_tokens(ws + w.tab0, tokens, tok_stride, V, P + L.w_hh[0], P + L.b_hh[0], lengths, B, T, H, ws + w.y0,
                                train ? ws + w.g0 : nullptr, st));
        const float* l1_in = ws + w.y0;
        if (pdrop > 0.f) {  // nn.GRU inter-layer dropout: layer 1 sees the dropped layer-0 output
            AS_STEP("gru.dropout", st, as_dropout(ws + w.y0, ws + w.y0d, (long)R * 2 * H, pdrop, opts->dropout_seed, st));
            l1_in = ws + w.y0d;
        }
        {
            // input projection of GRU layer 1: [R][2H] . [6H][2H]^T + b.  (diagnostic build, AS_XPROJ_LIN = 64 | 32: the LDS-DMA
            // kernel of the head layers on three 256-column blocks instead of the general kernel's 64 x 64 tiles)
            static const int xproj_lin = AS_DIAG_INT("AS_XPROJ_LIN", 0);
            int took = 0;
            if (xproj_lin && 6 * H % 256 == 0) {
                as_lin l{};
                l.A = l1_in; l.lda = 2 * H; l.a_batch = 0;
                l.B = P + L.w_ih[1]; l.ldb = 2 * H; l.b_batch = 256L * 2 * H; l.b_kc = 1;
                l.C = ws + w.xp1; l.ldc = 6 * H; l.c_batch = 256;
                l.bias = P + L.b_ih[1]; l.bias_batch = 256;
                l.M = R; l.N = 256; l.K = 2 * H; l.batch = 6 * H / 256; l.tile_rows = xproj_lin;
                AS_PROF("gru.xproj1", st);
                took = as_lin_try(&l, st);
                AS_REQUIRE(took >= 0, took, "gru.xproj1: launch failed");
            }
            // (split arithmetic measured here twice and not kept: the plane-fed kernel of the heads 39.5 vs 43 us + a 10-us plane
            // launch in front of the step's first kernel; both operands split in the kernel, gemm_s6.hip, 35.3 vs 33.1 us alone,
            // tools/bench_linear.py -- at 2.5 GFLOP the launch is bound by its prologue and fill, not by the matrix pipe)
            if (!took)
                AS_STEP("gru.xproj1", st, gemm_nt(l1_in, 2 * H, P + L.w_ih[1], 2 * H, ws + w.xp1, 6 * H, P + L.b_ih[1], R, 6 * H, 2 * H, 0, st));
        }
        AS_STEP("gru.fwd_l1", st, as_gru_bidir_fwd(ws + w.xp1, nullptr, 0, P + L.w_hh[1], P + L.b_hh[1], lengths, B, T, H, ws + w.y1,
                                train ? ws + w.g1 : nullptr, st));
        // (trunk Linear: 10.5 us on the fp32 instruction, 14 - 23 us on either split kernel: 50 - 200 short workgroups)
        AS_STEP("trunk.linear", st, gemm_nt(ws + w.y1, 2 * H, P + L.lin_w, 2 * H, ws + w.lin, H, P + L.lin_b, R, H, 2 * H, 1, st));
    } else if (pdrop > 0.f) {
        // SimpleArtSpeech in training mode (models.py:64,85): Dropout acts on the embedded frame, so every position has its
        // own mask and the token-table fold below does not apply: gather -> counter mask -> Linear + ReLU on all frames
        AS_TRY(as_gather_rows(P + L.embedding, tokens, tok_stride, T, R, E, ws + w.y0, st, V));
        AS_TRY(as_dropout(ws + w.y0, ws + w.y0, (long)R * E, pdrop, opts->dropout_seed, st));
        AS_TRY(gemm_nt(ws + w.y0, E, P + L.lin_w, E, ws + w.lin, H, P + L.lin_b, R, H, E, 1, st));
    } else {
        AS_TRY(gemm_nt(P + L.embedding, E, P + L.lin_w, E, ws + w.tab0, H, P + L.lin_b, V, H, E, 1, st));
        AS_TRY(as_gather_rows(ws + w.tab0, tokens, tok_stride, T, R, H, ws + w.lin, st, V));
    }
    if (sd && hipStreamWaitEvent(st, sd->join, 0) != hipSuccess) {  // folded weights ready before head GEMM 1
        as_set_error("as_artspeech_fwd: stream wait failed");
        return AS_ERR_BAD_ARG;
    }
    if (opts && opts->loss_targets) {
        AS_REQUIRE(train && lengths && opts->loss_out && opts->loss_dout && opts->loss_tgt_T >= T, AS_ERR_BAD_ARG,
                   "as_artspeech_fwd: fused criterion needs train, lengths, loss_out, loss_dout and loss_tgt_T >= T");
        const Criterion crit{opts->loss_targets, (long)opts->loss_tgt_T, lengths, T, opts->loss_scale, opts->loss_out, opts->loss_dout};
        return head_fwd_impl(*d, L, P, ws + w.lin, R, out, ws + w.head, st, &crit);
    }
    return head_fwd_impl(*d, L, P, ws + w.lin, R, out, ws + w.head, st);
}

extern "C" int as_artspeech_dw2(const as_dims* d, const float* P, int32_t B, int32_t T, float* G, float* ws, void* stream) {
    AS_TRY(check_dims(d, "as_artspeech_dw2"));
    AS_REQUIRE(!d->simple && P && G && ws && B > 0 && T > 0, AS_ERR_BAD_ARG, "as_artspeech_dw2: bad argument");
    as_layout L;
    AS_TRY(as_artspeech_layout(d, &L));
    const ModelWs w = model_ws(*d, B, T);
    const HeadWs hw = head_ws(*d, (int64_t)B * T);
    float* hws = ws + w.head;
    // beside a forward recurrence (2 B workgroups): 192 CUs to count on
    return head_bwd_dw(*d, L, P, (int64_t)B * T, G, hws, hws + hw.slab2, (hipStream_t)stream, nullptr, 192, SLAB2_FLOATS, nullptr, 2, 2);
}

extern "C" int as_artspeech_bwd(const as_dims* d, const float* P, const int64_t* tokens, int64_t tok_stride,
                                const int32_t* lengths, int32_t B, int32_t T, const float* out, const float* dout, float* G,
                                float* ws, const as_opts* opts, void* stream) {
    AS_TRY(check_dims(d, "as_artspeech_bwd"));
    const float pdrop = opts ? opts->gru_dropout : 0.f;
    AS_REQUIRE(pdrop >= 0.f && pdrop < 1.f, AS_ERR_BAD_ARG, "as_artspeech_bwd: bad dropout option");
    AS_REQUIRE(P && tokens && out && dout && G && ws && B > 0 && T > 0 && tok_stride >= T, AS_ERR_BAD_ARG,
               "as_artspeech_bwd: bad argument");
    AS_REQUIRE(d->simple || lengths, AS_ERR_BAD_ARG, "as_artspeech_bwd: lengths required for the GRU model");
    hipStream_t st = (hipStream_t)stream;
    as_layout L;
    AS_TRY(as_artspeech_layout(d, &L));
    const ModelWs w = model_ws(*d, B, T);
    const HeadWs hw = head_ws(*d, (int64_t)B * T);
    const int V = d->vocab, E = d->embed, H = d->hidden, R = B * T;
    float* slab = ws + w.head + hw.slab;
    float* slab2 = ws + w.head + hw.slab2;
    float* dzlin = ws + w.head + hw.dxhat;  // reused: d(trunk pre-activation) [R][H]
    float* hws = ws + w.head;
    // heads: input-gradient chain (+ the trunk ReLU mask fused into the last normalize-backward)
    const bool presig = opts && opts->dout_presigmoid;
    const float* dpre3 = presig ? dout : nullptr;
    AS_TRY(head_bwd_dx(*d, L, P, out, dout, R, dzlin, ws + w.lin, hws, st, presig));
    AS_REQUIRE(!(d->simple && opts && opts->defer_dw2), AS_ERR_UNSUPPORTED, "as_artspeech_bwd: defer_dw2 is for the GRU model");
    if (d->simple) {
        AS_TRY(head_bwd_dw(*d, L, P, R, G, hws, slab, st, dpre3));
        AS_TRY(record_heads_done(st, st));
        if (pdrop > 0.f) {  // per-frame path of the forward: Linear backward on all frames, mask regenerated from the seed
            AS_TRY(gemm_tn(dzlin, H, ws + w.y0, E, G + L.lin_w, E, H, E, R, st, slab, G + L.lin_b, 0));
            AS_TRY(gemm_nn(dzlin, H, P + L.lin_w, E, ws + w.dy0, E, R, E, H, st));
            AS_TRY(as_dropout(ws + w.dy0, ws + w.dy0, (long)R * E, pdrop, opts->dropout_seed, st));
            AS_TRY(as_token_segsum(ws + w.dy0, tokens, tok_stride, T, R, E, V, G + L.embedding, st, slab, SLAB_FLOATS));
            return 0;
        }
        // lin = gather(relu(Emb Wl^T + bl)); dzlin already carries the ReLU mask of the gathered rows
        AS_TRY(as_token_segsum(dzlin, tokens, tok_stride, T, R, H, V, ws + w.dtab0, st, slab, SLAB_FLOATS));
        AS_TRY(gemm_tn(ws + w.dtab0, H, P + L.embedding, E, G + L.lin_w, E, H, E, V, st, slab, G + L.lin_b, 0));
        AS_TRY(gemm_nn(ws + w.dtab0, H, P + L.lin_w, E, G + L.embedding, E, V, E, H, st));
        return 0;
    }
    StreamState* sd = side_for(st);
    hipStream_t s2 = sd ? sd->side : st;      // no side stream: everything in order on `st`
    float* sl2 = sd ? slab2 : slab;
    // second side stream: the small hidden-to-hidden weight gradients (few workgroups, latency bound), so that they run
    // beside the head / input-projection ones instead of queueing behind them: -9 us per step
    hipStream_t s3 = (sd && sd->side2) ? sd->side2 : s2;
    float* sl3 = (sd && sd->side2) ? ws + w.head + hw.slab3 : sl2;
    // ablation (AS_PLAIN_FORKS): every fork an event record on `st`.  Also while the per-phase timers are on (as_profile_enable):
    // a timing event recorded right behind an event-carrying dispatch reads ~20 us late, which would inflate the phase.
    static const bool plain_forks_env = AS_DIAG_SET("AS_PLAIN_FORKS");
    hipStreamCaptureStatus cap_ = hipStreamCaptureStatusNone;   // inside a stream capture: ordinary event records (graph edges)
    const bool capturing = hipStreamIsCapturing(st, &cap_) != hipSuccess || cap_ != hipStreamCaptureStatusNone;
    const bool plain_forks = plain_forks_env || as_profile_active() || capturing;
    // ---- fork 0: head + trunk weight gradients run beside the layer-1 recurrence.  The fork's event rides on the GEMM's own
    // dispatch (as_stop_event_set): no marker packet on `st` between it and the recurrence
    if (sd && !plain_forks) as_stop_event_set(sd->fork[0]);
    AS_STEP("trunkb.dx", st, gemm_nn(dzlin, H, P + L.lin_w, 2 * H, ws + w.dy1, 2 * H, R, 2 * H, H, st));
    if (sd) AS_TRY(fork_after(st, s2, sd->fork[0], plain_forks ? sd->fork[0] : as_stop_event_take()));
    AS_STEP("gru.bwd_l1", st, as_gru_bidir_bwd(ws + w.dy1, ws + w.y1, ws + w.g1, P + L.w_hh[1], lengths, B, T, H, ws + w.dgi1, ws + w.dgh1, st));
    const int side_cus = sd ? 192 : 0;  // the recurrence's 2 * B workgroups hold 64 CUs while the side stream works
    const TrunkJob trunk{dzlin, ws + w.y1, G + L.lin_w, G + L.lin_b, H};
    // with a side stream: layers 3, 1 and the trunk now, layer 2 + unfold beside the layer-0 recurrence (see head_bwd_dw)
    static const bool dw_one = AS_DIAG_SET("AS_HEAD_DW_ONE");   // ablation: everything in one launch here
    const bool defer = opts && opts->defer_dw2;                  // layer 2's is computed later by as_artspeech_dw2()
    const bool two_parts = (sd != nullptr && !dw_one) || defer;
    AS_TRY(head_bwd_dw(*d, L, P, R, G, hws, sl2, s2, dpre3, side_cus, sd ? SLAB2_FLOATS : SLAB_FLOATS, &trunk, two_parts ? 1 : 0,
                       defer ? 5 : -1));
    // [lin_w, total) of the flat gradient buffer is final from here on ([lin_w, ln2_g) with defer_dw2)
    if (!two_parts || defer) AS_TRY(record_heads_done(st, s2));
    hipEvent_t fork1_left = nullptr;
    bool fork1_armed = false;
    {
        // input gradient of GRU layer 1: [R][6H] . [6H][2H].  (diagnostic build, AS_DX1_LIN: the LDS-DMA kernel of the head
        // layers on 32-row x 256-column tiles instead of the general kernel's 64 x 64 tiles + in-kernel split-K)
        static const int dx1_lin = AS_DIAG_INT("AS_DX1_LIN", 0);   // 64 | 32 = tile rows
        int took = 0;
        if (dx1_lin && 2 * H == 256) {
            as_lin l{};
            l.A = ws + w.dgi1; l.lda = 6 * H;
            l.B = P + L.w_ih[1]; l.ldb = 2 * H; l.b_kc = 0;
            l.C = ws + w.dy0; l.ldc = 2 * H;
            l.M = R; l.N = 2 * H; l.K = 6 * H; l.batch = 1; l.tile_rows = dx1_lin;
            AS_PROF("grub.dx1", st);
            took = as_lin_try(&l, st);
            AS_REQUIRE(took >= 0, took, "grub.dx1: launch failed");
        }
        // (split arithmetic measured slower here, 56 vs 52 us: 200 workgroups x 1.2 MB of weight planes each from L2)
        if (!took) {
            if (sd && !plain_forks && !(pdrop > 0.f)) as_stop_event_set(sd->fork[1]);   // fork 1 rides on this GEMM (see fork 0)
            AS_STEP("grub.dx1", st, gemm_nn(ws + w.dgi1, 6 * H, P + L.w_ih[1], 2 * H, ws + w.dy0, 2 * H, R, 2 * H, 6 * H, st, 1, 0, 0, 0, slab));
            fork1_left = as_stop_event_take();
            fork1_armed = sd && !plain_forks && !(pdrop > 0.f);
        }
    }
    if (pdrop > 0.f)  // back through the inter-layer dropout: same mask, regenerated from the seed
        AS_STEP("gru.dropout", st, as_dropout(ws + w.dy0, ws + w.dy0, (long)R * 2 * H, pdrop, opts->dropout_seed, st));
    // ---- fork 1: layer-1 weight gradients run beside the layer-0 recurrence
    if (sd) AS_TRY(fork_after(st, s2, sd->fork[1], fork1_armed ? fork1_left : sd->fork[1]));
    if (s3 != s2) {   // ONE record on `st` for both side streams (every event record is a barrier packet on the critical stream)
        const hipError_t e = hipStreamWaitEvent(s3, sd->fork[1], 0);
        AS_REQUIRE(e == hipSuccess, (int)e, "as_artspeech_bwd: stream wait failed: %s", hipGetErrorString(e));
    }
    // layer 0 sits under the token table: the recurrence keeps per-token sums of its input-side gate gradients in LDS and
    // leaves B tables [V][6H] where dgi0 would have gone (V <= T: they fit), so there is no dgi0 and no segmented-sum
    // pass; else dgi0 + as_token_segsum below
    const int tok_sums = V <= T && as_gru_bwd_tokens_fits(V, H, T);
    // fork 2 (the tail's second job, below) rides on the recurrence's dispatch
    if (sd && !plain_forks) as_stop_event_set(s3 != s2 ? sd->fork2[2] : sd->fork[2]);
    if (tok_sums) {
        AS_PROF("gru.bwd_l0", st);
        const int rc = as_gru_bidir_bwd_tokens(ws + w.dy0, ws + w.y0, ws + w.g0, P + L.w_hh[0], lengths, B, T, H, ws + w.dgh0, tokens,
                                               tok_stride, V, ws + w.dgi0, st);
        AS_REQUIRE(rc == 1, rc < 0 ? rc : AS_ERR_UNSUPPORTED, "gru.bwd_l0: launch failed");
    } else {
        AS_STEP("gru.bwd_l0", st, as_gru_bidir_bwd(ws + w.dy0, ws + w.y0, ws + w.g0, P + L.w_hh[0], lengths, B, T, H, ws + w.dgi0, ws + w.dgh0, st));
    }
    const hipEvent_t fork2_left = (sd && !plain_forks) ? as_stop_event_take() : (sd ? (s3 != s2 ? sd->fork2[2] : sd->fork[2]) : nullptr);
    if (two_parts && !defer) {
        AS_TRY(head_bwd_dw(*d, L, P, R, G, hws, sl2, s2, dpre3, side_cus, sd ? SLAB2_FLOATS : SLAB_FLOATS, nullptr, 2));
        AS_TRY(record_heads_done(st, s2));  // [lin_w, total) of the flat gradient buffer is final from here on
    }
    AS_STEP("grub.dw_ih1", s2, gemm_tn(ws + w.dgi1, 6 * H, pdrop > 0.f ? ws + w.y0d : ws + w.y0, 2 * H, G + L.w_ih[1], 2 * H, 6 * H, 2 * H, R, s2, sl2, G + L.b_ih[1], 0));
    // dW_hh = dgh^T . h_{prev}: y shifted by -1 (forward) / +1 (reverse) frame; both directions as one batch of two
    AS_STEP("grub.dw_hh", s3, gemm_tn(ws + w.dgh1, 6 * H, ws + w.y1, 2 * H, G + L.w_hh[1], H, 3 * H, H, R, s3, sl3, G + L.b_hh[1], 3 * H, 2,
                   3 * H, H, 3L * H * H, -1, T, 2));
    // ---- layer-0 gradients.  What follows the last recurrence is the step's tail, and a cross-stream wait costs ~10 us
    // each way on top of the work it guards (measured on the timeline): the LONGER of the two remaining jobs therefore
    // stays on the caller's stream, back to back with the recurrence, and the shorter one forks off.
    //   token sums in the recurrence: hidden-to-hidden GEMM (~30 us) here, table reduction + embedding grads (~13 us) aside;
    //   otherwise: segmented sum + embedding grads (~40 us) here, the GEMM aside.
    // The first side stream has its last work of this call queued: the second one waits for it HERE, before its own tail work --
    // a wait that resolves while that stream idles through the recurrence -- so that the final join is one hop (second side
    // stream -> `st`) instead of two (14.6 -> 11 us between the last kernel and Adam; chained form: AS_CHAIN_JOIN).
    static const bool chain_join = AS_DIAG_SET("AS_CHAIN_JOIN");
    if (sd && s3 != s2 && !chain_join) AS_TRY(fork_to(s2, s3, sd->join));
    hipStream_t s_hh = tok_sums ? st : s3, s_emb = tok_sums ? s3 : st;
    float* sl_hh = tok_sums ? slab : sl3;
    if (sd) AS_TRY(fork_after(st, s3, s3 != s2 ? sd->fork2[2] : sd->fork[2], fork2_left));
    AS_STEP("grub.dw_hh", s_hh, gemm_tn(ws + w.dgh0, 6 * H, ws + w.y0, 2 * H, G + L.w_hh[0], H, 3 * H, H, R, s_hh, sl_hh, G + L.b_hh[0], 3 * H, 2,
                   3 * H, H, 3L * H * H, -1, T, 2));
    // embedding + layer-0 input projection through the token table
    if (tok_sums)
        AS_STEP("grub.segsum", s_emb, as_sum_partials(ws + w.dgi0, (long)V * 6 * H, B, ws + w.dtab0, s_emb));
    else
        AS_STEP("grub.segsum", s_emb, as_token_segsum(ws + w.dgi0, tokens, tok_stride, T, R, 6 * H, V, ws + w.dtab0, s_emb, slab, SLAB_FLOATS));
    if (E <= 256 && V <= 128 && (6 * H) % 16 == 0) {  // embedding + input-projection gradients of layer 0 from the token sums: one small launch
        AS_STEP("grub.dw_ih0", s_emb, as_emb_grads(ws + w.dtab0, P + L.embedding, P + L.w_ih[0], V, 6 * H, E, G + L.w_ih[0], G + L.b_ih[0],
                                                G + L.embedding, s_emb));
    } else {
        // (only reached with s_emb == st or with the main slab free: token sums need V <= T, these need V > 128)
        float* sl_e = tok_sums ? sl3 : slab;
        AS_STEP("grub.dw_ih0", s_emb, gemm_tn(ws + w.dtab0, 6 * H, P + L.embedding, E, G + L.w_ih[0], E, 6 * H, E, V, s_emb, sl_e, G + L.b_ih[0], 0));
        AS_STEP("grub.demb", s_emb, gemm_nn(ws + w.dtab0, 6 * H, P + L.w_ih[0], E, G + L.embedding, E, V, E, 6 * H, s_emb, 1, 0, 0, 0, sl_e));
    }
    // join: `st` continues only after the side streams' work.  One wait on `st` (each costs the tail a few microseconds):
    // the second side stream first waits for the first one, then `st` waits for it alone.
    if (sd && s3 != s2 && !chain_join) {
        // the second side stream already waited for the first one (above, before its tail work): one record behind its last
        // kernel, one wait on `st`
        AS_TRY(fork_to(s3, st, sd->join2));
    } else if (sd && s3 != s2) {
        AS_TRY(fork_to(s2, s3, sd->join));
        AS_TRY(fork_to(s3, st, sd->join2));
    } else if (sd) {
        AS_TRY(fork_to(s2, st, sd->join));
    }
    return 0;
}
