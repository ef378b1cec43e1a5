/*
 * artspeech_hip.h -- C ABI of libartspeech_hip.so, the MI355X (gfx950) engine for the
 * phoneme_to_articulation hot path of vribeiro1/artspeech.
 *
 * The reference has no FFI layer: the path sits behind PyTorch module APIs whose device work is
 * implicit (cuDNN GRU / cuBLAS / ATen).  Each entry point below replaces the device work of one of
 * those call sites (cited as file:line relative to the reference root).  Conventions for EVERY call:
 *   - plain pointers and sizes only; pointers are DEVICE pointers unless the name says host;
 *   - the caller owns every buffer (outputs, saved activations, workspaces): nothing is allocated,
 *     freed or synchronised inside; all work is enqueued on `stream` (a hipStream_t passed as void*)
 *     and is re-entrant per stream;
 *   - return value: 0 on success, otherwise a hipError_t (>0) or an AS_ERR_* code (<0);
 *     as_last_error() returns a static message for the last failing call of this thread;
 *   - all floating-point tensors are float32, row-major, contiguous unless a stride is given;
 *     the area function is float64 like the reference.
 */
#ifndef ARTSPEECH_HIP_H
#define ARTSPEECH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AS_ERR_BAD_ARG (-1)      /* null pointer / non-positive size / unsupported combination */
#define AS_ERR_UNSUPPORTED (-2)  /* e.g. a GRU hidden size that is not a multiple of 4 */
#define AS_ERR_WORKSPACE (-3)    /* caller-provided workspace too small */

#define AS_HEAD_HIDDEN 256 /* ArticulatorPredictor's fixed width (encoder_decoder/models.py:12-17) */

const char* as_version(void);    /* "artspeech_hip <semver>" */
const char* as_arch(void);       /* "gfx950" */
const char* as_last_error(void); /* message of the last failing call on this thread ("" if none) */

/* ------------------------------------------------------------------------------------------------
 * Model geometry and the flat parameter layout.
 * All parameters of ArtSpeech / SimpleArtSpeech live in ONE flat float buffer (one gradient buffer,
 * one RCCL all-reduce, one optimizer launch).  as_artspeech_layout() is the single source of truth
 * for where each state_dict tensor of the reference (encoder_decoder/models.py:99-124; key names in
 * SURVEY 8b) sits in that buffer.
 * ---------------------------------------------------------------------------------------------- */
typedef struct as_dims {
    int32_t vocab;   /* V  : nn.Embedding rows                          (models.py:110) */
    int32_t n_art;   /* A  : number of ArticulatorPredictor heads       (models.py:118-120) */
    int32_t embed;   /* E  : embed_dim                                  (models.py:104) */
    int32_t hidden;  /* H  : hidden_size (GRU width, head input width)  (models.py:105) */
    int32_t n_samp;  /* N  : n_samples per contour                      (models.py:106) */
    int32_t simple;  /* 0 = ArtSpeech (BiGRU, models.py:99), 1 = SimpleArtSpeech (models.py:53) */
} as_dims;

/* Offsets (in floats) of every parameter group inside the flat buffer. */
typedef struct as_layout {
    int64_t embedding;  /* [V][E] */
    /* GRU, layer l in {0,1}; forward and reverse direction are ADJACENT (reverse follows forward) so
       that both directions form one stacked matrix: w_ih[l] is [2][3H][I_l], I_0 = E, I_1 = 2H. */
    int64_t w_ih[2], b_ih[2]; /* [2][3H][I_l], [2][3H] */
    int64_t w_hh[2], b_hh[2]; /* [2][3H][H],   [2][3H] */
    int64_t lin_w, lin_b;     /* linear.0: [H][2H] (ArtSpeech) or [H][E] (SimpleArtSpeech), [H] */
    /* heads, stacked over the A predictors (predictors.{a}.*): */
    int64_t ln1_g, ln1_b;     /* linear.0.{weight,bias}  [A][H] */
    int64_t w1, b1;           /* linear.1                [A][256][H], [A][256] */
    int64_t ln2_g, ln2_b;     /* linear.3                [A][256]   -- the group {ln2_g, ln2_b, w2, b2} is the LAST one of */
    int64_t w2, b2;           /* linear.4                [A][256][256], [A][256]   the buffer: [ln2_g, total) (see as_opts) */
    int64_t ln3_g, ln3_b;     /* linear.6                [A][256] */
    int64_t w3, b3;           /* x_coords then y_coords  [A][2][N][256], [A][2][N] */
    int64_t total;            /* number of floats in the flat buffer (multiple of 64) */
} as_layout;

int as_artspeech_layout(const as_dims* dims, as_layout* out);

/* ------------------------------------------------------------------------------------------------
 * Workspace: one caller-allocated float buffer holds every intermediate and saved activation of a
 * training step for a (B, T) batch.  as_artspeech_workspace_floats() returns its size in floats.
 *
 * Token ids: the reference's nn.Embedding (models.py:135) raises IndexError for an id outside [0, V).  The kernels
 * never sync, so instead every kernel that indexes a table by token id clamps the id into the table (no out-of-range
 * access whatever the input), and as_artspeech_fwd leaves the NUMBER of out-of-range ids of its batch in the first
 * 32-bit word of the workspace (int32): the host raises when it reads a non-zero count (the drop-in modules read it
 * after every forward; the training engine whenever it reads the loss).
 * ---------------------------------------------------------------------------------------------- */
int64_t as_artspeech_workspace_floats(const as_dims* dims, int32_t B, int32_t T);

/* Per-call training options (NULL = none).  gru_dropout: nn.GRU's inter-layer dropout (models.py:111; applied
 * to the layer-0 output in training mode only).  The mask is a pure function of (dropout_seed, element): pass the
 * SAME options to the matching as_artspeech_bwd.  Statistical, not bitwise, parity with torch's generator. */
typedef struct as_opts {
    float gru_dropout;      /* 0 <= p < 1: nn.GRU inter-layer dropout (ArtSpeech, models.py:111) or, with dims->simple, the
                               nn.Dropout on the embedded frames of SimpleArtSpeech (models.py:64,85) */
    uint64_t dropout_seed;
    /* as_artspeech_bwd only: non-zero => `dout` already holds the gradient w.r.t. the PRE-sigmoid activations (written by
       as_euclid_masked_fwd_bwd_presigmoid), so the separate dout * out * (1 - out) pass over the contours is skipped. */
    int32_t dout_presigmoid;
    /* Software pipelining of the training loop across the step boundary (ArtSpeech only; artspeech_amd/engine.py):
       as_artspeech_bwd with defer_dw2 != 0 leaves out the weight gradient of the heads' SECOND Linear (9.2 of the step's
       22.9 GFLOP of weight gradients: grads [ln2_g, total) = the LAST group of the flat layout stays unwritten) and
       as_artspeech_dw2() computes it later from the same workspace -- the engine runs it on a side stream beside the NEXT
       step's forward recurrences (64 of 256 CUs busy), followed by that slice's all-reduce and Adam update.
       as_artspeech_fwd with fold_wait_event != NULL (a hipEvent_t) makes the stream that folds the head weights wait for
       that event first (the deferred update must be in place before the heads read their parameters); the recurrences do
       not wait.  The event must stand BEHIND the previous optimizer step on `stream` (the engine enqueues the deferred update
       after a wait for `stream`): the forward then relies on it alone to order its weight folds.  Same arithmetic on the same operands as the unpipelined step: bit-identical parameters. */
    int32_t defer_dw2;
    void* fold_wait_event;
    /* as_artspeech_fwd (training engine): the criterion of train_phoneme_to_articulation.py:86-90 fused into the epilogue of
       the heads' output layer.  loss_targets != NULL (float [B][loss_tgt_T][A][2][N] on the device, loss_tgt_T >= T): the
       forward also writes *loss_out = loss_scale * sum over valid frames of the point distances (loss_scale = 1 / (valid
       frames * A * N); a device scalar) and loss_dout [B][T][A][2][N] = d loss / d(pre-sigmoid activations), which
       as_artspeech_bwd takes as `dout` with dout_presigmoid set -- the separate pass of as_euclid_masked_fwd_bwd_presigmoid
       over the 28 MB of contours and targets disappears.  `out` is written as usual.  Needs 2 N <= 128. */
    const float* loss_targets; int64_t loss_tgt_T; float loss_scale; float* loss_out; float* loss_dout;
} as_opts;

/* ArtSpeech.forward / SimpleArtSpeech.forward (encoder_decoder/models.py:126-145, 75-96).
 *   tokens  : int64 [B][tok_stride] phoneme indices (only the first T columns are read)
 *   lengths : int32 [B] on the DEVICE, sorted descending, 1 <= len <= T, T == max(lengths)
 *             (ignored when dims->simple)
 *   out     : float [B][T][A][2][N]  sigmoid contours
 *   train   : non-zero => keep what as_artspeech_bwd needs in `ws`
 */
int as_artspeech_fwd(const as_dims* dims, const float* params, const int64_t* tokens, int64_t tok_stride,
                     const int32_t* lengths, int32_t B, int32_t T, float* out, float* ws, int32_t train,
                     const as_opts* opts, void* stream);

/* Backward of the above: d(out) -> gradient of EVERY parameter, written (not accumulated) into the
 * flat `grads` buffer (same layout as params).  Must follow as_artspeech_fwd(train=1) on the same ws. */
int as_artspeech_bwd(const as_dims* dims, const float* params, const int64_t* tokens, int64_t tok_stride,
                     const int32_t* lengths, int32_t B, int32_t T, const float* out, const float* dout,
                     float* grads, float* ws, const as_opts* opts, void* stream);

/* The part of the backward that as_artspeech_bwd(opts->defer_dw2) left out: weight / bias gradient of the heads' second Linear
 * and of the LayerNorm in front of it -> grads [ln2_g, total).  Reads what forward + backward left in `ws` (valid until the
 * next as_artspeech_fwd on that workspace reaches its heads, i.e. until that call's fold_wait_event is signalled). */
int as_artspeech_dw2(const as_dims* dims, const float* params, int32_t B, int32_t T, float* grads, float* ws, void* stream);

/* as_artspeech_bwd runs the weight-gradient GEMMs on a library-owned side stream beside the GRU backward
 * recurrences (fork/join by stream-ordered events; `stream` observes completion of everything on return
 * order).  The side stream and its events belong to the (device, `stream`) pair: callers on different streams never
 * share them.  They are created on the first call for that pair, never inside a stream capture (run one step before
 * capturing).  as_set_overlap(0) keeps every kernel on `stream` (default on; env ARTSPEECH_NO_OVERLAP=1 = off). */
void as_set_overlap(int32_t on);

/* Streaming copy dst[i] = src[i] of n floats (n % 4 == 0, 16-byte aligned): a float4 grid-stride kernel.  bench.py times it on
 * 1 GiB to quote the box's measured HBM copy rate (read + write bytes over time) beside the 8 TB/s specification. */
int as_copy_f32(const float* src, float* dst, int64_t n, void* stream);

/* How the library forms fp32 matrix products (every nn.Linear / weight gradient of the path; the reference computes them
 * in fp32: encoder_decoder/models.py:10-33, transformer/models.py:37-100):
 *   0  v_mfma_f32_32x32x2_f32 on the fp32 operands (the exact fp32 matrix instruction);
 *   1  (default) each fp32 operand is split EXACTLY into three bfloat16 numbers (8 + 8 + 8 significand bits) and the
 *      product is rebuilt from six of the nine plane products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the
 *      three dropped ones are each <= 2^-24 |a||b|.  Inputs, outputs and accumulation stay fp32; against an fp64 product
 *      of the same operands the result is at least as accurate as mode 0 (tests/test_gpu_parity.py::test_split_arith_*).
 *      Kernels without such a path use mode 0.  Inf operands give NaN (inf - inf in the split) where mode 0 gives inf.
 * Process-wide; read at launch time.  Env ARTSPEECH_MATRIX_ARITH=fp32|bf16x6 sets the initial value. */
void as_set_matrix_arith(int32_t mode);
int32_t as_get_matrix_arith(void);

/* Data-parallel hook: make `waiting_stream` wait until the most recent as_artspeech_bwd enqueued on `compute_stream` (same
 * device) has finished the gradients of the trunk Linear and of all heads, i.e. the tail [layout.lin_w, layout.total) of
 * the flat gradient buffer (74 % of the parameters) -- their all-reduce can then run while the GRU backward recurrences
 * are still going. */
int as_artspeech_wait_head_grads(void* compute_stream, void* waiting_stream);

/* ------------------------------------------------------------------------------------------------
 * Building blocks (each is also used by the composite entry points above)
 * ---------------------------------------------------------------------------------------------- */

/* Bidirectional GRU layer recurrence on a packed batch (nn.GRU at encoder_decoder/models.py:111,137;
 * gate order r,z,n; h0 = 0; reverse direction walks t = len_b-1..0; padded outputs are zeros).
 *   gi      : input projections W_ih x + b_ih.  If tokens != NULL it is a TABLE [V][2][3H] indexed by
 *             tokens[b*tok_stride + t] (embedding folded in); else it is [B*T][2][3H].
 *   w_hh    : [2][3H][H], b_hh : [2][3H]
 *   y       : [B][T][2H]  (forward direction in [:H], reverse in [H:])
 *   gates   : NULL (inference) or [B][T][2][4][H] receiving r, z, n, (W_hn h + b_hn) for the backward
 *   H       : 32, 64, 128 run register-resident kernels (W_hh stays in VGPRs for the whole sequence); any other multiple of 4
 *             runs plain kernels that stream W_hh from L2 every step (same results up to the summation order, several
 *             times slower per step) */
int as_gru_bidir_fwd(const float* gi, const int64_t* tokens, int64_t tok_stride, const float* w_hh,
                     const float* b_hh, const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* y,
                     float* gates, void* stream);

/* BPTT of the recurrence.  dy [B][T][2H] -> dgi, dgh [B*T][2][3H] (gradients w.r.t. the input- and
 * hidden-projection pre-activations, zeros at padded frames); weight gradients are time-batched GEMMs
 * over these (done by as_artspeech_bwd). */
int as_gru_bidir_bwd(const float* dy, const float* y, const float* gates, const float* w_hh,
                     const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* dgi, float* dgh,
                     void* stream);

/* Bidirectional LSTM layer, forward / backward (nn.LSTM semantics; the RNNType.LSTM switch of
 * phoneme_to_articulation/__init__.py:47-49 used by principal_components/models/rnn.py:58-68).  Packed-sequence
 * semantics as as_gru_bidir_fwd; gate row order i, f, g, o; h0 = c0 = 0.
 *   H     : 32, 64, 128 run register-resident kernels; any other multiple of 4 (up to 1636 with a backward) the plain
 *           kernels (W_hh streamed from L2 every step; several times slower per step), else AS_ERR_UNSUPPORTED
 *   gi    : W_ih x + b_ih, [B*T][2][4H], or (tokens != NULL) a table [V][2][4H] indexed by tokens[b*tok_stride + t]
 *   w_hh  : [2][4H][H], b_hh : [2][4H];  y : [B][T][2H]
 *   gates : NULL (inference) or [B][T][2][5][H] receiving i, f, g, o, c for the backward
 * Backward: dy [B][T][2H] -> dg [B][T][2][4H], the gradient of the gate pre-activations (input- and hidden-side
 * pre-activations share it; zeros at padded frames); weight gradients are time-batched GEMMs over it. */
int as_lstm_bidir_fwd(const float* gi, const int64_t* tokens, int64_t tok_stride, const float* w_hh, const float* b_hh,
                      const int32_t* lengths, int32_t B, int32_t T, int32_t H, float* y, float* gates, void* stream);
int as_lstm_bidir_bwd(const float* dy, const float* gates, const float* w_hh, const int32_t* lengths, int32_t B, int32_t T,
                      int32_t H, float* dg, void* stream);

/* Strided-batched fp32 GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32):
 *   C[g][i][j] (+)= act( sum_k Aop[g][i][k] * Bop[g][j][k] + bias[g][j] )
 * Aop[i][k] = A[i*a_i + k*a_k] and Bop[j][k] = B[j*b_j + k*b_k]; one stride of each pair must be 1.
 * act: 0 none, 1 ReLU, 2 sigmoid, 3 exact (erf) GELU.  accumulate != 0 adds to C.  b_kshift/b_kT: if b_kT > 0 the B operand's
 * reduction index k is read at k + b_kshift and is zero unless 0 <= (k % b_kT) + b_kshift < b_kT
 * (the h_{t-1} operand of dW_hh); batch g uses the shift b_kshift + g * b_kshift_batch (forward and reverse direction
 * of a bidirectional layer in one batched call: -1 and +1). */
typedef struct as_gemm {
    const float* A; const float* B; float* C; const float* bias;
    int32_t M, N, K;
    int64_t a_i, a_k, b_j, b_k, ldc;
    int32_t batch; int64_t a_batch, b_batch, c_batch, bias_batch;
    int32_t act, accumulate;
    int32_t b_kshift, b_kT;
    /* optional: workspace enabling deterministic split-K for long reductions with few output tiles
       (weight gradients); optional fused column sums of the A operand, colsum[g][i] = sum_k Aop[g][i][k]
       (bias gradients; needs a_i == 1). */
    float* splitk_ws; int64_t splitk_ws_floats;
    float* colsum; int64_t colsum_batch;
    /* optional GROUPED batches: device arrays of `batch` element offsets that replace g*x_batch for the
       corresponding operand (NULL = linear stride).  Offsets must be multiples of 4 floats.  Used by the
       multi-channel transformer decoder, whose 132 attention blocks per layer read/write irregular
       (channel, head) slices. */
    const int64_t* a_off; const int64_t* b_off; const int64_t* c_off; const int64_t* bias_off;
    /* 0: exact fp32 on the f32 MFMA (default).  1 / 2: forward linears (a_k == b_k == 1, float4-clean operands) may run on
       the bf16 MFMA with every fp32 element split on the fly into 2 / 3 bf16 pieces and the product rebuilt from 3 / 6
       cross terms with fp32 accumulation (error ~2^-16 / ~2^-23 of sum |a||b|); other shapes silently stay exact.
       3: the library's current matrix arithmetic (as_set_matrix_arith): in mode 1 the shapes gemm_s6.hip's kernel takes run on
       the bf16 matrix instruction with both operands split exactly into three planes inside the kernel, whatever their size --
       forward (a_k == b_k == 1; plain or relu_bits epilogue), input gradient (a_k == 1, b_j == 1; with res / mask_bits / k_seg)
       and weight gradient (a_i == b_j == 1, >= 256 output tiles of 128 x 128; with colsum); linear or grouped batches,
       K % 16 == 0, float4-clean operands.  Anything else, and mode 0, as 0.  The transformer modules use it for every backward
       GEMM (ARTSPEECH_GRAD_PRECISION, default "lib": 218 -> 187 ms at configs[3]) and offer it for the forward ones
       (ARTSPEECH_GEMM_PRECISION=lib: 173 ms; the full-width contours then sit at 1.19 x the 1e-4 bound against the reference
       fixture, 0.89 x with 0: default 0). */
    int32_t precision;
    int32_t b_kshift_batch;
    /* optional hint for weight-gradient shapes (a_i == b_j == 1 with splitk_ws): the number of CUs the launch can expect to
       have (0 = the whole chip), e.g. 192 while a 64-workgroup recurrence kernel runs beside it on another stream; only the
       split-K factor depends on it, never the result's summation order for a given factor. */
    int32_t cu_budget;
    /* optional extra operands of the general kernel (forward / input-gradient shapes with a reduction-contiguous A; refused
       together with colsum, splitk_ws, accumulate, split precision and weight-gradient shapes):
           C = keep ? act(res + acc + bias) : 0
       res       [.][M][N] through (res_ld, res_batch | res_off): the INITIAL VALUE of the accumulators (loaded while the
                 first operand tiles are staged: no extra pass, no epilogue loads); the sum is (res + sum_k in k order) +
                 bias.  In the backward of a ChannelProcessingLayer (transformer/models.py:98) the gradient that arrives
                 over the residual `q + out_proj(ctx)` joins the in-projection's input gradient this way.  Every partial sum
                 carries the residual's magnitude, so the product's rounding error grows with |res| / |sum|: meant for
                 gradients; the forward's residual is added by the LayerNorm that consumes it (as_layernorm_fwd_blockres).
       relu_bits out, with act == 1: one bit per output element, [.][M][ceil(N / 32)] 32-bit words (bit n % 32 of word
                 n / 32 of a row: result > 0), batch stride relu_bits_batch words -- what the backward of the ReLU needs;
       mask_bits in: such a bit image; elements whose bit is clear are stored as 0 (dz = dy * [y > 0], models.py:47-60
                 through autograd): the ReLU backward rides in the epilogue of the GEMM that produces dy, reading 1/32 of
                 what a pass over the saved activations would. */
    const float* res; int64_t res_ld, res_batch; const int64_t* res_off;
    const uint32_t* mask_bits; int64_t mask_batch;
    uint32_t* relu_bits; int64_t relu_bits_batch;
    /* optional SEGMENTED reduction (a_k == 1 input-gradient / forward shapes of the general kernel): the reduction index is
       cut into K / k_seg segments of k_seg (a multiple of 32) and segment s of batch member g reads its A rows from
       A + a_seg_off[g * nseg + s] and its B panel from B + b_seg_off[g * nseg + s] (element offsets, multiples of 4; the
       batch strides / offset tables of A and B are then ignored; inside a segment the ordinary strides apply with k counted
       from the segment's start).  Sums the input gradients of all blocks that read one source channel in ONE GEMM
       (dx[c] = sum_g dz[g] W[g] over the blocks g with src[g] = c) instead of per-block partial tensors + a reduce pass. */
    int32_t k_seg; const int64_t* a_seg_off; const int64_t* b_seg_off;
    /* optional: the A operand is EXACTLY zero below (1: A[i][k] == 0 for k < i) or above (2: for k > i) its diagonal, i and k
       counted in the same index space -- the key-major probabilities P^T[key][q] of a causally masked attention and their
       gradients (transformer/models.py:380-387: both decoder masks are causal).  An output tile then only walks the k-tiles
       in which one of its rows can be non-zero; the skipped products are exact zeros (a non-finite B element in a skipped
       range no longer turns 0 * inf into NaN).  General kernel only, like k_seg; a HINT: operands that are not float4-clean
       (16-byte aligned, strides and contiguous extents multiples of 4) are multiplied over the full range, same result. */
    int32_t k_tri;
} as_gemm;
int as_gemm_f32(const as_gemm* g, void* stream);

/* nn.Linear forward, out[M][N] = act(A[M][K] . W[N][K]^T + bias) (act: 0 none, 1 ReLU, 2 sigmoid; the Linear layers of
 * encoder_decoder/models.py:12-16, 113-116 and transformer/models.py:47-60), in the library's current matrix arithmetic
 * (as_set_matrix_arith): mode 1 emits W as three bfloat16 planes into `planes_ws` (>= as_linear_planes_floats(N, K) floats of
 * scratch, 16-byte aligned) and multiplies on v_mfma_f32_32x32x16_bf16 where the shape allows (K % 32 == 0; N <= 256 or
 * N % 256 == 0; lda % 4 == 0); planes_ws == NULL: both operands are split inside the kernel instead (128 x 128 tiles, K % 16
 * == 0, lda % 4 == 0, ldw % 4 == 0: the form the GRU input projection and the trunk Linear use); everything else -- and mode 0
 * -- runs as_gemm_f32.  tests/ hold both modes' error against an fp64
 * product of the same operands. */
int64_t as_linear_planes_floats(int32_t N, int32_t K);
int as_linear_fwd(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* out, int64_t ldo,
                  int32_t M, int32_t N, int32_t K, int32_t act, float* planes_ws, void* stream);

/* ArticulatorPredictor x A + stack + sigmoid (encoder_decoder/models.py:7-33, 141-145) on rows of
 * `x` [rows][in]: out [rows][A][2][N].  Parameters are read from the flat buffer through `lay`.
 * ws: as_head_workspace_floats(). */
int64_t as_head_workspace_floats(const as_dims* dims, int64_t rows);
int as_head_fwd(const as_dims* dims, const as_layout* lay, const float* params, const float* x, int64_t rows,
                float* out, float* ws, int32_t train, void* stream);
/* dout [rows][A][2][N] -> dx [rows][in] and the head parameter gradients inside `grads` (flat). */
int as_head_bwd(const as_dims* dims, const as_layout* lay, const float* params, const float* out,
                const float* dout, int64_t rows, float* dx, float* grads, float* ws, void* stream);

/* EuclideanDistance (phoneme_to_articulation/metrics.py:5-24), reduction "none":
 * out, tgt [frames][A][2][N] -> dist [frames][A][N]. */
int as_euclid_fwd(const float* out, const float* tgt, int64_t frames, int32_t A, int32_t N, float* dist,
                  void* stream);
/* d dist [frames][A][N] -> d out [frames][A][2][N] (NaN where dist == 0, like the reference). */
int as_euclid_bwd(const float* out, const float* tgt, const float* ddist, int64_t frames, int32_t A, int32_t N,
                  float* dout, void* stream);

/* The masked mean of train_phoneme_to_articulation.py:86-90 fused with EuclideanDistance and its
 * gradient: loss = scale * sum_{b, t < len_b} sum_{a,n} dist ; dout = d loss / d out (0 at padded frames).
 * scale = 1 / (N_valid * A * N) with N_valid the GLOBAL number of valid frames (so that DP shards sum
 * to the reference's mean).  out/tgt [B][T or tgt_T][A][2][N]; loss: one float, overwritten;
 * partial: workspace of as_euclid_masked_partials() floats; dout may be NULL (evaluation). */
int32_t as_euclid_masked_partials(void);
int as_euclid_masked_fwd_bwd(const float* out, const float* tgt, int64_t tgt_T, const int32_t* lengths, int32_t B,
                             int32_t T, int32_t A, int32_t N, float scale, float* loss, float* dout,
                             float* partial, void* stream);
/* Same, but `dout` receives d(loss)/d(pre-sigmoid activation) = dout * out * (1 - out): the criterion's backward and the
 * model's final sigmoid backward (models.py:145) in one pass over the contours; pair with as_opts.dout_presigmoid. */
int as_euclid_masked_fwd_bwd_presigmoid(const float* out, const float* tgt, int64_t tgt_T, const int32_t* lengths, int32_t B,
                             int32_t T, int32_t A, int32_t N, float scale, float* loss, float* dout,
                             float* partial, void* stream);

/* MeanP2CPDistance, reduction "none" (phoneme_to_articulation/metrics.py:27-46): for each of `tiles`
 * independent (u, v) pairs: 0.5 * (mean_i min_j |u_i - v_j| + mean_j min_i |u_i - v_j|).
 * Point p of tile t, coordinate c is at u[t*u_tile + p*u_pt + c*u_xy] (same for v): covers both the
 * (*, N, 2) layout the module receives and the (*, 2, N) storage it is a transposed view of.
 * Distances by direct differences (more accurate than torch.cdist's matmul expansion). */
int as_p2cp_fwd(const float* u, int64_t u_tile, int64_t u_pt, int64_t u_xy, int32_t n_u, const float* v,
                int64_t v_tile, int64_t v_pt, int64_t v_xy, int32_t n_v, int64_t tiles, float* out, void* stream);

/* P2CPDistance.forward (encoder_decoder/metrics.py:18-26) reduction: p2cp [B][T][A], lengths ->
 * result[0] = mean_b( mean_{t < len_b, a} p2cp * to_mm ). */
int as_p2cp_utterance_mean(const float* p2cp, const int32_t* lengths, int32_t B, int32_t T, int32_t A, float to_mm,
                           float* result, void* stream);

/* pearsons_correlation (root metrics.py:9-35): Pearson correlation over time of every (utterance, articulator, point)
 * column, x and y planes separately.  out / tgt: [B][T][A][2][N] with element strides out_b / out_t (tgt_b / tgt_t)
 * for the utterance and frame index, the [A][2][N] block of a frame contiguous.  x_corr, y_corr: [B][A][N].
 * corr = sum(vo * vt) / (sqrt(sum vo^2) * sqrt(sum vt^2) + eps); as in the reference the x TARGETS are centred with
 * the x OUTPUTS' mean (metrics.py:22), the y targets with their own (metrics.py:30). */
int as_pearson_fwd(const float* out, int64_t out_b, int64_t out_t, const float* tgt, int64_t tgt_b, int64_t tgt_t,
                   int32_t B, int32_t T, int32_t A, int32_t N, float eps, float* x_corr, float* y_corr, void* stream);

/* Tract variables (tract_variables.py:23-35, 73-125): for each frame and each of n_tv variables the
 * minimum pairwise distance between two point sets and the two closest points (first minimum over
 * arr1 for each arr2 point, then first minimum over arr2).  A set is up to two (channel, start, end)
 * segments of the frame's contours [A][2][N] concatenated.
 *   spec: int32 [n_tv][3 segments][3] = {channel, start, end} for arr1, arr2 part 1, arr2 part 2
 *         (channel < 0 => segment absent), on the DEVICE.
 *   values [frames][n_tv], poc1/poc2 [frames][n_tv][2], idx int32 [frames][n_tv][2] (may be NULL).
 * Limits: N <= 64 points per contour (a lane owns one arr2 point; the reference's slices hold 15 .. 50, n_samples = 50) and
 * a frame [A][2][N] of at most ~15 K floats (it is staged in LDS once): AS_ERR_UNSUPPORTED beyond. */
int as_tract_variables_fwd(const float* contours, int64_t frames, int32_t A, int32_t N, const int32_t* spec,
                           int32_t n_tv, float* values, float* poc1, float* poc2, int32_t* idx, void* stream);

/* area_function (area_function.py:113-142), float64.  Wall point p, coordinate c of frame f is at
 * w[f*frame_stride + p*pt_stride + c*xy_stride].  dists, fx: [frames][n_pts].  dists is the
 * sequential running sum of mid-line segment lengths (same order as the reference's loop).
 * 1 <= n_pts <= 1024 (the reference's air columns hold 100 points per wall). */
int as_area_function_fwd(const double* internal_wall, const double* external_wall, int64_t frame_stride,
                         int64_t pt_stride, int64_t xy_stride, int64_t frames, int32_t n_pts, double alpha,
                         double beta, double* dists, double* fx, void* stream);

/* ---- row operators used by the transformer variant (transformer/models.py) ---------------------------
 * LayerNorm over the last dim (eps 1e-5, biased variance): y = LN(x + res) * gamma + beta.
 * res, gamma/beta (NULL = none / affine-free), xhat and rstd (NULL = not kept) are optional.
 * gamma/beta: group_rows > 0: `group_rows` consecutive rows per parameter set (group g uses gamma + g*D);
 * group_rows < 0: parameter set = row % (-group_rows) (interleaved channels); 0 = one set. */
int as_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* xhat,
                     float* rstd, int64_t rows, int32_t D, int64_t group_rows, void* stream);

/* Affine-free LayerNorm of x + res where x [channels][rows][per * block] is the concatenation of `per` blocks per channel and
 * res [channels * per][rows][block] is block-major:  xhat[c][r][j*block + f] = LN_row(x[c][r][:] + res[c*per + j][r][f]).
 * This is `LayerNorm(cat_j(q_j + out_proj_j(ctx_j)))` of ChannelInteractionsLayer (transformer/models.py:98, 133-162) with x the
 * out-projections written into the concatenated layout and res the projected queries where the blocks left them: the
 * residual add costs no pass of its own and keeps the reference's order, (sum + bias) + q.  rstd [channels * rows] optional. */
int as_layernorm_fwd_blockres(const float* x, const float* res, float* xhat, float* rstd, int32_t channels, int64_t rows,
                              int32_t per, int32_t block, void* stream);

/* Fold a LayerNorm affine into the following Linear for `heads` independent (W [R][K], gamma/beta [K], b [R])
 * sets: Wf = W.diag(gamma), bf = b + W.beta (then Linear(LN(x)) == x_hat . Wf^T + bf). */
int as_fold_ln(const float* W, const float* gamma, const float* beta, const float* b, float* Wf, float* bf, int32_t heads,
               int32_t R, int32_t K, void* stream);

/* In-place masked softmax over the last dim of scores [Z][Tq][Tk] (nn.MultiheadAttention semantics with float
 * masks): p = softmax(s * scale + attn_mask[b][q][k] + key_padding_mask[b][k]), b = (z / heads) % B.
 * attn_mask / key_padding_mask may be NULL.  A fully masked row gives NaN, as in PyTorch. */
int as_attn_softmax(float* scores, int64_t Z, int32_t Tq, int32_t Tk, int32_t heads, int32_t B, float scale,
                    const float* attn_mask, const float* key_padding_mask, void* stream);

/* Backward of as_attn_softmax, in place on dprobs: dS = P * (dP - sum_k dP * P) * scale. */
int as_attn_softmax_bwd(const float* probs, float* dprobs, int64_t Z, int32_t Tq, int32_t Tk, float scale, void* stream);

/* Fused attention core of nn.MultiheadAttention as used by ChannelProcessingLayer (transformer/models.py:37-100), on already
 * projected tensors:  out[g][b*T + q][h*dh + c] = sum_k softmax_k(Q_h[q] . K_h[k] * scale + attn_mask[b][q][k] + kpm[b][k]) V_h[k][c]
 *   Q, out : [G][B*T][d]     K, V : [G][B*Tk][d]     dh = d / heads in {16, 32, 64}, Tk <= 256 (as_attention_supported)
 *   attn_mask_t      : [B][Tk32][T] additive float mask, KEY-major (the transpose of the reference's (B, T, Tk) mask) with
 *                      the key dimension padded to Tk32 = Tk rounded up to a multiple of 32 (finite padding), or NULL
 *   key_padding_mask : [B][Tk] additive float mask (0 / -inf) or NULL
 *   lse              : optional [G*B*heads][T] log-sum-exp of every score row (what a recomputing backward needs), or NULL
 *   probs_t          : optional [G*B*heads][Tk][Tp] the probabilities, KEY-major, rows padded to Tp = T rounded up to a multiple
 *                      of 32 floats (columns >= T are not written; a 32-query segment of a row is then one 128-byte line)
 *                      (training: consumed by the grouped GEMMs of the backward and by as_attn_softmax_bwd_t), or NULL
 *                      (inference: nothing but `out` is written)
 * Scores and probabilities stay in registers (the unfused path writes G*B*heads*T*Tk floats twice).  A fully masked query
 * row yields NaN, as in PyTorch. */
int as_attention_supported(int32_t T, int32_t Tk, int32_t d, int32_t heads);
int as_attention_fwd(const float* Q, const float* K, const float* V, const float* attn_mask_t, const float* key_padding_mask,
                     float* out, float* lse, float* probs_t, int32_t G, int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d,
                     float scale, void* stream);
/* The same for a mask the CALLER has checked to be -inf above the diagonal for every utterance (attn_mask[b][q][k] == -inf for
 * k > q: both decoder masks of transformer/models.py:380-387): the key blocks beyond a 32-query strip's own are not computed
 * (their probabilities are exact zeros either way) and the strips are dealt over the waves so that every SIMD gets the same
 * number of blocks.  Identical results. */
int as_attention_fwd_causal(const float* Q, const float* K, const float* V, const float* attn_mask_t, const float* key_padding_mask,
                            float* out, float* lse, float* probs_t, int32_t G, int32_t B, int32_t heads, int32_t T, int32_t Tk,
                            int32_t d, float scale, void* stream);

/* Softmax backward on KEY-major tensors, in place on dprobs_t:  dS^T[k][q] = P^T[k][q] * (dP^T[k][q] - D[q]) * scale with
 * D[q] = sum_c dctx[q][c] * ctx[q][c] over the head's dh columns (= sum_k P dP, without a second pass over the scores).
 *   probs_t, dprobs_t : [G*B*heads][Tk][Tp], Tp = T rounded up to 32      ctx, dctx : [G][B*T][d] (attention output and its gradient)
 *   dsum              : workspace of G*B*heads*T floats (receives D); dh = d / heads in {16, 32, 64} */
int as_attn_softmax_bwd_t(const float* probs_t, float* dprobs_t, const float* ctx, const float* dctx, float* dsum, int32_t G,
                          int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d, float scale, void* stream);

/* Backward of the attention core, first half, without a dP tensor:
 *   ds_t[z][k][q] = probs_t[z][k][q] * (sum_c V_h[k][c] dctx[q][h*dh + c] - D[q]) * scale,   D[q] = sum_c dctx[q][c] ctx[q][c]
 * (= as_attn_softmax_bwd_t applied to dP^T = V dctx^T).  Shapes as as_attention_fwd; ds_t [G*B*heads][Tk][Tp] is what the
 * two remaining grouped GEMMs (dQ = dS K, dK = dS^T Q) read. */
int as_attention_bwd_ds(const float* V, const float* dctx, const float* ctx, const float* probs_t, float* ds_t, int32_t G, int32_t B,
                        int32_t heads, int32_t T, int32_t Tk, int32_t d, float scale, void* stream);
/* The same when the forward's additive mask was -inf above the diagonal for every utterance (both decoder masks of
 * transformer/models.py:380-387 are causal), i.e. probs_t[z][key][q] == 0 for q < key: only the (32-query strip, 32-key block)
 * pairs on or below the diagonal are computed -- dealt evenly over the workgroup's waves -- the others are stored as zeros.  Same
 * values as as_attention_bwd_ds where P^T is non-zero; exact +0 elsewhere.  T, Tk <= 256 (else the general kernel runs). */
int as_attention_bwd_ds_causal(const float* V, const float* dctx, const float* ctx, const float* probs_t, float* ds_t, int32_t G,
                               int32_t B, int32_t heads, int32_t T, int32_t Tk, int32_t d, float scale, void* stream);

/* dst[c][:] = sum over groups g with src[g] == c of part[g][:] (rows of `len` floats; deterministic order): folds
 * the per-block input gradients of a grouped GEMM back onto the channels the blocks read. */
int as_group_reduce(const float* part, const int32_t* src, int32_t G, int32_t C, int64_t len, float* dst, void* stream);

/* Backward of the affine-free LayerNorm: dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)), optionally
 * times the ReLU mask (relu_src > 0).  dxhat and dx may alias.  Rows up to 512 wide (wider: D <= 2816). */
int as_layernorm_bwd(const float* dxhat, const float* xhat, const float* rstd, const float* relu_src, float* dx, int64_t rows,
                     int32_t D, void* stream);

/* Backward of as_fold_ln: (dWf, dbf) -> dW = dWf.diag(gamma) + dbf beta^T, dgamma, dbeta (db == dbf). */
int as_unfold_ln(const float* dWf, const float* dbf, const float* W, const float* gamma, const float* beta, float* dW,
                 float* dgamma, float* dbeta, int32_t heads, int32_t R, int32_t K, void* stream);

/* dst[i] = act[i] > 0 ? g[i] : 0 */
int as_relu_bwd(const float* g, const float* act, float* dst, int64_t n, void* stream);

/* out[m][:] = table[tokens[m]][:] + pe[m % T][:]  (Embedding + PositionalEncoding, transformer/models.py:9-34,368-369);
 * table == NULL: out[m][:] += pe[m % T][:] in place. */
int as_embed_posenc(const int64_t* tokens, int64_t tok_stride, const float* table, const float* pe, float* out,
                    int64_t rows, int32_t T, int32_t D, void* stream);

/* dst[i] = a[i] + b[i] (b may be NULL: copy) and dst[i] = a[i] * m[i / row_len] (row mask) */
int as_add(const float* a, const float* b, float* dst, int64_t n, void* stream);
int as_row_scale(const float* a, const float* row_scale, float* dst, int64_t rows, int32_t row_len, void* stream);

/* Inverted dropout with the library's counter-based mask: y[i] = x[i] * keep(seed, i) / (1 - p); x == y allowed.
 * The same (p, seed) regenerates the same mask (that is how the backward works). */
int as_dropout_fwd(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);

/* evenly_spaced_fx (area_function.py:145-159): resample fx over x (both [frames][n_pts], float64, x increasing) at
 * n_samples evenly spaced abscissae between x[0] and x[-1]; out float32 [frames][2][n_samples] = (abscissae, values),
 * i.e. the reference's vertical-line / polyline intersections = piecewise-linear interpolation. */
int as_evenly_spaced_fx(const double* x, const double* fx, int64_t frames, int32_t n_pts, int32_t n_samples, float* out,
                        void* stream);

/* ---- DeepSpeech2-style articulatory scorer, inference (phoneme_recognition/deepspeech2.py:90-195) -------------
 * Feature maps are channels-last  [B][T][D][32]  (the reference's (B, 32, D, T) permuted (0, 3, 2, 1)).
 * as_conv3x3_stem : nn.Conv2d(Cin, 32, 3, stride 1, padding 1) (:104) of a planar input whose element (b, ci, d, t)
 *                   sits at x[b*sb + ci*sc + d*sd + t*st]; w [9][32][Cin] with tap = kd*3 + kt (the torch weight
 *                   permuted (2, 3, 0, 1)); voicing (NULL or [B][T]) is added to every channel/feature (:175-177).
 * as_conv3x3_c32  : the 32 -> 32 convolutions of ResidualCNN (:22, 25) as an implicit GEMM on the f32 MFMA;
 *                   w [9][32][32]; res (NULL or [B][T][D][32]) is the block's skip input (`out += x`, :46). */
int as_conv3x3_stem(const float* x, int64_t sb, int64_t sc, int64_t sd, int64_t st, const float* w, const float* bias,
                    const float* voicing, float* y, int32_t B, int32_t T, int32_t D, int32_t Cin, void* stream);
int as_conv3x3_c32(const float* x, const float* w, const float* bias, const float* res, float* y, int32_t B, int32_t T,
                   int32_t D, void* stream);

/* ResidualCNN's transpose -> LayerNorm(num_features) -> transpose -> GELU (:30-36, 40-44) on x [rows][D][C]:
 * y[r][d][c] = gelu(LN over d of x[r][:][c], affine gamma[d], beta[d]); eps 1e-5; x == y allowed. */
int as_ln_feat_gelu(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t D, int32_t C,
                    void* stream);

/* y = gelu(x) (exact erf form, F.gelu default; RecurrentBlock :66); x == y allowed. */
int as_gelu(const float* x, float* y, int64_t n, void* stream);

/* nn.GRU(num_layers=1, bidirectional=False) forward over full-length or ragged rows (RecurrentBlock :54-60, 67), h0 = 0:
 * gi [B][T][3H] = W_ih x + b_ih, w_hh [3H][H], b_hh [3H], y [B][T][H] (zeros at t >= lengths[b]). */
int as_gru_unidir_fwd(const float* gi, const float* w_hh, const float* b_hh, const int32_t* lengths, int32_t B, int32_t T,
                      int32_t H, float* y, void* stream);

/* intersect_semipolar_grid (area_function.py:175-223), float64, batched over frames: air_column [frames][2 walls][2][n_pts]
 * (internal wall first, x row then y row -- the air-column file layout), grid [n_lines][grid_res][2].  For every frame and
 * grid line: flags bit 0 / 1 = the internal / external wall is crossed (0: the reference skips the line), bit 2 = more than
 * 16 crossings of one wall (extra ones dropped); p_int / p_ext [frames][n_lines][2] = the selected point on each wall (the
 * wall's own end point when that wall is not crossed).  Intersections are float64 segment tests ordered along the grid
 * line; the reference computes them with shapely/GEOS (parity unpinned). */
int as_intersect_semipolar_grid(const double* air_column, const double* grid, int64_t frames, int32_t n_pts, int32_t n_lines,
                                int32_t grid_res, int32_t* flags, double* p_int, double* p_ext, void* stream);

/* torch.optim.Adam semantics (L2 weight decay added to the gradient; train_phoneme_to_articulation.py:
 * 177-181) over flat buffers, one launch.  step is the 1-based step count after this update. */
int as_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                 void* stream);

/* Batched collate on the device (pad_sequence_collate_fn, encoder_decoder/dataset.py:27-65, for a data set that is resident
 * in HBM): utterance b's `lengths[b]` rows start at row `first_row[b]` of `src` ([rows][row_elems], 4-byte floats or 8-byte
 * token ids); out [B][T][row_elems] = those rows followed by `pad_value` (0 for tokens / contours, -1 for voicing). */
int as_gather_pad_rows(const void* src, const int64_t* first_row, const int32_t* lengths, int32_t B, int32_t T,
                       int64_t row_elems, int32_t elem_bytes, double pad_value, void* out, void* stream);

/* Optional per-kernel-phase timing with HIP events recorded on the launch stream (for bench.py's
 * roofline object).  as_profile_report writes "name count total_ms\n" lines (NUL terminated, truncated
 * to buflen) and returns the untruncated length; it waits for the recorded events to complete. */
void as_profile_enable(int32_t on);
void as_profile_reset(void);
/* Diagnostic: in-kernel cycle stamps of the fused head-layer kernels (lin_f32.hip).  buf = device array of
 * 8 x (number of workgroups of the next launches) uint64, or NULL to switch the stamps off (the default; a kernel
 * launched without a buffer executes no stamp).  Per workgroup: s_memtime at start, after the prologue's first wait,
 * after the main loop, at the end; s_memrealtime at start; HW_ID and XCC_ID registers.  Never used by the product path. */
void as_lin_debug_stamps(uint64_t* buf, int64_t max_workgroups);
/* Diagnostic: per-workgroup stamps of the GRU backward recurrences: buf = device array of 2 x (4 x 2 x B) uint64, or NULL
 * (default: no stamps).  Consecutive backward launches alternate between the two halves (a training step launches layer 1,
 * then layer 0: half 0 and half 1 after a call that reset the counter); per (direction, utterance):
 * {shader cycles, 100 MHz wall ticks, sequence length, wall start}. */
void as_gru_debug_stamps(uint64_t* buf);
int32_t as_profile_report(char* buf, int32_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* ARTSPEECH_HIP_H */
