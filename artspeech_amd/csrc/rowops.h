// Internal (non-ABI) launch helpers shared by the composite entry points.
#pragma once
#include "as_common.h"

// pos_bits (optional): x > 0 per element, ceil(D / 64) 64-bit words per row -- the ReLU mask of a post-ReLU input, which
// as_normalize_bwd can take as relu_bits instead of re-reading the activation itself (relu_src)
int as_normalize_fwd(const float* x, float* xhat, float* rstd, long rows, int D, hipStream_t st, unsigned long long* pos_bits = nullptr);
int as_normalize_bwd(const float* dy, const float* xhat, const float* rstd, const float* relu_src, float* dx, long rows,
                     int D, hipStream_t st, const unsigned long long* relu_bits = nullptr, int n_slab = 1, long slab_stride = 0);
// (n_slab > 1: dy is the sum of n_slab partial results slab_stride floats apart, added in slab order)
// Rpad (>= R, 0 = R): rows per head in Wf; the extra rows are written as zeros
int as_fold(const float* W, const float* gamma, const float* beta, const float* b, float* Wf, float* bf, int heads, int R,
            int K, hipStream_t st, int Rpad = 0);
int as_unfold(const float* dWf, const float* dbf, const float* W, const float* gamma, const float* beta, float* dW,
              float* dgamma, float* dbeta, int heads, int R, int K, hipStream_t st);
// the three layers of a head stack in one launch (index 0..2 = any order)
int as_unfold3(const float* const dWf[3], const float* const dbf[3], const float* const W[3], const float* const gamma[3],
               const float* const beta[3], float* const dW[3], float* const dgamma[3], float* const dbeta[3], const int R[3],
               const int K[3], int heads, hipStream_t st, int count = 3);   // the first `count` (1..3) entries
int as_token_segsum(const float* x, const int64_t* tokens, long tok_stride, int T, long rows, int C, int V, float* out,
                    hipStream_t st, float* scratch = nullptr, long scratch_floats = 0);
// dW [C][E] = dtab^T . emb, db [C] = column sums of dtab, demb [V][E] = dtab . W   (dtab [V][C], emb [V][E], W [C][E])
// tab [V][C] = emb [V][E] . W [C][E]^T + bias [C]   (V * E floats must fit the LDS: the caller checks V * E <= 16384)
int as_token_table(const float* emb, const float* W, const float* bias, int V, int C, int E, float* tab, hipStream_t st);
int as_emb_grads(const float* dtab, const float* emb, const float* W, int V, int C, int E, float* dW, float* db, float* demb,
                 hipStream_t st);
// V > 0: ids are clamped into [0, V) (the table has V rows)
int as_gather_rows(const float* table, const int64_t* tokens, long tok_stride, int T, long rows, int C, float* out,
                   hipStream_t st, int V = 0);
int as_sigmoid_bwd(const float* out, const float* dout, float* dpre, long n, hipStream_t st);
int as_relu_mask(const float* g, const float* act, float* dst, long n, hipStream_t st);
// y = x * mask(seed, i) / (1 - p); x == y allowed (in place); the same (seed, p) regenerates the same mask
int as_dropout(const float* x, float* y, long n, float p, unsigned long long seed, hipStream_t st);
