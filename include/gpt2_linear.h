/* gpt2_linear.h -- C ABI of the few-row linear layer (a caller-side kernel next to the attention path, like
 * include/gpt2_gelu.h and include/gpt2_layernorm.h; same library and conventions as include/cmhsa.h: libcmhsa_hip.so,
 * device pointers, no allocation, no synchronisation, status codes of cmhsa.h, `stream` is a hipStream_t).
 *
 * For the opt-in KV-cache decode loop, where every nn.Linear around the attention op
 * sees ONE row per prompt: the fused QKV layer and the output projection (reference causal_self_attention.py:224,
 * :236) and the two MLP layers (reference transformer_block.py:35-37).  With M = 1 the stock path is a bias copy plus
 * an fp32 GEMM tile kernel (2.5 + 9.9 us on GPT2-small's 768 x 2304 layer); the work is reading the weight matrix once.
 *
 * Function computed (torch.nn.functional.linear, optionally followed by the reference's tanh-GELU, gelu.py:8-12):
 *     y[m, n] = act( sum_k x[m, k] * weight[n, k] + bias[n] )        m < rows <= 8, n < n_out
 * Weight-streaming: one wave per output column (or a few), the whole matrix requested at once.  Measured on MI355X against
 * torch.nn.functional.linear: 2.2-3.6 us against 8.3-9.0 us at 1 row on GPT2-small's layers (GPT2-XL's: 4.1-13 against
 * 11-20 us), the 50 257-row LM head at 5.9 TB/s; ~1.4x faster at 4 rows, even at 8 rows on GPT2-small's widths and slower
 * than the GEMM kernels beyond that -- a kernel for 1-4 rows that stays correct up to 8.
 * fp32 throughout; the sum over k is formed as 64 strided partial sums (one per lane, 4 consecutive k at a time) that
 * are then added in a fixed tree, so results are deterministic and within a few ulp-of-the-sum of any other order.
 */
#ifndef GPT2_LINEAR_H
#define GPT2_LINEAR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPT2_LINEAR_MAX_ROWS 8
#define GPT2_LINEAR_ACT_NONE 0
#define GPT2_LINEAR_ACT_GELU_TANH 1

/* x: rows x k (row m at x + m * x_row_stride), weight: n_out x k (row n at weight + n * w_row_stride, the layout of
 * torch.nn.Linear.weight), bias: n_out floats or NULL, y: rows x n_out (row m at y + m * y_row_stride).
 * Requirements: 1 <= rows <= GPT2_LINEAR_MAX_ROWS (rows == 0 or n_out == 0 is a no-op), k a positive multiple of 4,
 * x and weight 16-B aligned with row strides that are multiples of 4, y and bias 4-B aligned; x, weight rows < 2^31 B.
 * Returns CMHSA_OK, CMHSA_ERR_NULL_POINTER, CMHSA_ERR_BAD_SHAPE (negative sizes or strides, k not a multiple of 4,
 * rows > 8, an unknown act), CMHSA_ERR_ALIGNMENT, CMHSA_ERR_TOO_LARGE or a positive hipError_t.
 *                       [torch.nn.Linear as used at causal_self_attention.py:224, :236, transformer_block.py:35-37] */
int gpt2_linear_rows_f32_fwd(const float *x, int64_t x_row_stride, const float *weight, int64_t w_row_stride,
                             const float *bias, float *y, int64_t y_row_stride,
                             int rows, int n_out, int k, int act, void *stream);

#define GPT2_LN_LINEAR_MAX_ROWS 4
#define GPT2_LN_LINEAR_MAX_K 4096

/* The same with the residual add and the LayerNorm of the input in front -- the step's QKV layer and first MLP layer,
 * whose input is `LayerNorm(x + sublayer_out)` (reference transformer_block.py:26-39): ONE launch instead of the
 * add + LayerNorm launch (include/gpt2_layernorm.h) and the linear launch.
 *     s[m, :] = x[m, :] + delta[m, :]            (delta == NULL: s = x; sum_out may then be NULL and is not written)
 *     h[m, :] = ln_weight * (rsqrt(var(s[m, :]) + eps) * (s[m, :] - mean(s[m, :]))) + ln_bias      (torch.nn.LayerNorm)
 *     y[m, n] = act( sum_k h[m, k] * weight[n, k] + bias[n] )
 * sum_out (rows x k, contiguous) receives s.  Every wave forms the two statistics of the rows itself (it reads whole rows
 * of x anyway).  Requirements as above, and: rows <= GPT2_LN_LINEAR_MAX_ROWS, k <= GPT2_LN_LINEAR_MAX_K; delta, ln_weight,
 * ln_bias, sum_out 16-B aligned; sum_out must not alias x or delta (other waves are still reading them).
 *                                   [transformer_block.py:26-39 + causal_self_attention.py:224 / transformer_block.py:35] */
int gpt2_ln_linear_rows_f32_fwd(const float *x, int64_t x_row_stride, const float *delta, int64_t delta_row_stride,
                                const float *ln_weight, const float *ln_bias, float eps, float *sum_out,
                                const float *weight, int64_t w_row_stride, const float *bias, float *y, int64_t y_row_stride,
                                int rows, int n_out, int k, int act, void *stream);

/* Algorithmic HBM bytes of one call: the weight matrix and the bias once, x once, y once. */
double gpt2_linear_rows_algorithmic_bytes(int rows, int n_out, int k, int with_bias);

#ifdef __cplusplus
}
#endif

#endif /* GPT2_LINEAR_H */
