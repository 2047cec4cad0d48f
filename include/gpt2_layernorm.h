/* gpt2_layernorm.h -- C ABI of the fused residual-add + LayerNorm forward (a caller-side fusion next to the attention
 * path, like include/gpt2_gelu.h; same library and conventions as include/cmhsa.h: libcmhsa_hip.so, device pointers,
 * no allocation, no synchronisation, status codes of cmhsa.h, `stream` is a hipStream_t).
 *
 * Replaces, in the generation loop, the pair
 *     x = x + sublayer_out                  (reference transformer_block.py:31, :38: the residual add, one launch)
 *     h = nn.LayerNorm(emb_dim)(x)          (reference transformer_block.py:28, :34, gpt2.py:37: the next sublayer's
 *                                            LayerNorm, another launch)
 * by ONE launch that reads x and the sublayer output once and writes both the new residual stream and its LayerNorm.
 * GPT2-small's 512-token generation is launch-latency bound (about 125 kernels of 4-20 us per token); this removes 24.
 *
 * Function computed per row r (eps and the biased variance as torch.nn.LayerNorm):
 *     s[r, :]   = x[r, :] + delta[r, :]                         (delta == NULL: s = x, nothing written to sum_out)
 *     mean, var = mean(s[r, :]), mean((s[r, :] - mean)^2)       (two passes over registers, fp32)
 *     y[r, :]   = weight * (rsqrt(var + eps) * (s[r, :] - mean)) + bias
 */
#ifndef GPT2_LAYERNORM_H
#define GPT2_LAYERNORM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* x, delta: fp32 device pointers, `rows` rows of `cols` floats, row r at ptr + r * row_stride (floats); delta may be
 * NULL (plain LayerNorm).  weight, bias: cols floats.  sum_out (may be NULL when delta is NULL; may equal x: in place)
 * and norm_out: contiguous (rows, cols).  Requirements: cols a multiple of 4, 4 <= cols <= 4096; 16-B-aligned pointers,
 * row strides multiples of 4.  rows == 0 is a no-op.
 * Returns CMHSA_OK, CMHSA_ERR_NULL_POINTER, CMHSA_ERR_BAD_SHAPE (negative sizes, cols out of range, a negative
 * stride), CMHSA_ERR_ALIGNMENT, CMHSA_ERR_TOO_LARGE (rows >= 2^31) or a positive hipError_t.
 *                                                  [transformer_block.py:26-39; torch.nn.LayerNorm semantics] */
int gpt2_add_layernorm_f32_fwd(const float *x, const float *delta, const float *weight, const float *bias,
                               float *sum_out, float *norm_out, int64_t rows, int cols,
                               int64_t x_row_stride, int64_t delta_row_stride, float eps, void *stream);

/* Algorithmic HBM bytes of one call: rows * cols * 4 * (2 reads + 2 writes with delta, 1 + 1 without) + 8 * cols. */
double gpt2_add_layernorm_algorithmic_bytes(int64_t rows, int cols, int with_delta);

#ifdef __cplusplus
}
#endif

#endif /* GPT2_LAYERNORM_H */
