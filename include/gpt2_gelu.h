/* gpt2_gelu.h -- C ABI of the fused tanh-GELU forward.
 *
 * Replaces the reference's GELU module (reference gelu.py:4-12), which PyTorch runs as eight
 * elementwise launches over the (T, 4*d_model) MLP activation -- pow, three scalar multiplies, two adds,
 * tanh, one product -- i.e. eight HBM round trips; here it is ONE launch that reads x once and writes y
 * once.  Same library and conventions as include/cmhsa.h (libcmhsa_hip.so: device pointers, no
 * allocation, no synchronisation, status codes of cmhsa.h, `stream` is a hipStream_t).
 *
 * Function computed, with the reference's constants AND its operation order, every product and sum
 * rounded to fp32 on its own (no FMA contraction), so that the result equals the unfused chain:
 *     y = (0.5 * x) * (1 + tanh( sqrt(2/pi) * (x + 0.44715 * (x*x*x)) ))
 * (0.44715 is the reference's constant, gelu.py:11 -- not the textbook 0.044715.)
 */
#ifndef GPT2_GELU_H
#define GPT2_GELU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* y[i] = gelu_tanh(x[i]) for i in [0, n).  x and y: fp32 device pointers, contiguous; may be equal
 * (in place); any alignment (16-B-aligned pointers take the vector path).  n == 0 is a no-op.
 * Returns CMHSA_OK, CMHSA_ERR_NULL_POINTER, CMHSA_ERR_BAD_SHAPE (n < 0), CMHSA_ERR_TOO_LARGE
 * (n >= 2^40) or a positive hipError_t.                                          [gelu.py:8-12] */
int gpt2_gelu_tanh_f32_fwd(const float *x, float *y, int64_t n, void *stream);

/* Algorithmic HBM bytes of one call: 8 * n (read x once, write y once). */
double gpt2_gelu_algorithmic_bytes(int64_t n);

#ifdef __cplusplus
}
#endif

#endif /* GPT2_GELU_H */
