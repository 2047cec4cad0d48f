/* gpt2_step.h -- C ABI of the two bookkeeping kernels of the graph-replayed greedy generation loop (the generation
 * harness of SURVEY.md section 8(f) row 1; same library and conventions as include/cmhsa.h: libcmhsa_hip.so, device
 * pointers, no allocation, no synchronisation, status codes of cmhsa.h, `stream` is a hipStream_t).
 *
 * GPT2-small's 512-token generation is bound by per-kernel latencies (about 125 launches of 4-20 us per token).  The
 * model's input side and the sampler's output side are chains of tiny torch launches; each chain is ONE launch here:
 *
 *   gpt2_embed_f32_fwd      x[b, t, :] = wte[tokens[b, t], :] + wpe[t, :]
 *                           (reference gpt2.py:43-44: two embedding lookups, an arange and an add: four launches)
 *   gpt2_greedy_step        next = argmax(logits[b, :]) with the FIRST maximum winning (torch.argmax, reference
 *                           sampler.py:16-17), or eos for a row that is done; done[b] |= next == eos;
 *                           tokens[b, pos + 1] = next          (reference sampler.py:14-19; seven launches in torch)
 * Both are exact: one fp32 add per element; comparisons only.
 */
#ifndef GPT2_STEP_H
#define GPT2_STEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tokens: int64 (batch rows of `tokens_row_stride` elements), the first seq_len of each row are used; wte: (vocab, d),
 * wpe: (>= seq_len, d), x: (batch, seq_len, d), all contiguous fp32; d a multiple of 4, 16-B-aligned pointers.
 * A token id outside [0, vocab) is a caller error; the kernel clamps it into range (no fault) -- check on the host.
 * Returns CMHSA_OK, CMHSA_ERR_NULL_POINTER, CMHSA_ERR_BAD_SHAPE, CMHSA_ERR_ALIGNMENT, CMHSA_ERR_TOO_LARGE or a positive
 * hipError_t.                                                                                  [gpt2.py:43-44] */
int gpt2_embed_f32_fwd(const int64_t *tokens, int64_t tokens_row_stride, const float *wte, const float *wpe, float *x,
                       int batch, int seq_len, int d, int vocab, void *stream);

/* logits: (batch, vocab) fp32, row stride logits_row_stride (floats); done: `batch` bytes (bool); tokens: int64 rows of
 * tokens_row_stride elements with capacity `cap`; pos: ONE int64 on the device = index of the last valid token (the
 * kernel writes column pos + 1 and leaves pos alone: increment it afterwards, in stream order).  The first maximum wins and NaN
 * ranks above every number -- exactly torch.argmax, which the reference's sampler calls.  pos + 1 >= cap writes nothing.
 * Returns as above.                                                                          [sampler.py:14-19] */
int gpt2_greedy_step(const float *logits, int64_t logits_row_stride, unsigned char *done, int64_t *tokens,
                     int64_t tokens_row_stride, const int64_t *pos, int batch, int vocab, int cap, int64_t eos, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* GPT2_STEP_H */
