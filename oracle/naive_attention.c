/* CPU ORACLE (test infrastructure only -- never linked into the product library).
 *
 * Plain-C fp32 restatement of the reference's naive causal multi-head self-attention,
 * scaled_dot_product_attention_naive_pytorch (reference causal_self_attention.py:173-194):
 *   scores = Q_h K_h^T (:187); scores[j>i] = -inf (:188); softmax(scores / sqrt(head_dim)) (:189);
 *   ctx = weights V_h (:191); heads merged back into (T, H*head_dim) (:192-193).
 * The softmax is the usual max-subtracted form (what torch.softmax computes).
 *
 * Parity pin: tests/test_oracle.py checks this against the golden vectors in tests/golden/,
 * which were produced by the reference module itself (tests/golden/make_golden.py).
 *
 * Row strides / batch strides are in floats, so the QKV-sliced layout of the model
 * (row stride 3*d_model, causal_self_attention.py:224-227) can be fed directly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_naive_attention_f32(const float *q, const float *k, const float *v, float *o,
                               int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                               int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                               int batch, int seq_len, int num_heads, int head_dim, int threads)
{
    if (batch < 0 || seq_len < 0 || num_heads <= 0 || head_dim <= 0) return -1;
    if (threads < 1) threads = 1;
    const float inv_sqrt_d = 1.0f / sqrtf((float)head_dim);
    const long total = (long)batch * num_heads * seq_len;
    int failed = 0;

#pragma omp parallel num_threads(threads)
    {
        float *w = (float *)malloc(sizeof(float) * (size_t)(seq_len > 0 ? seq_len : 1));
        if (!w) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp for schedule(dynamic, 16)
        for (long t = 0; t < total; ++t) {
            if (!w) continue;
            const int i = (int)(t % seq_len);
            const int h = (int)((t / seq_len) % num_heads);
            const int b = (int)(t / ((long)seq_len * num_heads));
            const float *qi = q + b * q_bs + (int64_t)i * q_rs + (int64_t)h * head_dim;
            /* scores for the unmasked keys j <= i; masked ones are -inf and drop out of the softmax */
            float mx = -INFINITY;
            for (int j = 0; j <= i; ++j) {
                const float *kj = k + b * k_bs + (int64_t)j * k_rs + (int64_t)h * head_dim;
                float acc = 0.0f;
                for (int d = 0; d < head_dim; ++d) acc += qi[d] * kj[d];
                acc *= inv_sqrt_d;
                w[j] = acc;
                if (acc > mx) mx = acc;
            }
            float sum = 0.0f;
            for (int j = 0; j <= i; ++j) {
                w[j] = expf(w[j] - mx);
                sum += w[j];
            }
            const float inv = 1.0f / sum;
            float *oi = o + b * o_bs + (int64_t)i * o_rs + (int64_t)h * head_dim;
            for (int d = 0; d < head_dim; ++d) oi[d] = 0.0f;
            for (int j = 0; j <= i; ++j) {
                const float *vj = v + b * v_bs + (int64_t)j * v_rs + (int64_t)h * head_dim;
                const float p = w[j] * inv;
                for (int d = 0; d < head_dim; ++d) oi[d] += p * vj[d];
            }
        }
        free(w);
    }
    return failed ? -2 : 0;
}
