// gpt2_step.hip -- the two bookkeeping kernels of the graph-replayed greedy loop (include/gpt2_step.h): the embedding
// sum on the model's input side and arg-max + EOS handling + token write on the sampler's side.  Both are
// latency-bound (a few hundred KB); they exist to replace chains of four and seven tiny torch launches by one each.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmhsa.h"
#include "gpt2_step.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per (b, t) row, four rows per workgroup
__global__ __launch_bounds__(256) void embed_kernel(const int64_t *tokens, int64_t tok_rs, const float *wte, const float *wpe, float *x,
                                                    int rows, int seq_len, int d, int vocab) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / seq_len, t = row - b * seq_len;
    int64_t id = tokens[b * tok_rs + t];
    id = id < 0 ? 0 : id >= vocab ? vocab - 1 : id;            // never fault on a bad id (documented: caller error)
    const f32x4 *e = reinterpret_cast<const f32x4 *>(wte + id * (int64_t)d);
    const f32x4 *p = reinterpret_cast<const f32x4 *>(wpe + (int64_t)t * d);
    f32x4 *o = reinterpret_cast<f32x4 *>(x + (int64_t)row * d);
    for (int i = lane; i < d / 4; i += 64) o[i] = e[i] + p[i];
}

constexpr int kPickThreads = 1024;

// strictly greater wins, equal keeps the smaller index: the first maximum, as torch.argmax (reference sampler.py:17) --
// which ranks NaN above every number, so the first NaN of a row wins
__device__ __forceinline__ void better(float &bv, int &bi, float v, int i) {
    const bool vn = v != v, bn = bv != bv;
    const bool gt = vn ? !bn : (!bn && v > bv), eq = vn ? bn : (!bn && v == bv);
    if (gt || (eq && i < bi)) { bv = v; bi = i; }
}

// one workgroup per batch row
__global__ __launch_bounds__(kPickThreads) void greedy_step_kernel(const float *logits, int64_t lrs, unsigned char *done, int64_t *tokens,
                                                                   int64_t tok_rs, const int64_t *pos, int vocab, int cap, int64_t eos) {
    __shared__ float sv[kPickThreads / 64];
    __shared__ int si[kPickThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *row = logits + b * lrs;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    // eight independent loads in flight per thread (a row of 50 257 logits is 49 per thread: latency, not bandwidth)
    for (int base = tid; base < vocab; base += 8 * kPickThreads) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + k * kPickThreads;
            v[k] = i < vocab ? row[i] : -INFINITY;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + k * kPickThreads < vocab) better(bv, bi, v[k], base + k * kPickThreads);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) better(bv, bi, __shfl_xor(bv, m, 64), __shfl_xor(bi, m, 64));
    if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kPickThreads / 64; ++w) better(bv, bi, sv[w], si[w]);
        if (bi == 0x7fffffff) bi = 0;                           // (not reached: -inf equals the initial value and takes its index)
        int64_t next = done[b] ? eos : (int64_t)bi;
        if (next == eos) done[b] = 1;
        const int64_t col = pos[0] + 1;
        if (col >= 0 && col < cap) tokens[b * tok_rs + col] = next;
    }
}

}  // namespace

extern "C" {

int gpt2_embed_f32_fwd(const int64_t *tokens, int64_t tokens_row_stride, const float *wte, const float *wpe, float *x,
                       int batch, int seq_len, int d, int vocab, void *stream) {
    if (batch < 0 || seq_len < 0 || d < 4 || (d & 3) || vocab <= 0 || tokens_row_stride < 0) return CMHSA_ERR_BAD_SHAPE;
    if (batch == 0 || seq_len == 0) return CMHSA_OK;
    if (!tokens || !wte || !wpe || !x) return CMHSA_ERR_NULL_POINTER;
    if ((((uintptr_t)wte | (uintptr_t)wpe | (uintptr_t)x) & 15) || ((uintptr_t)tokens & 7)) return CMHSA_ERR_ALIGNMENT;
    const int64_t rows64 = (int64_t)batch * seq_len;
    if (rows64 > 0x7fffffffLL - 4) return CMHSA_ERR_TOO_LARGE;
    int rows = (int)rows64;
    void *args[] = {&tokens, &tokens_row_stride, &wte, &wpe, &x, &rows, &seq_len, &d, &vocab};
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(embed_kernel), dim3((unsigned)((rows + 3) / 4)), dim3(256), args, 0,
                                         static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

int gpt2_greedy_step(const float *logits, int64_t logits_row_stride, unsigned char *done, int64_t *tokens,
                     int64_t tokens_row_stride, const int64_t *pos, int batch, int vocab, int cap, int64_t eos, void *stream) {
    if (batch < 0 || vocab <= 0 || cap <= 0 || logits_row_stride < 0 || tokens_row_stride < 0) return CMHSA_ERR_BAD_SHAPE;
    if (batch == 0) return CMHSA_OK;
    if (!logits || !done || !tokens || !pos) return CMHSA_ERR_NULL_POINTER;
    if (((uintptr_t)logits & 3) || (((uintptr_t)tokens | (uintptr_t)pos) & 7)) return CMHSA_ERR_ALIGNMENT;
    if (batch > 65535 * 256) return CMHSA_ERR_TOO_LARGE;
    void *args[] = {&logits, &logits_row_stride, &done, &tokens, &tokens_row_stride, &pos, &vocab, &cap, &eos};
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(greedy_step_kernel), dim3((unsigned)batch), dim3(kPickThreads), args, 0,
                                         static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

}  // extern "C"
