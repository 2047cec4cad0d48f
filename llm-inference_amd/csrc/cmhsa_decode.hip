// cmhsa_decode.hip -- one-token ("decode") step of the fp32 causal attention over a KV cache, for MI355X (gfx950).
// C ABI: cmhsa_fp32_decode_step in include/cmhsa.h.  SURVEY.md section 8(f) row 4: the reference has no KV cache
// (reference README.md:12-13; its sampler re-runs the full forward per token, sampler.py:14-19) -- this is the
// opt-in mode that computes ONLY the last row of the reference formula (causal_self_attention.py:173-194):
//     o[b, 64h:64h+64] = softmax( q_h . K_h[0..pos]^T / 8 ) V_h[0..pos]
// where row `pos` of K/V is the new token's own key/value, appended to the cache by this same launch.
//
// HBM/L2-bound (2 * (pos+1) * 256 B per (batch, head), ~2 flops per byte): no MFMA here.  One workgroup per
// (batch, head); its 32 16-lane groups each own the key rows j = g, g+32, ...: a group reads one 256-B K row and
// one 256-B V row per step (16 B per lane, fully coalesced; four rows in flight), reduces the 64-term dot product
// over its 16 lanes with DPP row operations, and keeps a private online-softmax state (m, l, o[4 per lane]) in the
// log2 domain; the 32 partial states are merged through LDS at the end.  `pos` is read from DEVICE memory, so a
// captured HIP graph of the whole decode step can be replayed for every token.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmhsa.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHeadDim = 64;
constexpr int kThreads = 512;
constexpr int kGroups = kThreads / 16;                 // 32 groups of 16 lanes: one key row per group per step
constexpr int kRowsInFlight = 4;                       // (1024 threads x 8 rows in flight for B*H <= 256 -- 512 keys in one round trip -- measured 3 % SLOWER on the 512-token KV loop)
constexpr float kScaleLog2 = 0.18033688011112042f;     // log2(e) / sqrt(64)

template <int kCtrl>
__device__ __forceinline__ float dpp_add(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), kCtrl, 0xf, 0xf, false);
    return x + __builtin_bit_cast(float, y);
}

// sum over the 16 lanes of a DPP row, result in every lane: xor 1, xor 2 (quad_perm), then mirror within 8 and 16
__device__ __forceinline__ float row16_sum(float x) {
    x = dpp_add<0xB1>(x);      // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);      // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);     // row_half_mirror
    x = dpp_add<0x140>(x);     // row_mirror
    return x;
}

struct DecodeParams {
    const float *qkv_new;      // (B, 3*d_model): [q | k | v] of the new token
    float *k_cache, *v_cache;  // (B, capacity, d_model)
    float *o;                  // (B, d_model)
    const int *pos;            // device scalar: index of the new token (keys 0..pos take part)
    int64_t qkv_rs, cache_rs, cache_bs, o_rs;
    int capacity, num_heads;
};

__global__ __launch_bounds__(kThreads) void cmhsa_decode_kernel(DecodeParams p) {
    __shared__ float s_m[kGroups], s_l[kGroups];
    __shared__ __attribute__((aligned(16))) float s_o[kGroups][kHeadDim];

    const int pos = *p.pos;
    const int b = blockIdx.x / p.num_heads, h = blockIdx.x - b * p.num_heads;
    const int tid = threadIdx.x, g = tid >> 4, c = tid & 15;         // group, 16-B chunk of the 256-B head row
    if (pos < 0 || pos >= p.capacity) {                  // uniform; a caller's bug: no key exists -- o = 0, caches untouched
        if (tid < kHeadDim) p.o[b * p.o_rs + (int64_t)h * kHeadDim + tid] = 0.f;
        return;
    }
    const int64_t d_model = (int64_t)p.num_heads * kHeadDim;
    const float *qkv = p.qkv_new + b * p.qkv_rs + (int64_t)h * kHeadDim;
    const float *knew = qkv + d_model, *vnew = qkv + 2 * d_model;
    float *kc = p.k_cache + b * p.cache_bs + (int64_t)h * kHeadDim;
    float *vc = p.v_cache + b * p.cache_bs + (int64_t)h * kHeadDim;

    // append the new token's key and value to the cache (row pos); the attention below takes them from qkv_new
    if (tid < 16) *reinterpret_cast<f32x4 *>(kc + (int64_t)pos * p.cache_rs + 4 * tid) = *reinterpret_cast<const f32x4 *>(knew + 4 * tid);
    else if (tid < 32) *reinterpret_cast<f32x4 *>(vc + (int64_t)pos * p.cache_rs + 4 * (tid - 16)) = *reinterpret_cast<const f32x4 *>(vnew + 4 * (tid - 16));

    const f32x4 q4 = *reinterpret_cast<const f32x4 *>(qkv + 4 * c) * kScaleLog2;
    float m = -INFINITY, l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = g; j0 <= pos; j0 += kGroups * kRowsInFlight) {
        f32x4 kr[kRowsInFlight], vr[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
            const int j = j0 + u * kGroups;
            // row pos is the new token (from qkv_new, not yet visible in the cache); rows past it are masked below and
            // read the new token's finite values too, never stale cache memory (0 * NaN would poison the sum)
            const float *kp = j < pos ? kc + (int64_t)j * p.cache_rs : knew;
            const float *vp = j < pos ? vc + (int64_t)j * p.cache_rs : vnew;
            kr[u] = *reinterpret_cast<const f32x4 *>(kp + 4 * c);
            vr[u] = *reinterpret_cast<const f32x4 *>(vp + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
            const int j = j0 + u * kGroups;
            float s = row16_sum(q4.x * kr[u].x + q4.y * kr[u].y + q4.z * kr[u].z + q4.w * kr[u].w);
            if (j > pos) s = -INFINITY;
            const float m_new = fmaxf(m, s);
            // m_new == -inf only while this group has seen no valid row: keep the state untouched
            const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
            const float pj = s == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s - m_new);
            l = l * alpha + pj;
            acc = acc * alpha + vr[u] * pj;
            m = m_new;
        }
    }
    if (c == 0) { s_m[g] = m; s_l[g] = l; }
    *reinterpret_cast<f32x4 *>(&s_o[g][4 * c]) = acc;
    __syncthreads();
    if (tid < kHeadDim) {
        float mt = -INFINITY;
#pragma unroll
        for (int i = 0; i < kGroups; ++i) mt = fmaxf(mt, s_m[i]);     // finite: the new token itself is always a key
        float lt = 0.f, ot = 0.f;
#pragma unroll
        for (int i = 0; i < kGroups; ++i) {
            const float f = __builtin_amdgcn_exp2f(s_m[i] - mt);      // 0 for groups that saw no row
            lt += s_l[i] * f;
            ot += s_o[i][tid] * f;
        }
        p.o[b * p.o_rs + (int64_t)h * kHeadDim + tid] = ot / lt;
    }
}

}  // namespace

extern "C" {

double cmhsa_decode_algorithmic_bytes(int batch, int num_keys, int num_heads, int head_dim) {
    // K and V rows of every key once, + q/k/v of the new token in, + the appended cache rows and o out
    return 4.0 * batch * num_heads * head_dim * (2.0 * num_keys + 3.0 + 3.0);
}

int cmhsa_fp32_decode_step(const float *qkv_new, int64_t qkv_row_stride,
                           float *k_cache, float *v_cache, int64_t cache_row_stride, int64_t cache_batch_stride,
                           float *o, int64_t o_row_stride, const int *pos_dev,
                           int batch, int capacity, int num_heads, int head_dim, void *stream) {
    if (batch < 0 || capacity <= 0 || num_heads <= 0) return CMHSA_ERR_BAD_SHAPE;
    if (head_dim != kHeadDim) return CMHSA_ERR_HEAD_DIM;
    if (batch == 0) return CMHSA_OK;
    if (!qkv_new || !k_cache || !v_cache || !o || !pos_dev) return CMHSA_ERR_NULL_POINTER;
    const int64_t d_model = (int64_t)num_heads * kHeadDim;
    if (qkv_row_stride < 3 * d_model || cache_row_stride < d_model || o_row_stride < d_model) return CMHSA_ERR_BAD_SHAPE;
    const uintptr_t ptr_bits = (uintptr_t)qkv_new | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)o;
    if ((ptr_bits & 15) || ((qkv_row_stride | cache_row_stride | cache_batch_stride | o_row_stride) & 3) || ((uintptr_t)pos_dev & 3))
        return CMHSA_ERR_ALIGNMENT;
    const int64_t blocks = (int64_t)batch * num_heads;
    if (blocks > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;
    DecodeParams p;
    p.qkv_new = qkv_new; p.k_cache = k_cache; p.v_cache = v_cache; p.o = o; p.pos = pos_dev;
    p.qkv_rs = qkv_row_stride; p.cache_rs = cache_row_stride; p.cache_bs = batch == 1 ? 0 : cache_batch_stride; p.o_rs = o_row_stride;
    p.capacity = capacity; p.num_heads = num_heads;
    // the launch's own status (hipGetLastError would also report, and clear, an unrelated earlier error of this thread)
    void *args[] = {&p};
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(cmhsa_decode_kernel), dim3((unsigned)blocks), dim3(kThreads),
                                         args, 0, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

}  // extern "C"
