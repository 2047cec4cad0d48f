// gelu_fwd.hip -- fused tanh-GELU forward for MI355X (gfx950), C ABI in include/gpt2_gelu.h.
//
// HBM-bound elementwise kernel (8 algorithmic bytes per element, ~20 flops): one 16-B load and one 16-B
// store per lane per step, four independent steps in flight per lane, grid sized to a few waves per SIMD and
// grid-strided beyond that.  Arithmetic follows the reference expression tree (reference gelu.py:9-12) one
// rounded fp32 operation at a time -- __fmul_rn / __fadd_rn keep hipcc from contracting x + c*x^3 into an
// FMA -- so the result matches PyTorch's eight-kernel evaluation of the same module bit for bit wherever
// tanhf does.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmhsa.h"
#include "gpt2_gelu.h"

namespace {

constexpr float kSqrt2OverPi = 0.7978845238685608f;   // sqrtf((float)(2/pi)) = torch.sqrt(torch.tensor(2.0 / torch.pi)), gelu.py:10
constexpr float kCubic = 0.44715f;                    // gelu.py:11
constexpr int kThreads = 256;
constexpr int kUnroll = 4;

__device__ __forceinline__ float gelu_one(float x) {
    const float x3 = __fmul_rn(__fmul_rn(x, x), x);                  // torch.pow(x, 3) == x*x*x
    const float inner = __fadd_rn(x, __fmul_rn(kCubic, x3));
    const float t = tanhf(__fmul_rn(kSqrt2OverPi, inner));
    return __fmul_rn(__fmul_rn(0.5f, x), __fadd_rn(1.0f, t));
}

__global__ __launch_bounds__(kThreads) void gelu_vec_kernel(const float4 *x, float4 *y, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + (kUnroll - 1) * stride < n4; i += kUnroll * stride) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            v[u].x = gelu_one(v[u].x); v[u].y = gelu_one(v[u].y); v[u].z = gelu_one(v[u].z); v[u].w = gelu_one(v[u].w);
            y[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) {
        float4 v = x[i];
        v.x = gelu_one(v.x); v.y = gelu_one(v.y); v.z = gelu_one(v.z); v.w = gelu_one(v.w);
        y[i] = v;
    }
}

// unaligned pointers and the < 4-element tail
__global__ __launch_bounds__(kThreads) void gelu_scalar_kernel(const float *x, float *y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) y[i] = gelu_one(x[i]);
}

int grid_for(int64_t items) {
    // 256 CUs x 8 workgroups of 256 threads = 8 waves per SIMD: enough loads in flight to cover HBM latency
    const int64_t want = (items + kThreads - 1) / kThreads;
    return (int)(want < 2048 ? (want > 0 ? want : 1) : 2048);
}

}  // namespace

extern "C" {

double gpt2_gelu_algorithmic_bytes(int64_t n) { return 8.0 * (double)n; }

int gpt2_gelu_tanh_f32_fwd(const float *x, float *y, int64_t n, void *stream) {
    if (n < 0) return CMHSA_ERR_BAD_SHAPE;
    if (n == 0) return CMHSA_OK;
    if (!x || !y) return CMHSA_ERR_NULL_POINTER;
    if (n >= (1LL << 40)) return CMHSA_ERR_TOO_LARGE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool aligned = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    const int64_t n4 = aligned ? n / 4 : 0;
    if (n4 > 0)
        hipLaunchKernelGGL(gelu_vec_kernel, dim3(grid_for((n4 + kUnroll - 1) / kUnroll)), dim3(kThreads), 0, s,
                           reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y), n4);
    const int64_t rest = n - 4 * n4;
    if (rest > 0)
        hipLaunchKernelGGL(gelu_scalar_kernel, dim3(grid_for(rest)), dim3(kThreads), 0, s, x + 4 * n4, y + 4 * n4, rest);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

}  // extern "C"
