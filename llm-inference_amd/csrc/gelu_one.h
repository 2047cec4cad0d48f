// gelu_one.h -- the tanh-GELU of one fp32 value, shared by gelu_fwd.hip and linear_rows_fwd.hip (same bits in both).
// Follows the reference expression tree (reference gelu.py:9-12) one rounded fp32 operation at a time: __fmul_rn /
// __fadd_rn keep hipcc from contracting x + c*x^3 into an FMA.
#pragma once
#include <hip/hip_runtime.h>

namespace {

constexpr float kSqrt2OverPi = 0.7978845238685608f;   // sqrtf((float)(2/pi)) = torch.sqrt(torch.tensor(2.0 / torch.pi)), gelu.py:10
constexpr float kCubic = 0.44715f;                    // gelu.py:11

__device__ __forceinline__ float gelu_one(float x) {
    const float x3 = __fmul_rn(__fmul_rn(x, x), x);                  // torch.pow(x, 3) == x*x*x
    const float inner = __fadd_rn(x, __fmul_rn(kCubic, x3));
    const float t = tanhf(__fmul_rn(kSqrt2OverPi, inner));
    return __fmul_rn(__fmul_rn(0.5f, x), __fadd_rn(1.0f, t));
}

}  // namespace
