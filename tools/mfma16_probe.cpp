// mfma16_probe.cpp -- checks, on the device, the operand / result layout of v_mfma_f32_16x16x4_f32 and the lane
// exchanges (v_permlane16_swap, v_permlane32_swap) that the 16-row-block attention kernel (cmhsa_fwd16.hip) assumes:
//   A (16 x 4):  lane l holds A[m = l % 16][k = l / 16]
//   B (4 x 16):  lane l holds B[k = l / 16][n = l % 16]
//   C (16 x 16): lane l, register i holds C[m = 4 * (l / 16) + i][n = l % 16]
//   permlane16_swap(x, x): {r0, r1} with max(r0, r1)[l] = max(x[l], x[l ^ 16]);  permlane32_swap: l ^ 32
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma16_probe tools/mfma16_probe.cpp     Prints "ok" or the mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(const float *A, const float *B, float *C, float *x16, float *x32) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) C[(4 * (l / 16) + i) * 16 + l % 16] = c[i];
    const float v = (float)((l * 37) % 64);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    x16[l] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    x32[l] = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}

int main() {
    float hA[64], hB[64], hC[256], h16[64], h32[64];
    for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 11) - 5.f; hB[i] = (float)((i * 5) % 13) - 6.f; }
    float *A, *B, *C, *x16, *x32;
    CHECK(hipMalloc(&A, 256)); CHECK(hipMalloc(&B, 256)); CHECK(hipMalloc(&C, 1024)); CHECK(hipMalloc(&x16, 256)); CHECK(hipMalloc(&x32, 256));
    CHECK(hipMemcpy(A, hA, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(B, hB, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, C, x16, x32);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hC, C, 1024, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h16, x16, 256, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h32, x32, 256, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float want = 0.f;
            for (int k = 0; k < 4; ++k) want += hA[m * 4 + k] * hB[k * 16 + n];
            if (want != hC[m * 16 + n] && ++bad < 8) printf("C[%d][%d] = %g, want %g\n", m, n, hC[m * 16 + n], want);
        }
    for (int l = 0; l < 64; ++l) {
        const float v = (float)((l * 37) % 64), v16 = (float)(((l ^ 16) * 37) % 64), v32 = (float)(((l ^ 32) * 37) % 64);
        if (h16[l] != fmaxf(v, v16) && ++bad < 16) printf("permlane16_swap lane %d: %g, want %g\n", l, h16[l], fmaxf(v, v16));
        if (h32[l] != fmaxf(v, v32) && ++bad < 24) printf("permlane32_swap lane %d: %g, want %g\n", l, h32[l], fmaxf(v, v32));
    }
    printf(bad ? "MISMATCHES: %d\n" : "ok: mfma 16x16x4 layout and lane swaps as assumed\n", bad);
    return bad ? 1 : 0;
}
