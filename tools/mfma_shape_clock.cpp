// mfma_shape_clock.cpp -- does the fp32 MFMA SHAPE change the clock the chip sustains?  (MI355X guide, DVFS give-back
// item 7, measured there for bf16.)  Bare chains of v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 at equal
// flops, operands = random values held in registers (rotated), two waves per SIMD, every CU busy; reports TFLOP/s from
// wall time and the in-kernel clock (s_memtime / s_memrealtime) after a long warm-up.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const float *rnd, float *out, unsigned long long *cyc, unsigned long long *rt, int iters) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = rnd[(threadIdx.x * 16 + i) & 8191]; b[i] = rnd[(threadIdx.x * 16 + i + 4096) & 8191]; }
    f32x16 c0 = {0}, c1 = {0};
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {          // 32 MFMAs x 4096 flop
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[i], a[i], c1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {          // 64 MFMAs x 2048 flop
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i], a[i], d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[(i + 1) & 15], d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i], a[(i + 1) & 15], d3, 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = c0[0] + c1[1] + d0[0] + d1[1] + d2[2] + d3[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int SHAPE>
int run(const float *rnd, int seconds_tenths) {
    const int blocks = 256, threads = 512, iters = 4000;       // 8 waves per CU = 2 per SIMD; one launch ~ 7 ms
    float *out; unsigned long long *cyc, *rt;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
    CHECK(hipMalloc(&cyc, blocks * 8)); CHECK(hipMalloc(&rt, blocks * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int warm = 14 * seconds_tenths, reps = 40;
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(threads), 0, 0, rnd, out, cyc, rt, iters);
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(threads), 0, 0, rnd, out, cyc, rt, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks), hr(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hr.data(), rt, blocks * 8, hipMemcpyDeviceToHost));
    double c = 0, r = 0; for (int i = 0; i < blocks; ++i) { c += h[i]; r += hr[i]; }
    const double flops = (double)reps * blocks * (threads / 64) * iters * 32.0 * 4096.0;
    printf("shape %s: %.1f TFLOP/s (%.1f %% of 157.3), in-kernel clock %.0f MHz, cycles per 4096 flop per SIMD %.1f\n",
           SHAPE == 32 ? "32x32x2" : "16x16x4", flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 1.573, c / r * 100,
           (c / blocks) / (iters * 32.0 * 2));
    hipFree(out); hipFree(cyc); hipFree(rt);
    return 0;
}

int main() {
    std::vector<float> h(8192);
    std::mt19937 rng(7);
    std::normal_distribution<float> n(0.f, 1.f);
    for (auto &x : h) x = n(rng);
    float *rnd; CHECK(hipMalloc(&rnd, 8192 * 4));
    CHECK(hipMemcpy(rnd, h.data(), 8192 * 4, hipMemcpyHostToDevice));
    for (int round = 0; round < 2; ++round) {
        if (run<32>(rnd, 10)) return 1;
        if (run<16>(rnd, 10)) return 1;
    }
    return 0;
}
