// mfma_peak.cpp -- wall-clock fp32 MFMA throughput (diagnostic): NACC independent accumulators per wave,
// W waves per SIMD.  Prints TFLOP/s and cycles/MFMA/SIMD derived from s_memtime and from wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, unsigned long long *rt, float a0, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = (float)(i + 1) * a0;
    float a = a0 + threadIdx.x * 1e-3f, b = a0 * 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + i % NACC, b, acc[i % NACC], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int NACC>
int run(int threads, int blocks) {
    float *out; unsigned long long *cyc, *rt;
    const int iters = 2000;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
    CHECK(hipMalloc(&cyc, blocks * 8)); CHECK(hipMalloc(&rt, blocks * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, rt, 1.0f, iters);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, rt, 1.0f, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks), hr(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hr.data(), rt, blocks * 8, hipMemcpyDeviceToHost));
    double c = 0, r = 0; for (int i = 0; i < blocks; ++i) { c += h[i]; r += hr[i]; }
    c /= blocks; r /= blocks;
    const double mfmas_per_wave = 64.0 * iters, waves = (double)blocks * threads / 64;
    const double tf = waves * mfmas_per_wave * 4096 / (ms * 1e-3) / 1e12;
    printf("NACC=%d threads=%4d blocks=%4d: wall %.3f ms  %.1f TFLOP/s | memtime ticks/MFMA/wave %.1f | memtime/realtime ratio %.2f (x100MHz = %.0f MHz)\n",
           NACC, threads, blocks, ms, tf, c / mfmas_per_wave, c / r, c / r * 100);
    hipFree(out); hipFree(cyc); hipFree(rt);
    return 0;
}

int main() {
    run<1>(256, 256); run<4>(256, 256); run<1>(512, 256); run<4>(512, 256); run<1>(1024, 256); run<2>(1024, 256);
    run<4>(256, 1024); run<1>(256, 2048);
    return 0;
}
