// mfma_ubench.cpp -- what may sit between two v_mfma_f32_32x32x2_f32 without slowing the chain (gfx950 diagnostic).
// One wave per SIMD (256 threads x 256 blocks); 256 MFMAs between two s_memtime stamps; prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

// NACC chains; per MFMA: NEXP independent v_exp, NADD independent v_add; LDSW: ds_write_b128 every LDSW MFMAs (0 = none)
template <int NACC, int NEXP, int NADD, int LDSW, int LDSR, int VMEM = 0, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void k(float *out, unsigned long long *cyc, float a0, const float *src = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 2];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = (float)(i + 1) * a0;
    float a = a0 + threadIdx.x, b = a0 * 2.f;
    float e[8], d[8];
    for (int i = 0; i < 8; ++i) { e[i] = a0 * i; d[i] = a0 + i; }
    f32x4 w = {a, b, a, b}, rd = {0, 0, 0, 0};
    f32x4 vm[8];
    for (int i = 0; i < 8; ++i) vm[i] = rd;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 20, 0x00020000);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 256; ++i) {
        acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + (i % NACC), b, acc[i % NACC], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < NEXP; ++x) e[(i * NEXP + x) & 7] = __builtin_amdgcn_exp2f(e[(i * NEXP + x) & 7]);
#pragma unroll
        for (int x = 0; x < NADD; ++x) d[(i * NADD + x) & 7] += a0;
        if (LDSW && (i % (LDSW ? LDSW : 1)) == 0) *reinterpret_cast<f32x4 *>(&lds[threadIdx.x * 4 + (i & 2 ? 1024 : 0)]) = w;
        if (VMEM && (i % (VMEM ? VMEM : 1)) == 0) vm[(i / (VMEM ? VMEM : 1)) & 7] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (threadIdx.x * 16 + (i & 63) * 4096) & 0xfffff, 0, 0));
        if (LDSR && (i % (LDSR ? LDSR : 1)) == 0) rd += *reinterpret_cast<f32x4 *>(&lds[threadIdx.x * 4]);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    float s = rd[0];
    for (int i = 0; i < 8; ++i) s += vm[i][0];
    for (int i = 0; i < 8; ++i) s += e[i] + d[i];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int NEXP, int NADD, int LDSW, int LDSR, int VMEM = 0, int OCC = 1>
int run() {
    float *out; unsigned long long *cyc;
    const int blocks = 256, threads = 256, waves = blocks * 4;
    CHECK(hipMalloc(&out, blocks * threads * 4));
    CHECK(hipMalloc(&cyc, waves * 8));
    float *src; CHECK(hipMalloc(&src, 1 << 20)); CHECK(hipMemset(src, 0, 1 << 20));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<NACC, NEXP, NADD, LDSW, LDSR, VMEM, OCC>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f, src);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(waves);
    CHECK(hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("occ=%d vmem_every=%d chains=%d  exp/MFMA=%d add/MFMA=%d ds_write_b128 every %d ds_read_b128 every %d : median cycles/MFMA = %.1f (min %.1f)\n",
           OCC, VMEM, NACC, NEXP, NADD, LDSW, LDSR, h[waves / 2] / 256.0, h[0] / 256.0);
    (void)hipFree(src); (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}
#include <algorithm>
int main() {
    run<2, 0, 0, 0, 0>(); run<2, 0, 0, 0, 0, 0, 2>();
    run<2, 1, 0, 0, 0>(); run<2, 1, 0, 0, 0, 0, 2>();
    run<2, 0, 4, 0, 0>(); run<2, 0, 4, 0, 0, 0, 2>();
    run<2, 0, 0, 2, 0>(); run<2, 0, 0, 2, 0, 0, 2>();
    run<2, 0, 0, 0, 4>(); run<2, 0, 0, 0, 4, 0, 2>();
    run<2, 0, 0, 0, 0, 1>(); run<2, 0, 0, 0, 0, 1, 2>();
    run<2, 0, 0, 0, 0, 2>(); run<2, 0, 0, 0, 0, 2, 2>();
    run<2, 0, 0, 0, 0, 4>(); run<2, 0, 0, 0, 0, 4, 2>();
    return 0;
}
