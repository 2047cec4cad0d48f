// add_layernorm_fwd.hip -- fused residual add + LayerNorm forward for MI355X (gfx950), C ABI in include/gpt2_layernorm.h.
//
// One wave64 per row, four rows per workgroup: a GPT-2 row (768..1600 floats) is 3..7 float4 per lane, held in
// registers for both statistics passes (sum -> mean; sum of squared deviations -> variance), so every byte is read
// once and written once.  In the generation loop the tensors are a few hundred KB: the kernel is launch-latency bound
// and exists to take two launches (add, LayerNorm) down to one.  Reductions: six xor-shuffle steps across the wave.
// Arithmetic as torch.nn.LayerNorm: biased variance, rsqrt(var + eps), y = weight * (rstd * (s - mean)) + bias.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmhsa.h"
#include "gpt2_layernorm.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowsPerWg = 4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// kVec = float4 per lane kept in registers: cols <= kVec * 256
template <int kVec, bool kDelta>
__global__ __launch_bounds__(kRowsPerWg * 64) void add_layernorm_kernel(const float *x, const float *delta, const float *weight,
                                                                        const float *bias, float *sum_out, float *norm_out,
                                                                        int64_t rows, int cols, int64_t x_rs, int64_t d_rs, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kRowsPerWg + (threadIdx.x >> 6);
    if (row >= rows) return;                       // wave-uniform; no barrier in this kernel
    const int nvec = cols >> 2;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x + row * x_rs);
    const f32x4 *d4 = kDelta ? reinterpret_cast<const f32x4 *>(delta + row * d_rs) : nullptr;
    f32x4 v[kVec];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        const int idx = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < nvec) {
            v[i] = x4[idx];
            if (kDelta) v[i] += d4[idx];
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(sum) / (float)cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        if (lane + 64 * i < nvec) {
            const f32x4 c = v[i] - mean;
            sq += (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)cols + eps);
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(weight), *b4 = reinterpret_cast<const f32x4 *>(bias);
    f32x4 *s4 = (kDelta && sum_out) ? reinterpret_cast<f32x4 *>(sum_out + row * (int64_t)cols) : nullptr;
    f32x4 *y4 = reinterpret_cast<f32x4 *>(norm_out + row * (int64_t)cols);
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nvec) {
            if (s4) s4[idx] = v[i];
            y4[idx] = w4[idx] * (rstd * (v[i] - mean)) + b4[idx];
        }
    }
}

// Few rows (the B = 1 generation loop: <= 512 rows): ONE WORKGROUP per row, so that a launch is rows x 4 waves instead of
// rows waves (GPT2-XL, 256 rows of 1600: one wave per row measured slower than torch's add + LayerNorm pair).  The two
// statistics cross the four waves through LDS.
__device__ __forceinline__ float wg_sum(float v, float *slots, int wave) {
    v = wave_sum(v);
    __syncthreads();                               // (the slots of the previous reduction have been read)
    if ((threadIdx.x & 63) == 0) slots[wave] = v;
    __syncthreads();
    return (slots[0] + slots[1]) + (slots[2] + slots[3]);
}

template <int kVec, bool kDelta, bool kStream>
__global__ __launch_bounds__(256) void add_layernorm_row_per_wg_kernel(const float *x, const float *delta, const float *weight,
                                                                       const float *bias, float *sum_out, float *norm_out,
                                                                       int64_t rows, int cols, int64_t x_rs, int64_t d_rs, float eps) {
    __shared__ float slots[4];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int nvec = cols >> 2;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x + row * x_rs);
    const f32x4 *d4 = kDelta ? reinterpret_cast<const f32x4 *>(delta + row * d_rs) : nullptr;
    f32x4 v[kVec];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        const int idx = tid + 256 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < nvec) {
            v[i] = kStream ? __builtin_nontemporal_load(x4 + idx) : x4[idx];
            if (kDelta) v[i] += kStream ? __builtin_nontemporal_load(d4 + idx) : d4[idx];
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wg_sum(sum, slots, wave) / (float)cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        if (tid + 256 * i < nvec) {
            const f32x4 c = v[i] - mean;
            sq += (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
        }
    }
    const float rstd = rsqrtf(wg_sum(sq, slots, wave) / (float)cols + eps);
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(weight), *b4 = reinterpret_cast<const f32x4 *>(bias);
    f32x4 *s4 = (kDelta && sum_out) ? reinterpret_cast<f32x4 *>(sum_out + row * (int64_t)cols) : nullptr;
    f32x4 *y4 = reinterpret_cast<f32x4 *>(norm_out + row * (int64_t)cols);
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        const int idx = tid + 256 * i;
        if (idx < nvec) {
            const f32x4 out = w4[idx] * (rstd * (v[i] - mean)) + b4[idx];
            if (kStream) {                         // arrays beyond the Infinity Cache: every byte is touched once
                if (s4) __builtin_nontemporal_store(v[i], s4 + idx);
                __builtin_nontemporal_store(out, y4 + idx);
            } else {
                if (s4) s4[idx] = v[i];
                y4[idx] = out;
            }
        }
    }
}

template <int kVec>
hipError_t launch_row_per_wg(bool with_delta, bool streaming, void **args, int64_t rows, hipStream_t s) {
    const void *k = with_delta ? (streaming ? reinterpret_cast<const void *>(add_layernorm_row_per_wg_kernel<kVec, true, true>)
                                            : reinterpret_cast<const void *>(add_layernorm_row_per_wg_kernel<kVec, true, false>))
                               : (streaming ? reinterpret_cast<const void *>(add_layernorm_row_per_wg_kernel<kVec, false, true>)
                                            : reinterpret_cast<const void *>(add_layernorm_row_per_wg_kernel<kVec, false, false>));
    return hipLaunchKernel(k, dim3((unsigned)rows), dim3(256), args, 0, s);
}

template <int kVec>
hipError_t launch(bool with_delta, void **args, int64_t rows, hipStream_t s) {
    const void *k = with_delta ? reinterpret_cast<const void *>(add_layernorm_kernel<kVec, true>)
                               : reinterpret_cast<const void *>(add_layernorm_kernel<kVec, false>);
    return hipLaunchKernel(k, dim3((unsigned)((rows + kRowsPerWg - 1) / kRowsPerWg)), dim3(kRowsPerWg * 64), args, 0, s);
}

}  // namespace

extern "C" {

double gpt2_add_layernorm_algorithmic_bytes(int64_t rows, int cols, int with_delta) {
    return (double)rows * cols * 4.0 * (with_delta ? 4.0 : 2.0) + 8.0 * cols;
}

int gpt2_add_layernorm_f32_fwd(const float *x, const float *delta, const float *weight, const float *bias,
                               float *sum_out, float *norm_out, int64_t rows, int cols,
                               int64_t x_row_stride, int64_t delta_row_stride, float eps, void *stream) {
    if (rows < 0 || cols < 4 || cols > 4096 || (cols & 3) || x_row_stride < 0 || delta_row_stride < 0) return CMHSA_ERR_BAD_SHAPE;
    if (rows == 0) return CMHSA_OK;
    if (!x || !weight || !bias || !norm_out || (delta && !sum_out)) return CMHSA_ERR_NULL_POINTER;
    if (rows >= (1LL << 31)) return CMHSA_ERR_TOO_LARGE;
    const uintptr_t bits = (uintptr_t)x | (uintptr_t)delta | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)sum_out | (uintptr_t)norm_out;
    if ((bits & 15) || ((x_row_stride | delta_row_stride) & 3)) return CMHSA_ERR_ALIGNMENT;
    void *args[] = {&x, &delta, &weight, &bias, &sum_out, &norm_out, &rows, &cols, &x_row_stride, &delta_row_stride, &eps};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool d = delta != nullptr;
    const int nvec_per_lane = (cols / 4 + 63) / 64;                // float4 per lane
    if (cols >= 512) {                                             // wide rows: one workgroup per row
        const int per_thread = (cols / 4 + 255) / 256;
        const bool st = (double)rows * cols * 4.0 * (d ? 4.0 : 2.0) > 128.0 * 1048576.0;      // larger than half the Infinity Cache
        const hipError_t e1 = per_thread <= 1 ? launch_row_per_wg<1>(d, st, args, rows, s) : per_thread <= 2 ? launch_row_per_wg<2>(d, st, args, rows, s)
                                                                                                             : launch_row_per_wg<4>(d, st, args, rows, s);
        return e1 == hipSuccess ? CMHSA_OK : (int)e1;
    }
    const hipError_t e = nvec_per_lane <= 4 ? launch<4>(d, args, rows, s) : nvec_per_lane <= 8 ? launch<8>(d, args, rows, s)
                                                                                              : launch<16>(d, args, rows, s);
    return e == hipSuccess ? CMHSA_OK : (int)e;     // this launch's own status
}

}  // extern "C"
