// linear_rows_fwd.hip -- y = act(x W^T + b) for a FEW rows of x (1..8), fp32, for MI355X (gfx950).
// C ABI: gpt2_linear_rows_f32_fwd in include/gpt2_linear.h.  The KV-cache decode loop's linear layers
// (reference causal_self_attention.py:224, :236, transformer_block.py:35-37 with one row per prompt).
//
// HBM-bound: the work is reading the n_out x k weight matrix once (7 MB for GPT2-small's QKV layer); no MFMA -- at one row
// there are 2 flops per weight byte.  One WAVE per output column n: its 64 lanes read the weight row n as 16-B pieces,
// lane l the pieces l, l + 64, ... (a wave instruction = 1 KiB contiguous), ALL of a row's pieces requested before the
// first is used (up to 16 per lane), so that a launch has the whole matrix in flight at once -- 5 MB in flight is what
// ~5 TB/s at ~1 us of latency takes.  x comes through the same piece pattern (L2 / L1 hits: every wave reads the same
// few KB).  Loads go through buffer descriptors sized k * 4 bytes: pieces past the end of a row return 0, no tail code.
// Each lane keeps one partial sum per row of x; the 64 partials are added in a fixed tree (DPP within 16 lanes, then
// across the four 16-lane rows); lane 0 adds the bias, applies the activation and stores.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmhsa.h"
#include "gpt2_linear.h"
#include "gelu_one.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWavesPerWg = 4;

template <int kCtrl>
__device__ __forceinline__ float dpp_add(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), kCtrl, 0xf, 0xf, false);
    return x + __builtin_bit_cast(float, y);
}

// sum over the 64 lanes of a wave, the same value in every lane
__device__ __forceinline__ float wave_sum(float x) {
    x = dpp_add<0xB1>(x);      // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);      // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);     // row_half_mirror
    x = dpp_add<0x140>(x);     // row_mirror: every lane of a 16-lane row holds the row's sum
    // the four row sums through scalar registers (v_readlane), not two more LDS-crossbar round trips per value
    const int bits = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float *row, int k) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row), 0, k * 4, 0x00020000);
}

struct LinearParams {
    const float *x, *w, *bias;
    float *y;
    int64_t x_rs, w_rs, y_rs;
    int n_out, k, act;
};

// kM rows of x (absent rows: descriptors of zero bytes, results not stored); kPieces 16-B pieces per lane per pass; a wave
// computes kCols consecutive output columns (absent ones: zero-byte descriptors), so that with several rows of x each
// piece of x fetched (from L1 / L2) serves kCols weight rows: at 8 rows and one column per wave the kernel is bound by
// those fetches, not by the weights (GPT2-small QKV: 6.7 us; the weights alone take 3.4).
template <int kM, int kPieces, int kCols>
__global__ __launch_bounds__(kWavesPerWg * 64) void linear_rows_kernel(LinearParams p, int rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * kWavesPerWg + wave) * kCols;
    if (n0 >= p.n_out) return;                                   // wave-uniform; the kernel has no barrier
    __amdgpu_buffer_rsrc_t wr[kCols], xr[kM];
#pragma unroll
    for (int c = 0; c < kCols; ++c)
        wr[c] = row_rsrc(p.w + (uint64_t)(uint32_t)min(n0 + c, p.n_out - 1) * (uint64_t)p.w_rs, n0 + c < p.n_out ? p.k : 0);
#pragma unroll
    for (int m = 0; m < kM; ++m) xr[m] = row_rsrc(p.x + (uint64_t)(uint32_t)min(m, rows - 1) * (uint64_t)p.x_rs, m < rows ? p.k : 0);
    float acc[kCols][kM];
#pragma unroll
    for (int c = 0; c < kCols; ++c)
#pragma unroll
        for (int m = 0; m < kM; ++m) acc[c][m] = 0.f;
    for (int piece0 = 0; piece0 * 4 < p.k; piece0 += 64 * kPieces) {
        f32x4 wv[kCols][kPieces];
#pragma unroll
        for (int u = 0; u < kPieces; ++u)
#pragma unroll
            for (int c = 0; c < kCols; ++c)
                wv[c][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr[c], (piece0 + u * 64 + lane) * 16, 0, 0));
#pragma unroll
        for (int u = 0; u < kPieces; ++u) {
#pragma unroll
            for (int m = 0; m < kM; ++m) {
                const f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr[m], (piece0 + u * 64 + lane) * 16, 0, 0));
                // one FMA chain per (column, row), spelled out: the same bits whatever the rows and columns beside it
#pragma unroll
                for (int c = 0; c < kCols; ++c)
                    acc[c][m] = __builtin_fmaf(wv[c][u].w, xv.w, __builtin_fmaf(wv[c][u].z, xv.z, __builtin_fmaf(wv[c][u].y, xv.y, __builtin_fmaf(wv[c][u].x, xv.x, acc[c][m]))));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < kCols; ++c) {
        const int n = n0 + c;
        if (n >= p.n_out) break;                                 // wave-uniform
        const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            float s = wave_sum(acc[c][m]) + b;
            if (p.act == GPT2_LINEAR_ACT_GELU_TANH) s = gelu_one(s);
            if (lane == 0 && m < rows) p.y[(int64_t)m * p.y_rs + n] = s;
        }
    }
}

// Few output columns, long rows (the second MLP layer: 768 x 3072, 1600 x 6400): one wave per column leaves most of the
// chip without a wave and each wave with a long row.  Here kSplit waves of a workgroup share a column, each a contiguous
// slice of k (slice_pieces 16-B pieces, a multiple of 64); the kSplit partial sums meet in LDS and are added in slice order.
template <int kM, int kPieces, int kSplit>
__global__ __launch_bounds__(kWavesPerWg * 64) void linear_rows_split_kernel(LinearParams p, int rows, int slice_pieces) {
    __shared__ float part[kWavesPerWg][kM];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slice = wave % kSplit;
    const int n = blockIdx.x * (kWavesPerWg / kSplit) + wave / kSplit;
    const bool live = n < p.n_out;                               // a dead wave loads nothing (zero-byte descriptors) but meets the barrier
    const int k0 = slice * slice_pieces * 4;
    const int klen = live ? max(0, min(p.k - k0, slice_pieces * 4)) : 0;
    const __amdgpu_buffer_rsrc_t wr = row_rsrc(p.w + (uint64_t)(uint32_t)min(n, p.n_out - 1) * (uint64_t)p.w_rs + k0, klen);
    __amdgpu_buffer_rsrc_t xr[kM];
#pragma unroll
    for (int m = 0; m < kM; ++m) xr[m] = row_rsrc(p.x + (uint64_t)(uint32_t)min(m, rows - 1) * (uint64_t)p.x_rs + k0, m < rows ? klen : 0);
    float acc[kM];
#pragma unroll
    for (int m = 0; m < kM; ++m) acc[m] = 0.f;
    for (int piece0 = 0; piece0 < slice_pieces; piece0 += 64 * kPieces) {
        f32x4 wv[kPieces];
#pragma unroll
        for (int u = 0; u < kPieces; ++u)
            wv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (piece0 + u * 64 + lane) * 16, 0, 0));
#pragma unroll
        for (int u = 0; u < kPieces; ++u) {
#pragma unroll
            for (int m = 0; m < kM; ++m) {
                const f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr[m], (piece0 + u * 64 + lane) * 16, 0, 0));
                acc[m] = __builtin_fmaf(wv[u].w, xv.w, __builtin_fmaf(wv[u].z, xv.z, __builtin_fmaf(wv[u].y, xv.y, __builtin_fmaf(wv[u].x, xv.x, acc[m]))));
            }
        }
    }
#pragma unroll
    for (int m = 0; m < kM; ++m) {
        const float t = wave_sum(acc[m]);
        if (lane == 0) part[wave][m] = t;
    }
    __syncthreads();
    if (slice == 0 && live && lane == 0) {
        const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            float t = part[wave][m];
#pragma unroll
            for (int j = 1; j < kSplit; ++j) t += part[wave + j][m];
            t += b;
            if (p.act == GPT2_LINEAR_ACT_GELU_TANH) t = gelu_one(t);
            if (m < rows) p.y[(int64_t)m * p.y_rs + n] = t;
        }
    }
}

template <int kM, int kSplit>
hipError_t launch_split(const LinearParams &p, int rows, hipStream_t s) {
    int slice_pieces = ((p.k / 4 + kSplit - 1) / kSplit + 63) / 64 * 64;
    const int per_lane = slice_pieces / 64;
    const void *fn = per_lane <= 4 ? reinterpret_cast<const void *>(linear_rows_split_kernel<kM, 4, kSplit>)
                     : (per_lane <= 8 || kM > 2) ? reinterpret_cast<const void *>(linear_rows_split_kernel<kM, 8, kSplit>)
                                                 : reinterpret_cast<const void *>(linear_rows_split_kernel<kM, 16, kSplit>);
    const int per_wg = kWavesPerWg / kSplit;
    const dim3 grid((unsigned)((p.n_out + per_wg - 1) / per_wg)), block(kWavesPerWg * 64);
    LinearParams pp = p;
    void *args[] = {&pp, &rows, &slice_pieces};
    return hipLaunchKernel(fn, grid, block, args, 0, s);
}

template <int kM>
hipError_t launch_rows(const LinearParams &p, int rows, hipStream_t s) {
    // pieces per lane per pass: the whole row in one pass up to k = 4096 (one and two rows of x), 2048 beyond two rows
    const int per_lane = (p.k / 4 + 63) / 64;
    // waves per column: a function of the layer's shape ONLY (a row of x must give the same bits whatever the rows beside it)
    if (p.n_out < 1024 && per_lane >= 8) return launch_split<kM, 4>(p, rows, s);
    if (p.n_out < 2048 && per_lane >= 4) return launch_split<kM, 2>(p, rows, s);
    const int pieces = per_lane <= 4 ? 4 : (per_lane <= 8 || kM > 2) ? 8 : 16;
    // columns per wave: one for a single row of x (the weights are all there is to fetch: as many waves as possible);
    // otherwise up to 16 weight pieces per lane in flight, as long as ~1024 waves remain
    int cols = 1;
    if (kM > 1) {
        cols = 16 / pieces > 4 ? 4 : 16 / pieces;
        while (cols > 1 && p.n_out / cols < 1024) cols /= 2;
    }
    const void *fn = nullptr;
    if (pieces == 4) fn = cols == 4 ? reinterpret_cast<const void *>(linear_rows_kernel<kM, 4, 4>)
                        : cols == 2 ? reinterpret_cast<const void *>(linear_rows_kernel<kM, 4, 2>)
                                    : reinterpret_cast<const void *>(linear_rows_kernel<kM, 4, 1>);
    else if (pieces == 8) fn = cols == 2 ? reinterpret_cast<const void *>(linear_rows_kernel<kM, 8, 2>)
                                         : reinterpret_cast<const void *>(linear_rows_kernel<kM, 8, 1>);
    else fn = reinterpret_cast<const void *>(linear_rows_kernel<kM, 16, 1>);
    const int per_wg = kWavesPerWg * cols;
    const dim3 grid((unsigned)((p.n_out + per_wg - 1) / per_wg)), block(kWavesPerWg * 64);
    LinearParams pp = p;
    void *args[] = {&pp, &rows};
    // the launch's own status (hipGetLastError would also report, and clear, an unrelated earlier error of this thread)
    return hipLaunchKernel(fn, grid, block, args, 0, s);
}

// ---- the same with the residual add and the LayerNorm of x in front (the step's QKV and first MLP layer):
//      s = x + delta, h = LayerNorm(s) (torch.nn.LayerNorm arithmetic: biased variance, rsqrt(var + eps),
//      lnw * (rstd * (s - mean)) + lnb), y = act(h W^T + b).  Every wave reads whole rows of x anyway (lane l the pieces
//      l, l + 64, ...), so each computes the two statistics itself -- two wave reductions per row -- instead of a
//      launch of its own in front (4 us for one row of 768: latency, not bytes).  The first workgroup's first wave
//      stores s for the residual stream.  One pass: k <= 64 lanes * kPieces * 4.
struct LnLinearParams {
    LinearParams lin;
    const float *delta, *lnw, *lnb;
    float *sum_out;
    int64_t delta_rs;
    float eps;
};

template <int kM, int kPieces>
__global__ __launch_bounds__(kWavesPerWg * 64) void ln_linear_rows_kernel(LnLinearParams q, int rows) {
    const LinearParams &p = q.lin;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * kWavesPerWg + wave;
    if (n >= p.n_out) return;                                    // wave-uniform; the kernel has no barrier
    const __amdgpu_buffer_rsrc_t wr = row_rsrc(p.w + (uint64_t)(uint32_t)n * (uint64_t)p.w_rs, p.k);
    f32x4 wv[kPieces];
#pragma unroll
    for (int u = 0; u < kPieces; ++u) wv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (u * 64 + lane) * 16, 0, 0));
    const __amdgpu_buffer_rsrc_t gr = row_rsrc(q.lnw, p.k), br = row_rsrc(q.lnb, p.k);
    f32x4 g[kPieces], be[kPieces];
#pragma unroll
    for (int u = 0; u < kPieces; ++u) {
        g[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, (u * 64 + lane) * 16, 0, 0));
        be[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(br, (u * 64 + lane) * 16, 0, 0));
    }
    const float b = p.bias ? p.bias[n] : 0.f;
    const float inv_k = 1.0f / (float)p.k;
#pragma unroll
    for (int m = 0; m < kM; ++m) {
        if (m >= rows) break;                                    // wave-uniform
        const __amdgpu_buffer_rsrc_t xr = row_rsrc(p.x + (uint64_t)(uint32_t)m * (uint64_t)p.x_rs, p.k);
        const __amdgpu_buffer_rsrc_t dr = row_rsrc(q.delta ? q.delta + (uint64_t)(uint32_t)m * (uint64_t)q.delta_rs : p.x, q.delta ? p.k : 0);
        f32x4 sv[kPieces];
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < kPieces; ++u) {
            const f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (u * 64 + lane) * 16, 0, 0));
            const f32x4 dv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, (u * 64 + lane) * 16, 0, 0));
            sv[u] = xv + dv;                                     // pieces past the end of the row: 0 + 0
            sum += (sv[u].x + sv[u].y) + (sv[u].z + sv[u].w);
        }
        const float mean = wave_sum(sum) * inv_k;
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < kPieces; ++u) {
            const bool in_row = (u * 64 + lane) * 4 < p.k;
            const f32x4 c = sv[u] - mean;
            sq += in_row ? (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w) : 0.f;
        }
        const float rstd = rsqrtf(wave_sum(sq) * inv_k + q.eps);
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < kPieces; ++u) {
            const f32x4 h = g[u] * (rstd * (sv[u] - mean)) + be[u];            // 0 * (..) + 0 past the end of the row
            acc = __builtin_fmaf(wv[u].w, h.w, __builtin_fmaf(wv[u].z, h.z, __builtin_fmaf(wv[u].y, h.y, __builtin_fmaf(wv[u].x, h.x, acc))));
        }
        if (q.sum_out && blockIdx.x == 0 && wave == 0) {
            const __amdgpu_buffer_rsrc_t sr = row_rsrc(q.sum_out + (uint64_t)(uint32_t)m * (uint64_t)p.k, p.k);
#pragma unroll
            for (int u = 0; u < kPieces; ++u)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv[u]), sr, (u * 64 + lane) * 16, 0, 0);   // out of range: dropped
        }
        float r = wave_sum(acc) + b;
        if (p.act == GPT2_LINEAR_ACT_GELU_TANH) r = gelu_one(r);
        if (lane == 0) p.y[(int64_t)m * p.y_rs + n] = r;
    }
}

}  // namespace

extern "C" {

int gpt2_ln_linear_rows_f32_fwd(const float *x, int64_t x_row_stride, const float *delta, int64_t delta_row_stride,
                                const float *ln_weight, const float *ln_bias, float eps, float *sum_out,
                                const float *weight, int64_t w_row_stride, const float *bias, float *y, int64_t y_row_stride,
                                int rows, int n_out, int k, int act, void *stream) {
    if (rows < 0 || n_out < 0 || k <= 0 || (k & 3) || rows > GPT2_LN_LINEAR_MAX_ROWS || k > GPT2_LN_LINEAR_MAX_K) return CMHSA_ERR_BAD_SHAPE;
    if (act != GPT2_LINEAR_ACT_NONE && act != GPT2_LINEAR_ACT_GELU_TANH) return CMHSA_ERR_BAD_SHAPE;
    if (x_row_stride < 0 || delta_row_stride < 0 || w_row_stride < 0 || y_row_stride < 0) return CMHSA_ERR_BAD_SHAPE;
    if (rows == 0 || n_out == 0) return CMHSA_OK;
    if (!x || !ln_weight || !ln_bias || !weight || !y || (delta && !sum_out)) return CMHSA_ERR_NULL_POINTER;
    if ((((uintptr_t)x | (uintptr_t)delta | (uintptr_t)ln_weight | (uintptr_t)ln_bias | (uintptr_t)sum_out | (uintptr_t)weight) & 15) ||
        ((x_row_stride | delta_row_stride | w_row_stride) & 3) || (((uintptr_t)y | (uintptr_t)bias) & 3))
        return CMHSA_ERR_ALIGNMENT;
    if (rows > 1 && (x_row_stride < k || (delta && delta_row_stride < k) || y_row_stride < n_out)) return CMHSA_ERR_BAD_SHAPE;
    if (w_row_stride < k && n_out > 1) return CMHSA_ERR_BAD_SHAPE;
    LnLinearParams q;
    q.lin.x = x; q.lin.w = weight; q.lin.bias = bias; q.lin.y = y;
    q.lin.x_rs = x_row_stride; q.lin.w_rs = w_row_stride; q.lin.y_rs = y_row_stride;
    q.lin.n_out = n_out; q.lin.k = k; q.lin.act = act;
    q.delta = delta; q.delta_rs = delta_row_stride; q.lnw = ln_weight; q.lnb = ln_bias; q.sum_out = sum_out; q.eps = eps;
    const int per_lane = (k / 4 + 63) / 64;
    const void *fn = nullptr;
    if (per_lane <= 4) fn = rows == 1 ? reinterpret_cast<const void *>(ln_linear_rows_kernel<1, 4>) : rows == 2 ? reinterpret_cast<const void *>(ln_linear_rows_kernel<2, 4>)
                                                                                                              : reinterpret_cast<const void *>(ln_linear_rows_kernel<4, 4>);
    else if (per_lane <= 8) fn = rows == 1 ? reinterpret_cast<const void *>(ln_linear_rows_kernel<1, 8>) : rows == 2 ? reinterpret_cast<const void *>(ln_linear_rows_kernel<2, 8>)
                                                                                                                   : reinterpret_cast<const void *>(ln_linear_rows_kernel<4, 8>);
    else fn = rows == 1 ? reinterpret_cast<const void *>(ln_linear_rows_kernel<1, 16>) : rows == 2 ? reinterpret_cast<const void *>(ln_linear_rows_kernel<2, 16>)
                                                                                                 : reinterpret_cast<const void *>(ln_linear_rows_kernel<4, 16>);
    const dim3 grid((unsigned)((n_out + kWavesPerWg - 1) / kWavesPerWg)), block(kWavesPerWg * 64);
    void *args[] = {&q, &rows};
    const hipError_t e = hipLaunchKernel(fn, grid, block, args, 0, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

double gpt2_linear_rows_algorithmic_bytes(int rows, int n_out, int k, int with_bias) {
    return 4.0 * ((double)n_out * k + (with_bias ? n_out : 0) + (double)rows * k + (double)rows * n_out);
}

int gpt2_linear_rows_f32_fwd(const float *x, int64_t x_row_stride, const float *weight, int64_t w_row_stride,
                             const float *bias, float *y, int64_t y_row_stride,
                             int rows, int n_out, int k, int act, void *stream) {
    if (rows < 0 || n_out < 0 || k <= 0 || (k & 3) || rows > GPT2_LINEAR_MAX_ROWS) return CMHSA_ERR_BAD_SHAPE;
    if (act != GPT2_LINEAR_ACT_NONE && act != GPT2_LINEAR_ACT_GELU_TANH) return CMHSA_ERR_BAD_SHAPE;
    if (x_row_stride < 0 || w_row_stride < 0 || y_row_stride < 0) return CMHSA_ERR_BAD_SHAPE;
    if (rows == 0 || n_out == 0) return CMHSA_OK;
    if (!x || !weight || !y) return CMHSA_ERR_NULL_POINTER;
    if ((((uintptr_t)x | (uintptr_t)weight) & 15) || ((x_row_stride | w_row_stride) & 3) || (((uintptr_t)y | (uintptr_t)bias) & 3))
        return CMHSA_ERR_ALIGNMENT;
    if (x_row_stride < k && rows > 1) return CMHSA_ERR_BAD_SHAPE;
    if (w_row_stride < k && n_out > 1) return CMHSA_ERR_BAD_SHAPE;
    if (y_row_stride < n_out && rows > 1) return CMHSA_ERR_BAD_SHAPE;
    // 32-bit byte offsets inside a row (buffer loads); 32-bit piece counter
    if ((int64_t)k * 4 + 64 * 16 * 16 > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;
    LinearParams p;
    p.x = x; p.w = weight; p.bias = bias; p.y = y;
    p.x_rs = x_row_stride; p.w_rs = w_row_stride; p.y_rs = y_row_stride;
    p.n_out = n_out; p.k = k; p.act = act;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (rows == 1) e = launch_rows<1>(p, rows, s);
    else if (rows == 2) e = launch_rows<2>(p, rows, s);
    else if (rows <= 4) e = launch_rows<4>(p, rows, s);
    else e = launch_rows<8>(p, rows, s);
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

}  // extern "C"
