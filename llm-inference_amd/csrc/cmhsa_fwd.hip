// cmhsa_fwd.hip -- fp32 causal multi-head self-attention forward for MI355X (gfx950 / CDNA4).
//
// Written from scratch for wave64 + fp32 MFMA; it replaces the reference's CUDA kernel and
// launcher (reference torch_extensions/causal_multihead_self_attention/
// causal_multihead_self_attention.cu:64-462, "CMHSA.cu" below) behind the C ABI of include/cmhsa.h.
//
// Decomposition (see DESIGN.md for the why):
//   * one workgroup (4 wave64s, one per SIMD) = one 32-row query block of one (batch, head);
//   * the KV range [0, q_block_end) is cut into 32-key units; wave w takes units w, w+4, ...
//     (split-KV inside the workgroup), so even the T=1024, B=1 shape fills 256 CUs;
//   * S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x2_f32: with S^T the query sits on the
//     lane, so softmax statistics are per-lane scalars and the S^T accumulator registers are,
//     as they stand, the B operand of the PV product (no LDS round trip, no lane shuffles);
//   * each wave stages its own K/V units in a private, double-buffered LDS region
//     (K XOR-swizzled for conflict-free ds_read_b128), so the main loop has no barrier at all;
//   * online softmax in the log2 domain (Q pre-scaled by log2(e)/sqrt(64)), normalisation
//     deferred to the very end; the 4 partial (m, l, O) triples are merged through LDS.
//
// Function computed per (batch b, head h), exactly CMHSA.cu:64-405 / the naive formula of
// reference causal_self_attention.py:173-194:
//     O[b, :, 64h:64h+64] = softmax(mask(Q_h K_h^T) / 8) V_h ,  key j <= query i.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "cmhsa.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHeadDim = 64;            // CMHSA.cu:432
constexpr int kQBlock = 32;             // query rows per workgroup (one MFMA N-tile)
constexpr int kKeyUnit = 32;            // keys per unit (one MFMA M-tile)
constexpr int kWaves = 8;               // split-KV ways inside a workgroup; two waves per SIMD
constexpr int kThreads = kWaves * 64;
constexpr int kUnitFloats = kKeyUnit * kHeadDim;          // 2048 floats = 8 KiB
constexpr int kWaveLdsFloats = 2 * kUnitFloats;           // K + V = 16 KiB per wave
constexpr int kLdsBytes = kWaves * kWaveLdsFloats * 4;    // 128 KiB per workgroup
constexpr int kCombStride = 33;                           // O^T merge image: [64 d][32 q + 1 pad]
constexpr float kScaleLog2 = 0.18033688011112042f;        // log2(e) / sqrt(64)
constexpr float kRescaleThreshold = 16.f;                 // lazy rescale: keep the old max while it is within 2^16

struct Params {
    const float *q, *k, *v;
    float *o;
    int64_t q_rs, k_rs, v_rs, o_rs;     // row strides (floats)
    int64_t q_bs, k_bs, v_bs, o_bs;     // batch strides (floats)
    int seq_len;                        // N
    int num_heads;                      // H
    int bh;                             // batch * H
    int nqb;                            // ceil(N / 32)
    int xcd_map;                        // 1: blockIdx -> (bh, qb) keeps one (b,h) per XCD group
};

// key/d index held by accumulator register r of a 32x32 MFMA result in lane half `half`
// (C/D map of v_mfma_f32_32x32x2_f32: row = (r&3) + 8*(r>>2) + 4*half, col = lane&31).
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

__device__ __forceinline__ float xhalf_max(float x) {
    // lanes l and l^32 exchange: v_permlane32_swap, then max
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Buffer descriptor over the <= 32 rows [row0, row0+32) of one head's 64-float column slab.  The
// hardware range check returns 0 for rows >= N (their byte offset is >= num_records), so ragged
// tails need no branches and masked keys can never inject NaN/garbage into 0 * V.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t unit_rsrc(const float *head_base, int64_t rs, int row0, int n_rows) {
    const int valid = min(kKeyUnit, n_rows - row0);       // >= 1 for every unit the kernel touches
    const int bytes = ((valid - 1) * (int)rs + kHeadDim) * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(head_base + (int64_t)row0 * rs), 0, bytes, 0x00020000);
}

// One 32-row x 64-float unit (8 KiB) global -> registers: lane l, step i reads the 16-B chunk
// (l & 15) of row 4i + (l >> 4): 16 lanes cover one 256-B row, fully coalesced.
// voff = per-lane byte offset of (row l>>4, chunk l&15); step = 4 rows in bytes.
__device__ __forceinline__ void load_unit(__amdgpu_buffer_rsrc_t rsrc, int voff, int step, f32x4 (&r)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
        r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + i * step, 0, 0));
}

// registers -> this wave's LDS unit.  K is XOR-swizzled per row (16-B chunk c of row r lives at
// chunk c ^ (r & 15)), V stays linear.
template <bool kSwizzle>
__device__ __forceinline__ void store_unit(float *lds, int lane, const f32x4 (&r)[8]) {
    const int sub = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + sub;
        const int c = kSwizzle ? (chunk ^ (row & 15)) : chunk;
        *reinterpret_cast<f32x4 *>(lds + row * kHeadDim + c * 4) = r[i];
    }
}

__global__ __launch_bounds__(kThreads, 2) void cmhsa_fwd_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;      // which k-slot of the MFMA operand this lane feeds
    const int l31 = lane & 31;       // query (B/C column) or key / d (A row) index of this lane

    // ---- blockIdx -> (batch*head, query block); heaviest query blocks first ----
    int bh, qb;
    {
        const int id = blockIdx.x;
        if (p.xcd_map) {             // blocks b and b+8 share an XCD: give each XCD group its own (b,h)
            const int xcd = id & 7, j = id >> 3;
            bh = (j / p.nqb) * 8 + xcd;
            qb = p.nqb - 1 - (j % p.nqb);
        } else {
            bh = id % p.bh;
            qb = p.nqb - 1 - id / p.bh;
        }
    }
    const int b = bh / p.num_heads, h = bh - b * p.num_heads;
    const int N = p.seq_len;
    const int q0 = qb * kQBlock;
    const int n_units = qb + 1;      // causal early exit: units past the diagonal are never touched (CMHSA.cu:133-136)

    const float *qg = p.q + b * p.q_bs + (int64_t)h * kHeadDim;
    const float *kg = p.k + b * p.k_bs + (int64_t)h * kHeadDim;
    const float *vg = p.v + b * p.v_bs + (int64_t)h * kHeadDim;

    float *wlds = lds + wave * kWaveLdsFloats;   // private to this wave: K | V
    float *kb = wlds, *vb = wlds + kUnitFloats;

    f32x16 o0 = {0}, o1 = {0};       // O^T accumulators: d = acc_row(r) + 4*half (+32 for o1), query = l31
    float m_run = -INFINITY;         // running max (log2 domain), same in both lane halves
    float l_run = 0.f;               // running sum over THIS lane half's keys only (merged at the end)

    if (wave < n_units) {
        // ---- Q^T fragment: B operand of S^T = K Q^T.  k-step s = 4c + j uses d = 8c + 4*half + j ----
        float qf[32];
        {
            const __amdgpu_buffer_rsrc_t qsrc = unit_rsrc(qg, p.q_rs, q0, N);
            const int qoff = (l31 * (int)p.q_rs + 4 * half) * 4;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qsrc, qoff + 32 * c, 0, 0));
#pragma unroll
                for (int j = 0; j < 4; ++j) qf[4 * c + j] = t[j] * kScaleLog2;
            }
        }

        const int sub = lane >> 4, chunk = lane & 15;
        const int koff = (sub * (int)p.k_rs + chunk * 4) * 4, kstep = 16 * (int)p.k_rs;
        const int voff = (sub * (int)p.v_rs + chunk * 4) * 4, vstep = 16 * (int)p.v_rs;

        f32x4 kst[8], vst[8];        // register staging of the next unit (prefetched one unit ahead)
        load_unit(unit_rsrc(kg, p.k_rs, wave * kKeyUnit, N), koff, kstep, kst);
        load_unit(unit_rsrc(vg, p.v_rs, wave * kKeyUnit, N), voff, vstep, vst);

        for (int u = wave; u < n_units; u += kWaves) {
            // ---- staged registers -> LDS (the previous unit's reads are all issued: in-order LDS queue) ----
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            store_unit<true>(kb, lane, kst);
            store_unit<false>(vb, lane, vst);
            if (u + kWaves < n_units) {
                load_unit(unit_rsrc(kg, p.k_rs, (u + kWaves) * kKeyUnit, N), koff, kstep, kst);
                load_unit(unit_rsrc(vg, p.v_rs, (u + kWaves) * kKeyUnit, N), voff, vstep, vst);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---- S^T[key][query] = sum_d K[key][d] * Q[query][d] : 32 MFMAs ----
            f32x16 s = {0};
            {
                const float *krow = kb + l31 * kHeadDim;
                const int sw = l31 & 15;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const f32x4 kk = *reinterpret_cast<const f32x4 *>(krow + (((2 * c + half) ^ sw) << 2));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[j], qf[4 * c + j], s, 0, 0, 0);
                }
            }

            // ---- causal mask: only the diagonal unit has key > query pairs (CMHSA.cu:228, :249) ----
            if (u == qb) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc_row(r) + 4 * half > l31) s[r] = -INFINITY;
            }

            // ---- online softmax, log2 domain; every processed unit has >=1 unmasked key per row.
            //      Lazy rescale: O and l are only rescaled when some row's max grew by more than
            //      2^kRescaleThreshold (fp32 has the exponent range; precision is unaffected). ----
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = xhalf_max(mx);
            if (!__all(mx <= m_run + kRescaleThreshold)) {       // always taken on the first unit (m_run = -inf)
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - m_run);
                psum += s[r];
            }
            l_run += psum;

            // ---- O^T[d][query] += sum_key V[key][d] * P[query][key] : 32 MFMAs.
            //      k-step r pairs keys acc_row(r) (lanes 0-31) and acc_row(r)+4 (lanes 32-63), which is
            //      exactly where accumulator register r of S^T holds them: P feeds the B operand as is.
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *vrow = vb + (acc_row(r) + 4 * half) * kHeadDim + l31;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], s[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], s[r], o1, 0, 0, 0);
            }
        }
        l_run = xhalf_sum(l_run);
    }

    // ---- merge the split-KV partials.  Each wave first writes into ITS OWN region, so no
    //      barrier is needed before the writes; idle waves publish (m=-inf, l=0, O=0). ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        float *ob = wlds;                                   // [64 d][33]
        float *ml = wlds + kHeadDim * kCombStride;          // m[32], l[32]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = acc_row(r) + 4 * half;
            ob[d * kCombStride + l31] = o0[r];
            ob[(d + 32) * kCombStride + l31] = o1[r];
        }
        if (lane < 32) {
            ml[lane] = m_run;
            ml[32 + lane] = l_run;
        }
    }
    __syncthreads();

    // thread -> (query = wave + 8i, d = lane): a wave stores one whole 256-B output row per step
    float *__restrict__ og = p.o + b * p.o_bs + (int64_t)h * kHeadDim;
#pragma unroll
    for (int i = 0; i < kQBlock / kWaves; ++i) {
        const int q = wave + kWaves * i;
        if (q0 + q >= N) break;                              // wave-uniform
        float mw[kWaves], lw[kWaves];
        float m_tot = -INFINITY;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const float *ml = lds + w * kWaveLdsFloats + kHeadDim * kCombStride;
            mw[w] = ml[q];
            lw[w] = ml[32 + q];
            m_tot = fmaxf(m_tot, mw[w]);
        }
        float acc = 0.f, l_tot = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const float f = __builtin_amdgcn_exp2f(mw[w] - m_tot);
            l_tot += lw[w] * f;
            acc += lds[w * kWaveLdsFloats + lane * kCombStride + q] * f;
        }
        og[(int64_t)(q0 + q) * p.o_rs + lane] = acc / l_tot;
    }
}

std::atomic<int> g_attr_set[64];     // per device: dynamic-LDS limit raised once, not per call

}  // namespace

extern "C" {

const char *cmhsa_status_string(int status) {
    switch (status) {
        case CMHSA_OK: return "success";
        case CMHSA_ERR_NULL_POINTER: return "a tensor pointer is NULL";
        case CMHSA_ERR_BAD_SHAPE: return "batch, seq_len and num_heads must be non-negative and num_heads positive";
        case CMHSA_ERR_HEAD_DIM: return "Head dimension must be 64.";
        case CMHSA_ERR_ALIGNMENT: return "pointers must be 16-byte aligned and row/batch strides multiples of 4 for float4 loads to work";
        case CMHSA_ERR_TOO_LARGE: return "problem too large for 32-bit tile indexing";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown cmhsa status";
    }
}

int cmhsa_version(void) { return (1 << 16) | 0; }

double cmhsa_algorithmic_flops(int batch, int seq_len, int num_heads, int head_dim) {
    return 2.0 * batch * num_heads * head_dim * (double)seq_len * ((double)seq_len + 1.0);
}

int cmhsa_fp32_fwd(const float *q, const float *k, const float *v, float *o,
                   int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                   int64_t q_batch_stride, int64_t k_batch_stride, int64_t v_batch_stride, int64_t o_batch_stride,
                   int batch, int seq_len, int num_heads, int head_dim, void *stream) {
    if (batch < 0 || seq_len < 0 || num_heads <= 0) return CMHSA_ERR_BAD_SHAPE;
    if (head_dim != kHeadDim) return CMHSA_ERR_HEAD_DIM;
    if (batch == 0 || seq_len == 0) return CMHSA_OK;
    if (!q || !k || !v || !o) return CMHSA_ERR_NULL_POINTER;
    if (batch == 1) q_batch_stride = k_batch_stride = v_batch_stride = o_batch_stride = 0;
    const uintptr_t ptr_bits = (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o;
    const int64_t stride_bits = q_row_stride | k_row_stride | v_row_stride | o_row_stride |
                                q_batch_stride | k_batch_stride | v_batch_stride | o_batch_stride;
    if ((ptr_bits & 15) || (stride_bits & 3)) return CMHSA_ERR_ALIGNMENT;

    const int64_t nqb = ((int64_t)seq_len + kQBlock - 1) / kQBlock;
    const int64_t bh = (int64_t)batch * num_heads;
    const int64_t blocks = bh * nqb;
    if (blocks > 0x7fffffffLL || bh > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;

    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64 || !g_attr_set[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(cmhsa_fwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 64) g_attr_set[dev].store(1, std::memory_order_release);
    }

    Params p;
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.q_rs = q_row_stride; p.k_rs = k_row_stride; p.v_rs = v_row_stride; p.o_rs = o_row_stride;
    p.q_bs = q_batch_stride; p.k_bs = k_batch_stride; p.v_bs = v_batch_stride; p.o_bs = o_batch_stride;
    p.seq_len = seq_len;
    p.num_heads = num_heads;
    p.bh = (int)bh;
    p.nqb = (int)nqb;
    p.xcd_map = (bh % 8 == 0) ? 1 : 0;

    hipLaunchKernelGGL(cmhsa_fwd_kernel, dim3((unsigned)blocks), dim3(kThreads), kLdsBytes,
                       static_cast<hipStream_t>(stream), p);
    e = hipGetLastError();                 // launch-configuration errors only; no synchronisation (CMHSA.cu:450)
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

int cmhsa_fp32_fwd_2d(const float *q, const float *k, const float *v, float *o,
                      int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                      int seq_len, int d_model, int num_heads, void *stream) {
    if (num_heads <= 0 || d_model < 0 || d_model % num_heads != 0) return CMHSA_ERR_BAD_SHAPE;
    return cmhsa_fp32_fwd(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, 0, 0, 0, 0,
                          1, seq_len, num_heads, d_model / num_heads, stream);
}

}  // extern "C"
