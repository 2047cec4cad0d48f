// cmhsa_fwd16.hip -- the "16-row blocks" family of the fp32 causal attention forward for MI355X (gfx950 / CDNA4).
//
// Same function as cmhsa_fwd.hip (reference torch_extensions/causal_multihead_self_attention/
// causal_multihead_self_attention.cu:64-405, "CMHSA.cu"; the naive formula of reference causal_self_attention.py:173-194):
//     O[b, :, 64h:64h+64] = softmax(mask(Q_h K_h^T) / 8) V_h ,  key j <= query i,
// decomposed for grids that fit the chip in ONE round (BASELINE config 2, and the B = 1 launches of a generation, config 3),
// where the kernel of cmhsa_fwd.hip pays for its 32 x 32 units: a query block's keys can only be spread over the four
// SIMDs of one CU, the work comes in quanta of 4 096 MFMA cycles, and the prologue (K and Q through LDS, a barrier)
// and merge are 7-9 k cycles of a 12-40 k cycle launch (profiles/r03_experiments/stamps_b1_h12_t*.txt).  Here:
//   * query blocks of 16 rows, key tiles of 16 rows, v_mfma_f32_16x16x4_f32 (same flop rate as the 32x32x2 shape): a
//     tile is 16 QK + 16 PV MFMAs = 1 024 cycles, a quarter of a unit, and a block's tiles need half the SIMD time;
//   * S^T = K Q^T again: the query sits on the lane (l % 16), the S^T accumulator registers are, as they stand, the B
//     operand of O^T += V^T P^T; row max / sum cross the four lane groups with v_permlane16_swap + v_permlane32_swap
//     (the wave64 counterpart of the reference's 16-lane __shfl_down_sync tree, CMHSA.cu:267-284);
//   * no LDS and no barrier in front of the first MFMA: K and Q fragments go straight from global memory into the
//     A / B operand layout (lane (g, r) reads the 16-B piece 4c + g of row r: every row's four lanes cover 64
//     contiguous bytes), V likewise (lane (g, r) reads floats 4r..4r+3 of key 4g + i: whole 256-B rows);
//   * a workgroup = four waves, one per SIMD.  Each wave runs a LIST of jobs from a host-made table (cmhsa_tiles16.h):
//     a job = a run of key tiles of one block.  The blocks of a workgroup are laid end to end and cut into four equal
//     runs, so all four SIMDs carry the same number of tiles; a block cut by a run boundary is merged through LDS
//     behind the kernel's only barrier, a whole block is normalised and stored straight from the accumulators;
//   * K / V tiles are prefetched two steps (of two tiles) ahead into registers, across job boundaries; the next job's
//     query fragments are requested at the start of the current job;
//   * online softmax in the log2 domain, deferred normalisation, speculative lazy rescale -- as in cmhsa_fwd.hip.
// Causal early exit (CMHSA.cu:133-136) = the table never names a tile past the diagonal one.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "cmhsa.h"
#include "cmhsa_tiles16.h"

namespace cmhsa_t16 {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kHeadDim = 64;
constexpr float kScaleLog2 = 0.18033688011112042f;        // log2(e) / sqrt(64)
constexpr float kRescaleSum = 1.8446744e19f;              // 2^64: a row sum of P above it triggers the rescale path
// LDS: [wave][slot A, B][image: [4 rows i][64 lanes] float4 | (m, l) of the 16 queries]
constexpr int kImageFloats = 4 * 64 * 4;
constexpr int kSlotFloats = kImageFloats + 32;

struct Params16 {
    const float *q, *k, *v;
    float *o;
    int64_t q_bs, k_bs, v_bs, o_bs;     // batch strides (floats)
    unsigned q_rs, k_rs, v_rs, o_rs;    // row strides (floats; <= 2^24, host check)
    int seq_len, num_heads;
    int wgs_per_bh, total_wgs, per_xcd; // workgroups per (b,h); bh * wgs_per_bh; ceil(total_wgs / 8)
    unsigned heads_magic, wgs_magic;    // floor((2^32-1) / num_heads), / wgs_per_bh
    unsigned table[kTableWords];        // [workgroup of a (b,h)][wave][kJobs] job words (cmhsa_tiles16.h)
};

#ifdef CMHSA_STAMPS
__device__ unsigned long long *g_stamp_buf = nullptr;
__device__ int g_stamp_block = 0;
__device__ unsigned long long *g_block_times = nullptr;
#define STAMP(slot)                                                                              \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (stamp_on && (slot) < 128) stamp_ptr[(slot)] = t_;                                    \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int udiv_magic(int n, int d, unsigned magic) {
    const unsigned q = __umulhi((unsigned)n, magic);
    return (int)(q + (((unsigned)n - q * (unsigned)d) >= (unsigned)d ? 1u : 0u));
}

// lanes l, l ^ 16 (v_permlane16_swap) and l, l ^ 32 (v_permlane32_swap) exchange
__device__ __forceinline__ float x16_max(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float x32_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float x16_sum(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float x32_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Buffer descriptor over the rows [row0, row0 + rows) of one head's 64-float column slab, clipped to the N valid rows: the
// hardware range check returns 0 for the others (ragged tails need no branches, and a masked key can never put NaN or
// garbage into 0 * V).  enable = false: zero records, every load returns 0 and touches no memory.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const float *head_base, unsigned rs, int row0, int rows, int N, bool enable) {
    const int valid = min(rows, N - row0);
    const int bytes = (valid > 0 && enable) ? ((valid - 1) * (int)rs + kHeadDim) * 4 : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(head_base + (uint64_t)(uint32_t)row0 * rs), 0, bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 load16(__amdgpu_buffer_rsrc_t rsrc, int voffset) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, 0, 0));
}

template <int kJobs>
__global__ __launch_bounds__(kWaves * 64, 1) void fwd16_kernel(Params16 p) {
    __shared__ __attribute__((aligned(16))) float lds[kWaves * 2 * kSlotFloats];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, r = lane & 15;      // lane group (k index of both MFMA operands), row / query within the tile
#ifdef CMHSA_STAMPS
    const bool stamp_on = g_stamp_buf != nullptr && (int)blockIdx.x == g_stamp_block && lane == 0;
    unsigned long long *stamp_ptr = g_stamp_buf + wave * 128;
    int stamp_i = 4;
    STAMP(0);
    if (g_block_times && threadIdx.x == 0) g_block_times[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- workgroup -> (b,h), workgroup j of it.  Workgroups are dealt round-robin over the 8 XCDs in linear order, so
    //      XCD x gets the ids x, x + 8, ...: it is given a CONTIGUOUS run of the (b,h)-major workgroup list, so that a
    //      (b,h)'s K and V are read through one XCD's L2 (two at a boundary), whatever B*H is. ----
    const int id = blockIdx.x;
    const int pos = (id & 7) * p.per_xcd + (id >> 3);
    if (pos >= p.total_wgs) return;                       // padding of the last XCDs' runs (whole workgroup, before any barrier)
    const int bh = udiv_magic(pos, p.wgs_per_bh, p.wgs_magic), j = pos - bh * p.wgs_per_bh;
    // every kernel argument and this wave's job words in one batch of scalar loads (one wait)
    unsigned jw[kJobs];
#pragma unroll
    for (int i = 0; i < kJobs; ++i) jw[i] = p.table[(j * kWaves + wave) * kJobs + i];
    if constexpr (kJobs == 8)
        asm volatile("" ::"s"(jw[4]), "s"(jw[5]), "s"(jw[6]), "s"(jw[7]));
    asm volatile("" ::"s"(jw[0]), "s"(jw[1]), "s"(jw[2]), "s"(jw[3]), "s"(p.q), "s"(p.k), "s"(p.v), "s"(p.o), "s"(p.q_rs), "s"(p.k_rs),
                 "s"(p.v_rs), "s"(p.o_rs));
    asm volatile("" ::"s"(p.q_bs), "s"(p.k_bs), "s"(p.v_bs), "s"(p.o_bs), "s"(p.seq_len), "s"(p.num_heads), "s"(p.heads_magic));
    STAMP(124);

    const int b = udiv_magic(bh, p.num_heads, p.heads_magic), h = bh - b * p.num_heads;
    const int N = p.seq_len;
    const float *qg = p.q + (uint64_t)(uint32_t)b * (uint64_t)p.q_bs + (uint32_t)h * kHeadDim;
    const float *kg = p.k + (uint64_t)(uint32_t)b * (uint64_t)p.k_bs + (uint32_t)h * kHeadDim;
    const float *vg = p.v + (uint64_t)(uint32_t)b * (uint64_t)p.v_bs + (uint32_t)h * kHeadDim;
    float *__restrict__ og = p.o + (uint64_t)(uint32_t)b * (uint64_t)p.o_bs + (uint32_t)h * kHeadDim;

    // job i of this wave, i wave-uniform and dynamic: lane i of a register (lanes >= kJobs hold "no job")
    unsigned jobs_v = 0;
#pragma unroll
    for (int i = 0; i < kJobs; ++i) jobs_v = lane == i ? jw[i] : jobs_v;
    auto job = [&](int i) { return (unsigned)__builtin_amdgcn_readlane((int)jobs_v, i); };
    const bool wg_merges = (jw[0] & kWgMergesBit) != 0;
    int total_steps = 0;                                   // a step = two tiles of a job (its last step: one, if the run is odd)
    unsigned word_a = 0, word_b = 0;                       // this wave's pieces of cut blocks: a later piece (slot A), a first piece (slot B)
#pragma unroll
    for (int i = 0; i < kJobs; ++i) {
        const int cnt = (jw[i] >> 16) & 255;
        total_steps += (cnt + 1) >> 1;
        if (cnt && ((jw[i] >> 26) & 7) > 1) {
            if ((jw[i] >> 24) & 3) word_a = jw[i]; else word_b = jw[i];
        }
    }

    // per-lane byte offsets inside a step's 32 rows
    const int koff0 = (r * (int)p.k_rs + 4 * g) * 4, koff1 = koff0 + 16 * (int)p.k_rs * 4;       // K: row r (+16), piece 4c + g: + 64 c
    int voff[2][4];                                                                              // V: key 4g + i (+16), floats 4r..4r+3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        voff[0][i] = ((4 * g + i) * (int)p.v_rs + 4 * r) * 4;
        voff[1][i] = voff[0][i] + 16 * (int)p.v_rs * 4;
    }
    const int qoff = (r * (int)p.q_rs + 4 * g) * 4;

    // ---- prefetch cursor: the key tiles of the wave's jobs, in order, two at a time ----
    int pf_job = -1, pf_tile = 0, pf_end = 0;
    auto pf_take = [&](bool &en) {
        if (pf_tile >= pf_end) {
            pf_job = min(pf_job + 1, kJobs);                // (lane kJobs holds "no job": the cursor stays exhausted)
            const unsigned w = job(pf_job);
            pf_tile = (w >> 8) & 255;
            pf_end = pf_tile + ((w >> 16) & 255);
        }
        en = pf_tile < pf_end;
        const int row0 = pf_tile * 16;
        pf_tile += 2;
        return row0;
    };

    f32x4 kx[2][4], ky[2][4];        // K of two steps in flight: [tile of the step][c] = floats 16c + 4g .. +3 of key r
    f32x4 vx[2][4], vy[2][4];        // V:                       [tile][i] = floats 4r .. 4r+3 of key 4g + i
    f32x4 qn[4];                     // the NEXT job's query fragments, raw
    auto load_step = [&](f32x4 (&kb)[2][4], f32x4 (&vb)[2][4]) {
        bool en;
        const int row0 = pf_take(en);
        const __amdgpu_buffer_rsrc_t ks = rows_rsrc(kg, p.k_rs, row0, 32, N, en), vs = rows_rsrc(vg, p.v_rs, row0, 32, N, en);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kb[0][c] = load16(ks, koff0 + 64 * c);
            kb[1][c] = load16(ks, koff1 + 64 * c);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vb[0][i] = load16(vs, voff[0][i]);
            vb[1][i] = load16(vs, voff[1][i]);
        }
    };
    auto load_q = [&](unsigned w) {
        const bool en = ((w >> 16) & 255) != 0;
        const __amdgpu_buffer_rsrc_t qs = rows_rsrc(qg, p.q_rs, (int)(w & 255) * 16, 16, N, en);
#pragma unroll
        for (int c = 0; c < 4; ++c) qn[c] = load16(qs, qoff + 64 * c);
    };
    // one memory round trip ahead of the first MFMA: the first job's query fragments and the first two steps' K and V
    load_q(jw[0]);
    load_step(kx, vx);
    load_step(ky, vy);
    STAMP(121);

    float qf[16];                    // Q^T fragment of the current job, pre-scaled: k-step 4c + jj uses d = 16c + 4g + jj
    f32x4 acc[4];                    // O^T: acc[dt][i] <-> d = 16g + 4i + dt, query r
    float m_run = -INFINITY, l_run = 0.f;
    int cj = 0, ctile = 0, cend = 0, blk = 0;
    unsigned cw = 0;
    bool new_job = true;

    float *my_slots = lds + wave * 2 * kSlotFloats;

    auto do_step = [&](f32x4 (&kb)[2][4], f32x4 (&vb)[2][4]) __attribute__((always_inline)) {
        if (new_job) {
            cw = job(cj);
            blk = cw & 255;
            ctile = (cw >> 8) & 255;
            cend = ctile + ((cw >> 16) & 255);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) qf[4 * c + jj] = qn[c][jj] * kScaleLog2;
            load_q(job(min(cj + 1, kJobs)));               // in flight for the whole of this job
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run = -INFINITY;
            l_run = 0.f;
        }
        const bool first = new_job;
        new_job = false;
        const bool diag = cend - 1 == blk && ctile + 2 >= cend;     // the step ends with the block's diagonal tile
        // descriptors of the step after next (its K and V take this step's registers as they fall free)
        bool en;
        const int row0n = pf_take(en);
        const __amdgpu_buffer_rsrc_t ksn = rows_rsrc(kg, p.k_rs, row0n, 32, N, en), vsn = rows_rsrc(vg, p.v_rs, row0n, 32, N, en);
        // kTwo (compile time): the step has two tiles, A and B; else one (the last step of a run of odd length)
        auto body = [&](auto two_c) __attribute__((always_inline)) {
            constexpr bool kTwo = decltype(two_c)::value;
#ifdef CMHSA_STAMPS
            STAMP(stamp_i);
#endif
            // ---- S^T[key][query] = sum_d K[key][d] Q[query][d]: two independent chains (the shape's dependent latency
            //      is 40 cycles against 32 of issue): tiles A and B, or the even / odd k-steps of a single tile ----
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (kTwo) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        sa = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[0][c][jj], qf[4 * c + jj], sa, 0, 0, 0);
                        sb = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[1][c][jj], qf[4 * c + jj], sb, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; jj += 2) {
                        sa = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[0][c][jj], qf[4 * c + jj], sa, 0, 0, 0);
                        sb = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[0][c][jj + 1], qf[4 * c + jj + 1], sb, 0, 0, 0);
                    }
                }
            }
            if constexpr (!kTwo) sa += sb;
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 1);
#endif
            // ---- causal mask: only the diagonal tile has key > query pairs (CMHSA.cu:228, :249); keys past the sequence
            //      end exist in that tile only and are masked with it ----
            if (diag) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (4 * g + i > r) {
                        if constexpr (kTwo) sb[i] = -INFINITY; else sa[i] = -INFINITY;
                    }
                }
            }
            // ---- online softmax, log2 domain, speculative against the old running max (see cmhsa_fwd.hip): the max is
            //      only computed on a job's first step and when some row sum leaves [0, 2^64) ----
            f32x4 pa, pb = {0.f, 0.f, 0.f, 0.f};
            float psum = 0.f;
            bool redo = first;
            if (!first) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pa[i] = __builtin_amdgcn_exp2f(sa[i] - m_run);
                    psum += pa[i];
                    if constexpr (kTwo) {
                        pb[i] = __builtin_amdgcn_exp2f(sb[i] - m_run);
                        psum += pb[i];
                    }
                }
                redo = !__all(psum < kRescaleSum);          // NaN and inf compare false: they take the max path too
            }
            if (redo) {
                float mx = fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3]));
                if constexpr (kTwo) mx = fmaxf(mx, fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3])));
                mx = x32_max(x16_max(mx));
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 0 on a job's first step (m_run = -inf)
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[dt] *= alpha;
                psum = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pa[i] = __builtin_amdgcn_exp2f(sa[i] - m_new);
                    psum += pa[i];
                    if constexpr (kTwo) {
                        pb[i] = __builtin_amdgcn_exp2f(sb[i] - m_new);
                        psum += pb[i];
                    }
                }
            }
            l_run += psum;
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 2);
#endif
            // ---- O^T[d][query] += sum_key V[key][d] P[query][key]: k-step (tile, i) pairs the keys 4g + i of the four
            //      lane groups, which is where register i of the S^T accumulator holds them: P feeds the B operand as
            //      it stands.  Behind each k-step's four MFMAs: the loads of the step after next into the registers
            //      that just fell free (a one-tile step never touched its B registers: they are reloaded beside A's). ----
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[0][i][dt], pa[i], acc[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                vb[0][i] = load16(vsn, voff[0][i]);
                kb[0][i] = load16(ksn, koff0 + 64 * i);
                if constexpr (!kTwo) {
                    vb[1][i] = load16(vsn, voff[1][i]);
                    kb[1][i] = load16(ksn, koff1 + 64 * i);
                }
            }
            if constexpr (kTwo) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[1][i][dt], pb[i], acc[dt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    vb[1][i] = load16(vsn, voff[1][i]);
                    kb[1][i] = load16(ksn, koff1 + 64 * i);
                }
            }
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 3);
            stamp_i += 4;
#endif
        };
        if (cend - ctile >= 2) body(std::true_type{}); else body(std::false_type{});
        ctile += 2;
        if (ctile >= cend) {
            // ---- the job is done.  A whole block: normalise and store, lane (g, r) = query r, floats 16g + 4i .. + 3.
            //      A piece of a cut block: the un-normalised image and (m, l) go to one of the wave's two LDS slots
            //      (a piece that starts its block: slot B, a later piece: slot A) for the merge behind the barrier. ----
            const float l_tot = x32_sum(x16_sum(l_run));
            const int pieces = (cw >> 26) & 7;
            if (pieces <= 1) {
                const float inv = 1.0f / l_tot;
                const int row = blk * 16 + r;
                if (row < N) {
                    float *dst = og + (uint64_t)(uint32_t)row * p.o_rs + 16 * g;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<f32x4 *>(dst + 4 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]} * inv;
                }
            } else {
                float *slot = my_slots + (((cw >> 24) & 3) ? 0 : kSlotFloats);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4 *>(slot + (i * 64 + lane) * 4) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
                if (g == 0) *reinterpret_cast<f32x2 *>(slot + kImageFloats + 2 * r) = f32x2{m_run, l_tot};
            }
            ++cj;
            new_job = true;
        }
    };

    STAMP(1);
    for (int st = 0; st < total_steps; st += 2) {
        do_step(kx, vx);
        if (st + 1 < total_steps) do_step(ky, vy);
    }
    STAMP(2);

    // ---- merge of the cut blocks: the g pieces of a block sit on consecutive waves (piece 0 in slot B of its wave, the
    //      others in slot A of theirs); the wave of rank k combines rows i = k, k + g, ... of the image, pieces in order
    //      (the bits do not depend on timing), divides by the merged l and stores. ----
    if (wg_merges) {
        __syncthreads();
        STAMP(120);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned w = t == 0 ? word_a : word_b;
            if (!w) continue;                                                    // wave-uniform
            const int mblk = w & 255, rank = (w >> 24) & 3, pieces = (w >> 26) & 7, wave0 = wave - rank;
            const int row = mblk * 16 + r;
            for (int i = rank; i < 4; i += pieces) {
                float pm[kWaves], pl[kWaves];
                f32x4 po[kWaves];
#pragma unroll
                for (int q = 0; q < kWaves; ++q) {
                    pm[q] = -INFINITY; pl[q] = 0.f; po[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (q < pieces) {                                            // wave-uniform
                        const float *slot = lds + ((wave0 + q) * 2 + (q == 0 ? 1 : 0)) * kSlotFloats;
                        const f32x2 ml = *reinterpret_cast<const f32x2 *>(slot + kImageFloats + 2 * r);
                        pm[q] = ml[0];
                        pl[q] = ml[1];
                        po[q] = *reinterpret_cast<const f32x4 *>(slot + (i * 64 + lane) * 4);
                    }
                }
                float m_tot = pm[0];
#pragma unroll
                for (int q = 1; q < kWaves; ++q) m_tot = fmaxf(m_tot, pm[q]);
                float l_tot = 0.f;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < kWaves; ++q) {
                    const float f = __builtin_amdgcn_exp2f(pm[q] - m_tot);       // 0 for absent pieces (m = -inf)
                    l_tot += pl[q] * f;
                    sum += po[q] * f;
                }
                if (row < N) *reinterpret_cast<f32x4 *>(og + (uint64_t)(uint32_t)row * p.o_rs + 16 * g + 4 * i) = sum * (1.0f / l_tot);
            }
        }
    }
#ifdef CMHSA_STAMPS
    STAMP(3);
    if (g_block_times && threadIdx.x == 0) g_block_times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace

// ---- host: the plan ---------------------------------------------------------------------------------------------------
// Blocks (block t-1 has t tiles) go, longest first, to the workgroup of the (b,h) that carries the fewest tiles so far
// (longest-processing-time-first: the loads end up within a tile or two of each other); a workgroup's blocks are laid
// end to end in that order and cut into four runs of q = ceil(load / 4) tiles, one per wave.
// Per-wave cost model (cycles), fitted to measured launch times (tools/run_t16.sh): a tile = 32 MFMAs of 32 cycles plus the
// softmax arithmetic that cannot overlap them; a job = query fragments, a first step down the max path, a store or an LDS image.
constexpr double kFixedCycles = 4500.0, kTileCycles = 1150.0, kJobCycles = 350.0;

// Blocks (block t-1 has t tiles) -> the G workgroups of a (b,h).  method 0: longest first into the workgroup with the fewest
// tiles so far (few blocks per workgroup, but the fullest ends up to a block above the mean when blocks are large against a
// workgroup's share: T = 1024 on 21 workgroups, 109 tiles against a mean of 99); method 1: best fit decreasing into
// workgroups of capacity C = the mean rounded up to the waves' quantum, C + 4, ... (reaches the mean; the last workgroup
// collects the small blocks).  A workgroup's blocks are then laid end to end, longest first, and cut into four runs of
// q = ceil(load / 4) tiles, one per wave.
static bool build(int n16, int G, int J, int method, Plan16 &out) {
    std::vector<std::vector<int>> bins;
    std::vector<int> load;
    if (method == 0) {
        bins.assign((size_t)G, {});
        load.assign((size_t)G, 0);
        for (int t = n16; t >= 1; --t) {
            int pick = 0;
            for (int j = 1; j < G; ++j) if (load[j] < load[pick]) pick = j;
            bins[pick].push_back(t);
            load[pick] += t;
        }
    } else {
        const int total = n16 * (n16 + 1) / 2;
        int C = std::max(kWaves * ((total + kWaves * G - 1) / (kWaves * G)), kWaves * ((n16 + kWaves - 1) / kWaves));
        for (;; C += kWaves) {
            bins.clear();
            load.clear();
            bool fits = true;
            for (int t = n16; t >= 1 && fits; --t) {
                int pick = -1;
                for (int j = 0; j < (int)bins.size(); ++j)
                    if (load[j] + t <= C && (pick < 0 || load[j] > load[pick])) pick = j;
                if (pick < 0) {
                    if ((int)bins.size() == G) { fits = false; break; }
                    bins.emplace_back();
                    load.push_back(0);
                    pick = (int)bins.size() - 1;
                }
                bins[pick].push_back(t);
                load[pick] += t;
            }
            if (fits) break;
        }
        G = (int)bins.size();
    }
    // heaviest workgroups first (dispatched first)
    std::vector<int> order((size_t)G);
    for (int j = 0; j < G; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return load[a] > load[b2]; });
    out.n16 = n16;
    out.wgs_per_bh = G;
    out.jobs_per_wave = J;
    out.tiles_per_wave = 0;
    out.max_jobs = 0;
    out.cycles = 0.0;
    std::fill(out.table, out.table + kTableWords, 0u);
    for (int jo = 0; jo < G; ++jo) {
        const int j = order[jo], S = load[j], q = (S + kWaves - 1) / kWaves;
        out.tiles_per_wave = std::max(out.tiles_per_wave, q);
        int njobs[kWaves] = {0, 0, 0, 0};
        bool merges = false;
        int pos = 0;
        for (int t : bins[j]) {
            const int w_first = pos / q, w_last = (pos + t - 1) / q, pieces = w_last - w_first + 1;
            if (pieces > 4 || w_last >= kWaves) return false;
            merges |= pieces > 1;
            for (int w = w_first; w <= w_last; ++w) {
                const int lo = std::max(pos, w * q), hi = std::min(pos + t, (w + 1) * q);
                if (njobs[w] >= J) return false;
                out.table[((size_t)jo * kWaves + w) * J + njobs[w]++] = job_word(t - 1, lo - pos, hi - lo, w - w_first, pieces);
            }
            pos += t;
        }
        for (int w = 0; w < kWaves; ++w) {
            out.max_jobs = std::max(out.max_jobs, njobs[w]);
            const int tiles_w = std::max(0, std::min(S, (w + 1) * q) - w * q);
            out.cycles = std::max(out.cycles, kFixedCycles + kTileCycles * tiles_w + kJobCycles * njobs[w]);
            if (merges) out.table[((size_t)jo * kWaves + w) * J] |= kWgMergesBit;
        }
    }
    return true;
}

bool make_plan(int64_t bh, int n16, int cus, Plan16 &out) {
    out.ok = false;
    if (n16 < 1 || n16 > kMaxBlocks || bh < 1 || bh > cus) return false;
    const int g_cap = (int)std::min<int64_t>(cus / bh, n16);      // one workgroup per CU, one wave per SIMD; a block is never shared
    Plan16 best;
    for (int J = 4; J <= 8; J += 4) {
        const int G = std::min(g_cap, kTableWords / (kWaves * J));
        if (G < 1) continue;
        for (int method = 0; method < 2; ++method) {
            Plan16 cand;
            if (!build(n16, G, J, method, cand)) continue;
            if (!best.ok || cand.cycles < best.cycles) { best = cand; best.ok = true; }
        }
    }
    if (!best.ok) return false;
    out = best;
    return true;
}

int launch(const Plan16 &plan, const float *q, const float *k, const float *v, float *o,
           int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
           int64_t bh, int seq_len, int num_heads, void *stream) {
    Params16 p;
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs;
    p.q_rs = (unsigned)q_rs; p.k_rs = (unsigned)k_rs; p.v_rs = (unsigned)v_rs; p.o_rs = (unsigned)o_rs;
    p.seq_len = seq_len;
    p.num_heads = num_heads;
    p.wgs_per_bh = plan.wgs_per_bh;
    p.total_wgs = (int)(bh * plan.wgs_per_bh);
    p.per_xcd = (p.total_wgs + 7) / 8;
    p.heads_magic = 0xffffffffu / (unsigned)num_heads;
    p.wgs_magic = 0xffffffffu / (unsigned)plan.wgs_per_bh;
    memcpy(p.table, plan.table, sizeof(p.table));
    void *args[] = {&p};
    const void *fn = plan.jobs_per_wave == 8 ? reinterpret_cast<const void *>(fwd16_kernel<8>) : reinterpret_cast<const void *>(fwd16_kernel<4>);
    return (int)hipLaunchKernel(fn, dim3((unsigned)p.per_xcd * 8u), dim3(kWaves * 64), args, 0, static_cast<hipStream_t>(stream));
}

#ifdef CMHSA_STAMPS
int debug_set_stamps(unsigned long long *buf, int block) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_block), &block, sizeof(block));
    return (int)e;
}
int debug_set_block_times(unsigned long long *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_block_times), &buf, sizeof(buf)); }
#endif

}  // namespace cmhsa_t16
