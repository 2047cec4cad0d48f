// cmhsa_fwd16.hip -- the "16-key tiles" family of the fp32 causal attention forward for MI355X (gfx950 / CDNA4).
//
// Same function as cmhsa_fwd.hip (reference torch_extensions/causal_multihead_self_attention/
// causal_multihead_self_attention.cu:64-405, "CMHSA.cu"; the naive formula of reference causal_self_attention.py:173-194):
//     O[b, :, 64h:64h+64] = softmax(mask(Q_h K_h^T) / 8) V_h ,  key j <= query i,
// decomposed for grids that fit the chip in ONE round (BASELINE config 2, and the B = 1 launches of a generation, config 3),
// where the kernel of cmhsa_fwd.hip pays for its 32 x 32 units: a query block's keys can only be spread over the four
// SIMDs of one CU, the work comes in quanta of 4 096 MFMA cycles -- 7 units on the busiest SIMD of config 2 where an even
// spread is 6.19 --, and the prologue (K and Q through LDS, a barrier) and merge are 7-9 k cycles of a 12-40 k cycle launch
// (profiles/r03_experiments/baseline_r02_kernel_stamps_*.txt).  Here:
//   * key tiles of 16 rows on v_mfma_f32_16x16x4_f32 (same flop rate as the 32x32x2 shape).  A query block is 16 rows
//     (kSub = 1: a tile is 16 QK + 16 PV MFMAs = 1 024 cycles, a quarter of a unit; for the small grids of a generation,
//     where a block's tiles then need half the SIMD time) or two 16-row halves that share every K and V register
//     (kSub = 2: the arithmetic intensity of the 32-row kernel -- with one half per load the family asks the L2 for
//     32 B/clk/CU, more than it delivers -- at half its quantum);
//   * S^T = K Q^T again: the query sits on the lane (l % 16), the S^T accumulator registers are, as they stand, the B
//     operand of O^T += V^T P^T; row max / sum cross the four lane groups with v_permlane16_swap + v_permlane32_swap
//     (the wave64 counterpart of the reference's 16-lane __shfl_down_sync tree, CMHSA.cu:267-284);
//   * no LDS and no barrier in front of the first MFMA: K and Q fragments go straight from global memory into the
//     A / B operand layout (lane (g, r) reads the 16-B piece 4c + g of row r: every row's four lanes cover 64
//     contiguous bytes), V likewise (lane (g, r) reads floats 4r..4r+3 of key 4g + i: whole 256-B rows);
//   * a workgroup = four waves (one per SIMD) or eight (two per SIMD: one covers the other's memory waits), one
//     workgroup per CU.  Each wave runs a LIST of jobs from a host-made table (cmhsa_tiles16.h): a job = a run of key
//     tiles of one block.  The blocks of a workgroup are laid end to end and cut into four equal runs, one per SIMD (halved
//     between its two waves), so all SIMDs carry the same number of tiles; a block cut by a run boundary is merged
//     through LDS behind the kernel's only barrier, a whole block is normalised and stored straight from the accumulators;
//   * K / V of the next step (two tiles) are requested as soon as the registers of the current one fall free (K behind
//     each tile's QK chain, V behind each PV k-step), across job boundaries; the next job's query fragments are requested
//     at the start of the current job;
//   * online softmax in the log2 domain, deferred normalisation, speculative lazy rescale -- as in cmhsa_fwd.hip.
// Causal early exit (CMHSA.cu:133-136) = the table never names a tile past a block's last diagonal tile.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>
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
// LDS: [list u][slot A, B][per half: image [4 rows i][64 lanes] float4] [per half: (m, l) of the 16 queries]
constexpr int kImageFloats = 4 * 64 * 4;
constexpr int slot_floats(int sub) { return sub * (kImageFloats + 32); }
constexpr int lds_bytes(int sub, int waves) { return waves * 2 * slot_floats(sub) * 4; }        // 33 KiB .. 132 KiB

struct Params16 {
    const float *q, *k, *v;
    float *o;
    int64_t q_bs, k_bs, v_bs, o_bs;     // batch strides (floats)
    unsigned q_rs, k_rs, v_rs, o_rs;    // row strides (floats; <= 2^24, host check)
    int seq_len, num_heads;
    int wgs_per_bh, total_wgs, per_xcd; // workgroups per (b,h); bh * wgs_per_bh; ceil(total_wgs / 8)
    int gshift;                         // a job word counts tiles in granules of 1 << gshift (Plan16::granule)
    unsigned heads_magic, wgs_magic;    // floor((2^32-1) / num_heads), / wgs_per_bh
    unsigned table[kTableWords];        // [workgroup of a (b,h)][list u][32 / waves] job words (cmhsa_tiles16.h)
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

// kSub: 16-row halves per query block.  kWaves: 4 or 8 waves per workgroup (32 / kWaves jobs per wave).
// kContig: the workgroup -> (b,h) map (cmhsa_tiles16.h, Plan16::contig).
template <int kSub, int kWaves, bool kContig>
__global__ __launch_bounds__(kWaves * 64, 2) void fwd16_kernel(Params16 p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // kWaves * 2 * slot_floats(kSub)
    constexpr int kJobs = kWordsPerWg / kWaves;
    constexpr int kSlot = slot_floats(kSub);
    constexpr int kBlockRows = 16 * kSub;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = kWaves == 8 ? 2 * (wave & 3) + (wave >> 2) : wave;       // this wave's job list, in run order (cmhsa_tiles16.h)
    const int g = lane >> 4, r = lane & 15;      // lane group (k index of both MFMA operands), row / query within the tile
#ifdef CMHSA_STAMPS
    const int lin_block = blockIdx.x + gridDim.x * blockIdx.y;
    const bool stamp_on = g_stamp_buf != nullptr && lin_block == g_stamp_block && lane == 0;
    unsigned long long *stamp_ptr = g_stamp_buf + wave * 128;
    int stamp_i = 4;
    STAMP(0);
    if (g_block_times && threadIdx.x == 0) g_block_times[2 * lin_block] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- workgroup -> (b,h), workgroup j of it.
    //      kContig: workgroups are dealt round-robin over the 8 XCDs in linear order, so XCD x gets the ids x, x + 8, ...:
    //      it is given a CONTIGUOUS run of the (b,h)-major workgroup list, so that a (b,h)'s K and V are read through one
    //      XCD's L2 (two at a boundary), whatever B*H is.  Else grid = (workgroup, (b,h)): j comes with the wave, and this
    //      wave's job words ride in the first batch of scalar loads (short launches: one round trip less). ----
    int bh, j;
    if constexpr (kContig) {
        const int id = blockIdx.x;
        const int pos = (id & 7) * p.per_xcd + (id >> 3);
        if (pos >= p.total_wgs) return;                   // padding of the last XCDs' runs (whole workgroup, before any barrier)
        bh = udiv_magic(pos, p.wgs_per_bh, p.wgs_magic);
        j = pos - bh * p.wgs_per_bh;
    } else {
        j = blockIdx.x;
        bh = blockIdx.y;
    }
    unsigned jw[kJobs];
#pragma unroll
    for (int i = 0; i < kJobs; ++i) jw[i] = p.table[j * kWordsPerWg + u * kJobs + i];
    if constexpr (kJobs == 8)
        asm volatile("" ::"s"(jw[4]), "s"(jw[5]), "s"(jw[6]), "s"(jw[7]));
    asm volatile("" ::"s"(jw[0]), "s"(jw[1]), "s"(jw[2]), "s"(jw[3]), "s"(p.q), "s"(p.k), "s"(p.v), "s"(p.o), "s"(p.q_rs), "s"(p.k_rs),
                 "s"(p.v_rs), "s"(p.o_rs));
    asm volatile("" ::"s"(p.q_bs), "s"(p.k_bs), "s"(p.v_bs), "s"(p.o_bs), "s"(p.seq_len), "s"(p.num_heads), "s"(p.heads_magic), "s"(p.gshift));
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
    int njobs = 0;
    unsigned word_a = 0, word_b = 0;                       // this wave's pieces of cut blocks: a later piece (slot A), a first piece (slot B)
#pragma unroll
    for (int i = 0; i < kJobs; ++i) {
        const int cnt = (jw[i] >> 16) & 255;
        njobs += cnt ? 1 : 0;
        if (cnt && ((jw[i] >> 27) & 15) > 1) {
            if ((jw[i] >> 24) & 7) word_a = jw[i]; else word_b = jw[i];
        }
    }

    // per-lane byte offsets inside a step's 32 rows, every one in a register of its own (an add of a constant that the
    // compiler hoists out of the loop is no longer folded into the instruction's offset field)
    int koffs[2][4], voffs[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        koffs[0][c] = (r * (int)p.k_rs + 4 * g + 16 * c) * 4;                       // K: row r (+16), floats 16c + 4g .. +3
        koffs[1][c] = koffs[0][c] + 16 * (int)p.k_rs * 4;
        voffs[0][c] = ((4 * g + c) * (int)p.v_rs + 4 * r) * 4;                      // V: key 4g + i (+16), floats 4r .. 4r+3
        voffs[1][c] = voffs[0][c] + 16 * (int)p.v_rs * 4;
    }
    const int qoff = (r * (int)p.q_rs + 4 * g) * 4;

    f32x4 kb[2][4];                  // K of the step: [tile][c] = floats 16c + 4g .. +3 of key r
    f32x4 vb[2][4];                  // V of the step: [tile][i] = floats 4r .. 4r+3 of key 4g + i
    f32x4 qn[kSub][4];               // the NEXT job's query fragments, raw: [half][c]
    auto load_q = [&](unsigned w) {
        const bool en = ((w >> 16) & 255) != 0;
        const __amdgpu_buffer_rsrc_t qs = rows_rsrc(qg, p.q_rs, (int)(w & 255) * kBlockRows, kBlockRows, N, en);
#pragma unroll
        for (int s = 0; s < kSub; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) qn[s][c] = load16(qs, qoff + (s * 16 * (int)p.q_rs + 16 * c) * 4);
    };
    // one memory round trip ahead of the first MFMA, in the order of use: the first job's query fragments, its first
    // step's K (tile A, then B), then V
    {
        load_q(jw[0]);
        const int row0 = (int)(((jw[0] >> 8) & 255) << p.gshift) * 16;
        const bool en = njobs > 0;
        const __amdgpu_buffer_rsrc_t ks = rows_rsrc(kg, p.k_rs, row0, 32, N, en), vs = rows_rsrc(vg, p.v_rs, row0, 32, N, en);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) kb[t][c] = load16(ks, koffs[t][c]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) vb[t][i] = load16(vs, voffs[t][i]);
    }
    STAMP(121);

    float *my_slots = lds + u * 2 * kSlot;
    // the second-dispatched half of an 8-wave workgroup loses every arbitration against its SIMD partner (both at priority 0
    // the older wave's chains run at 8.4 k cycles per step, the younger's at 13 k, and the younger finishes alone): one static
    // priority step for waves 4-7 (MI355X guide, "Static priority for the younger half")
    if constexpr (kWaves == 8) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    STAMP(1);
    for (int ji = 0; ji < njobs; ++ji) {
        const unsigned cw = job(ji), nw = job(ji + 1);       // (lane kJobs holds "no job")
        const int blk = cw & 255, lo = (int)(((cw >> 8) & 255) << p.gshift), hi = lo + (int)(((cw >> 16) & 255) << p.gshift);
        const int nlo = (int)(((nw >> 8) & 255) << p.gshift);
        const bool have_next = ((nw >> 16) & 255) != 0;
        float qf[kSub][16];          // Q^T fragments of the job, pre-scaled: k-step 4c + jj uses d = 16c + 4g + jj
#pragma unroll
        for (int s = 0; s < kSub; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) qf[s][4 * c + jj] = qn[s][c][jj] * kScaleLog2;
        load_q(nw);                  // in flight for the whole of this job
        f32x4 acc[kSub][4];          // O^T: acc[s][dt][i] <-> d = 16g + 4i + dt, query r of half s
        float m_run[kSub], l_run[kSub];
#pragma unroll
        for (int s = 0; s < kSub; ++s) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run[s] = -INFINITY;
            l_run[s] = 0.f;
        }

        for (int t = lo; t < hi; t += 2) {
            const bool first = t == lo;
            // the step after this one: the job's next two tiles, else the next job's first two, else none
            const bool more = t + 2 < hi;
            const int row0n = (more ? t + 2 : nlo) * 16;
            const bool en = more || have_next;
            const __amdgpu_buffer_rsrc_t ksn = rows_rsrc(kg, p.k_rs, row0n, 32, N, en), vsn = rows_rsrc(vg, p.v_rs, row0n, 32, N, en);
            // the step's second tile is absent at the end of a run of odd length: its MFMAs are skipped (wave-uniform
            // branches that define nothing but the chain's own result; every load stays outside them, so that the
            // loop-carried K / V registers have ONE definition each and the compiler keeps them where they are)
            const bool has_b = t + 1 < hi;
            constexpr int kPar = kSub == 1 ? 2 : 1;          // one half per block: its even and odd k-steps as two chains
#ifdef CMHSA_STAMPS
            STAMP(stamp_i);
#endif
            // ---- S^T[key][query] = sum_d K[key][d] Q[query][d], tile A then tile B; at least two independent chains in
            //      turn (the shape's dependent latency is 40 cycles against 32 of issue).  K of the NEXT step goes into the
            //      registers as they fall free, one load behind each group of MFMAs (issued under the matrix pipe's
            //      shadow, not in a burst in front of the softmax): tile A's between the groups of tile B's chain, tile
            //      B's between the first PV k-steps. ----
            f32x4 sc[kSub][2];
            {
                f32x4 ch[2][kSub][kPar];
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                    for (int s = 0; s < kSub; ++s)
#pragma unroll
                        for (int q2 = 0; q2 < kPar; ++q2) ch[tl][s][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int s = 0; s < kSub; ++s)
                            ch[0][s][jj % kPar] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[0][c][jj], qf[s][4 * c + jj], ch[0][s][jj % kPar], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (has_b) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int s = 0; s < kSub; ++s)
                                ch[1][s][jj % kPar] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[1][c][jj], qf[s][4 * c + jj], ch[1][s][jj % kPar], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    kb[0][c] = load16(ksn, koffs[0][c]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                    for (int s = 0; s < kSub; ++s) sc[s][tl] = kPar == 2 ? ch[tl][s][0] + ch[tl][s][kPar - 1] : ch[tl][s][0];
            }
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 1);
#endif
#pragma unroll
            for (int s = 0; s < kSub; ++s) {
                // ---- causal mask (CMHSA.cu:228, :249): tile kSub * blk + s is the diagonal tile of half s (key > query
                //      pairs; keys past the sequence end exist in a diagonal tile only and are masked with it), the
                //      block's later tile is all masked for half 0; so is an absent tile B ----
                const int dtile = kSub * blk + s;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    if (t + tl >= dtile || (tl == 1 && !has_b)) {                    // wave-uniform
                        const bool all = t + tl > dtile || (tl == 1 && !has_b);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (all || 4 * g + i > r) sc[s][tl][i] = -INFINITY;
                    }
                }
            }
            // ---- online softmax, log2 domain, speculative against the old running max (see cmhsa_fwd.hip): the max is
            //      only computed on a job's first step and when some row sum leaves [0, 2^64) ----
            f32x4 pr[kSub][2];
#pragma unroll
            for (int s = 0; s < kSub; ++s) {
                float psum = 0.f;
                bool redo = first;
                if (!first) {
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            pr[s][tl][i] = __builtin_amdgcn_exp2f(sc[s][tl][i] - m_run[s]);
                            psum += pr[s][tl][i];
                        }
                    redo = !__all(psum < kRescaleSum);      // NaN and inf compare false: they take the max path too
                }
                if (redo) {
                    float mx = fmaxf(fmaxf(fmaxf(sc[s][0][0], sc[s][0][1]), fmaxf(sc[s][0][2], sc[s][0][3])),
                                     fmaxf(fmaxf(sc[s][1][0], sc[s][1][1]), fmaxf(sc[s][1][2], sc[s][1][3])));
                    mx = x32_max(x16_max(mx));
                    const float m_new = fmaxf(m_run[s], mx);
                    // a piece that holds only masked keys for this half (the block's last tile, half 0) stays EMPTY:
                    // (m, l, O) = (-inf, 0, 0) -- the partial that CMHSA.cu:45-48 handles with its sum == 0 branches
                    const float m_ref = m_new == -INFINITY ? 0.f : m_new;
                    const float alpha = __builtin_amdgcn_exp2f(m_run[s] - m_ref);       // 0 on a job's first step (m_run = -inf)
                    l_run[s] *= alpha;
                    m_run[s] = m_new;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) acc[s][dt] *= alpha;
                    psum = 0.f;
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            pr[s][tl][i] = __builtin_amdgcn_exp2f(sc[s][tl][i] - m_ref);
                            psum += pr[s][tl][i];
                        }
                }
                l_run[s] += psum;
            }
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 2);
#endif
            // ---- O^T[d][query] += sum_key V[key][d] P[query][key]: k-step (tile, i) pairs the keys 4g + i of the four
            //      lane groups, which is where register i of the S^T accumulator holds them: P feeds the B operand as it
            //      stands.  Behind each k-step's MFMAs: V of the next step into the registers that just fell free. ----
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (tl == 0 || has_b) {
#pragma unroll
                        for (int s = 0; s < kSub; ++s)
#pragma unroll
                            for (int dt = 0; dt < 4; ++dt)
                                acc[s][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[tl][i][dt], pr[s][tl][i], acc[s][dt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    vb[tl][i] = load16(vsn, voffs[tl][i]);
                    if (tl == 0) kb[1][i] = load16(ksn, koffs[1][i]);
                }
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 3);
            stamp_i += 4;
#endif
        }

        // ---- the job is done.  A whole block: normalise and store, lane (g, r) = query r, floats 16g + 4i .. + 3.
        //      A piece of a cut block: the un-normalised image and (m, l) go to one of the list's two LDS slots (a piece
        //      that starts its block: slot B, a later piece: slot A) for the merge behind the barrier. ----
        const int pieces = (cw >> 27) & 15;
        float *slot = my_slots + (((cw >> 24) & 7) ? 0 : kSlot);
#pragma unroll
        for (int s = 0; s < kSub; ++s) {
            const float l_tot = x32_sum(x16_sum(l_run[s]));
            if (pieces <= 1) {
                const float inv = 1.0f / l_tot;
                const int row = blk * kBlockRows + 16 * s + r;
                if (row < N) {
                    float *dst = og + (uint64_t)(uint32_t)row * p.o_rs + 16 * g;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<f32x4 *>(dst + 4 * i) = f32x4{acc[s][0][i], acc[s][1][i], acc[s][2][i], acc[s][3][i]} * inv;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4 *>(slot + s * kImageFloats + (i * 64 + lane) * 4) = f32x4{acc[s][0][i], acc[s][1][i], acc[s][2][i], acc[s][3][i]};
                if (g == 0) *reinterpret_cast<f32x2 *>(slot + kSub * kImageFloats + s * 32 + 2 * r) = f32x2{m_run[s], l_tot};
            }
        }
    }
    STAMP(2);

    // ---- merge of the cut blocks: the pieces of a block sit on consecutive lists (piece 0 in slot B of its list, the
    //      others in slot A of theirs); the 4 kSub (half, image row) items go round-robin over the block's waves, each
    //      combines its items' pieces in order (the bits do not depend on timing), divides by the merged l and stores. ----
    if (wg_merges) {
        __syncthreads();
        STAMP(120);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned w = t == 0 ? word_a : word_b;
            if (!w) continue;                                                    // wave-uniform
            const int mblk = w & 255, rank = (w >> 24) & 7, pieces = (w >> 27) & 15, u0 = u - rank;
            for (int it = rank; it < 4 * kSub; it += pieces) {
                const int s = it >> 2, i = it & 3;
                const int row = mblk * kBlockRows + 16 * s + r;
                // every piece's (m, l) and image row requested at once (absent pieces: (-inf, 0, 0))
                float pm[kWaves], pl[kWaves];
                f32x4 po[kWaves];
#pragma unroll
                for (int q = 0; q < kWaves; ++q) {
                    pm[q] = -INFINITY; pl[q] = 0.f; po[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (q < pieces) {                                            // wave-uniform
                        const float *sl = lds + ((u0 + q) * 2 + (q == 0 ? 1 : 0)) * kSlot;
                        const f32x2 ml = *reinterpret_cast<const f32x2 *>(sl + kSub * kImageFloats + s * 32 + 2 * r);
                        pm[q] = ml[0];
                        pl[q] = ml[1];
                        po[q] = *reinterpret_cast<const f32x4 *>(sl + s * kImageFloats + (i * 64 + lane) * 4);
                    }
                }
                float m_tot = pm[0];
#pragma unroll
                for (int q = 1; q < kWaves; ++q) m_tot = fmaxf(m_tot, pm[q]);
                float l_tot = 0.f;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < kWaves; ++q) {
                    const float f = __builtin_amdgcn_exp2f(pm[q] - m_tot);       // 0 for an absent or empty piece (m = -inf)
                    l_tot += pl[q] * f;
                    sum += po[q] * f;
                }
                if (row < N) *reinterpret_cast<f32x4 *>(og + (uint64_t)(uint32_t)row * p.o_rs + 16 * g + 4 * i) = sum * (1.0f / l_tot);
            }
        }
    }
#ifdef CMHSA_STAMPS
    STAMP(3);
    if (g_block_times && threadIdx.x == 0) g_block_times[2 * lin_block + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace

// ---- host: the plan ---------------------------------------------------------------------------------------------------
// Cost model (cycles at 2.4 GHz, on the scale of cmhsa_plan.h's single-round estimates: launch time minus ~1.7 k), fitted to
// measured launch times (tools/run_t16_sweep.sh, profiles/r03_experiments/t16_*.log).  One wave per SIMD, 16-row blocks:
// 9.5 k fixed + 1.3 k per tile on the busiest SIMD; a tile costs more once the launch's K and V (kv_mb, MiB) pass ~3 MiB --
// every CU then streams at once and the memory side, not the matrix pipe, sets the pace (one 16-row half per load asks for
// 32 B/clk/CU): + 250 cycles per tile and MiB.  Two halves per load (sub = 2) and two waves per SIMD are priced from their
// own measurements.  A job = query fragments, a first step down the max path, a store or an LDS image.
constexpr double kFixedCycles = 9500.0, kJobCycles = 300.0;
static double tiles_cost(int sub, int waves, int tiles, double kv_mb) {
    // 32-row blocks (a tile = 2 048 MFMA cycles): fits to single long prompts, B = 1..2, T = 1536..4000 (tools/run_t16_long.sh:
    // 2.45 k cycles per tile with two waves per SIMD, 2.6 k with one, and a tail of 10-15 k in which a SIMD's younger wave, which
    // loses the arbitration while both run, finishes alone); short grids come out above their measured times and are not picked
    if (sub == 2) return waves == 8 ? 2450.0 * tiles + 10500.0 : 2600.0 * tiles + 15000.0;
    return (1300.0 + 250.0 * std::max(0.0, kv_mb - 3.0)) * tiles + (waves == 8 ? 2000.0 : 0.0);
}

// Blocks (block b has sub (b + 1) tiles) -> the G workgroups of a (b,h).  method 0: longest first into the workgroup with the
// fewest tiles so far (few blocks per workgroup, but the fullest ends up to a block above the mean when blocks are large
// against a workgroup's share: T = 1024 on 21 workgroups, 109 tiles against a mean of 99); method 1: best fit decreasing into
// workgroups of capacity C = the mean rounded up to the SIMDs' quantum, C + 4, ... (reaches the mean; the last workgroup
// collects the small blocks).  A workgroup's blocks are then laid end to end, longest first, and cut into four runs of
// q = ceil(load / 4) tiles, one per SIMD; with eight waves each run is halved between the SIMD's two waves.
static bool build(int nblocks, int sub, int G, int waves, int method, double kv_mb, Plan16 &out) {
    const int J = kWordsPerWg / waves;
    // share of a SIMD's run that its first-dispatched wave (even list) takes in an 8-wave workgroup.  With the static priority
    // step of the kernel the YOUNGER wave wins the arbitration while both run; 40 : 60 lets them finish together on long runs
    // (32-row blocks, B1 H12 T4096: 206.5 us at 50 : 50, 202.8 at 40 : 60, 208.3 at 24 : 76; tools/ab experiments, round 3)
    const int older_pct = sub == 2 ? 40 : 50;
    // a job word counts its tiles in 8 bits: blocks of more than 255 tiles (32-row blocks beyond T = 4080) are cut at even
    // tile positions only -- every 32-row block has an even number of tiles, a step is two tiles -- and the words count pairs
    const int gran = sub * nblocks > kMaxTiles ? 2 : 1;
    if (gran == 2 && (sub != 2 || sub * nblocks > 2 * kMaxTiles)) return false;
    std::vector<std::vector<int>> bins;
    std::vector<int> load;
    const bool mix = method >= 2;            // a workgroup's blocks laid out large, small, large, small, ... (below)
    method &= 1;
    if (method == 0) {
        bins.assign((size_t)G, {});
        load.assign((size_t)G, 0);
        for (int b = nblocks; b >= 1; --b) {
            int pick = 0;
            for (int j = 1; j < G; ++j) if (load[j] < load[pick]) pick = j;
            bins[pick].push_back(b);
            load[pick] += sub * b;
        }
    } else {
        const int total = sub * nblocks * (nblocks + 1) / 2;
        int C = std::max(4 * ((total + 4 * G - 1) / (4 * G)), 4 * ((sub * nblocks + 3) / 4));
        for (;; C += 4) {
            bins.clear();
            load.clear();
            bool fits = true;
            for (int b = nblocks; b >= 1 && fits; --b) {
                const int t = sub * b;
                int pick = -1;
                for (int j = 0; j < (int)bins.size(); ++j)
                    if (load[j] + t <= C && (pick < 0 || load[j] > load[pick])) pick = j;
                if (pick < 0) {
                    if ((int)bins.size() == G) { fits = false; break; }
                    bins.emplace_back();
                    load.push_back(0);
                    pick = (int)bins.size() - 1;
                }
                bins[pick].push_back(b);
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
    out.sub = sub;
    out.granule = gran;
    out.blocks = nblocks;
    out.wgs_per_bh = G;
    out.waves = waves;
    out.tiles_per_simd = 0;
    out.max_jobs = 0;
    out.cycles = 0.0;
    std::fill(out.table, out.table + kTableWords, 0u);
    for (int jo = 0; jo < G; ++jo) {
        const int j = order[jo], S = load[j], q = ((S + 3) / 4 + gran - 1) / gran * gran;
        out.tiles_per_simd = std::max(out.tiles_per_simd, q);
        // list boundaries in run order: bound[u] .. bound[u + 1]
        int bound[kMaxWaves + 1];
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int a = std::min(S, sidx * q), e = std::min(S, (sidx + 1) * q);
            if (waves == 8) {
                bound[2 * sidx] = a;
                // the older wave of the SIMD (even list) wins most arbitrations while both run: it takes the larger part
                bound[2 * sidx + 1] = std::min(e, a + (((e - a) * older_pct + 99) / 100 + gran - 1) / gran * gran);
            } else {
                bound[sidx] = a;
            }
        }
        bound[waves] = S;
        int njobs[kMaxWaves] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool merges = false;
        int pos = 0;
        // order of the blocks along the workgroup's line of tiles: longest first -- or, when that leaves some list with more
        // jobs than its words (long sequences: the short blocks all end up on the last lists), alternately from both ends
        std::vector<int> line = bins[j];
        if (mix) {
            const size_t nb = line.size();
            for (size_t i = 0; i < nb; ++i) line[i] = (i & 1) ? bins[j][nb - 1 - i / 2] : bins[j][i / 2];
        }
        for (int b : line) {
            const int t = sub * b;
            // the lists [bound[u], bound[u+1]) that the block's tiles [pos, pos + t) meet
            int u_first = -1, u_last = -1;
            for (int uu = 0; uu < waves; ++uu)
                if (std::max(pos, bound[uu]) < std::min(pos + t, bound[uu + 1])) { if (u_first < 0) u_first = uu; u_last = uu; }
            if (u_first < 0) return false;
            const int pieces = u_last - u_first + 1;
            if (pieces > 8) return false;
            merges |= pieces > 1;
            for (int uu = u_first; uu <= u_last; ++uu) {
                const int lo = std::max(pos, bound[uu]), hi = std::min(pos + t, bound[uu + 1]);
                if (hi <= lo) return false;                                        // (an empty list inside a block's span: cannot happen)
                if (njobs[uu] >= J) return false;
                if ((lo - pos) % gran || (hi - lo) % gran) return false;            // (cannot happen: every boundary is a multiple)
                out.table[(size_t)jo * kWordsPerWg + uu * J + njobs[uu]++] = job_word(b - 1, (lo - pos) / gran, (hi - lo) / gran, uu - u_first, pieces);
            }
            pos += t;
        }
        for (int sidx = 0; sidx < 4; ++sidx) {
            int tiles_s = 0, jobs_s = 0;
            for (int k2 = 0; k2 < waves / 4; ++k2) {
                const int uu = waves == 8 ? 2 * sidx + k2 : sidx;
                tiles_s += bound[uu + 1] - bound[uu];
                jobs_s += njobs[uu];
                out.max_jobs = std::max(out.max_jobs, njobs[uu]);
            }
            out.cycles = std::max(out.cycles, kFixedCycles + tiles_cost(sub, waves, tiles_s, kv_mb) + kJobCycles * sub * std::max(0, jobs_s - 1));
        }
        if (merges)
            for (int uu = 0; uu < waves; ++uu) out.table[(size_t)jo * kWordsPerWg + uu * J] |= kWgMergesBit;
    }
    return true;
}

bool make_plan(int64_t bh, int seq_len, int cus, Plan16 &out, int force_sub, int force_waves, int force_contig) {
    out.ok = false;
    if (seq_len < 1 || bh < 1 || bh > cus) return false;
    Plan16 best;
    for (int sub = 1; sub <= 2; ++sub) {
        if (force_sub && force_sub != sub) continue;
        const int nblocks = (seq_len + 16 * sub - 1) / (16 * sub);
        if (sub * nblocks > (sub == 2 ? 2 : 1) * kMaxTiles || nblocks > 255) continue;
        // one workgroup per CU; a block is never shared between workgroups
        const int G = (int)std::min<int64_t>(std::min<int64_t>(cus / bh, nblocks), kMaxWgs);
        if (G < 1) continue;
        for (int waves = 4; waves <= 8; waves += 4) {
            if (force_waves && force_waves != waves) continue;
            bool built = false;
            for (int method = 0; method < 4; ++method) {
                if (method == 2 && built) break;                      // the mixed order only where the plain one does not fit
                Plan16 cand;
                if (!build(nblocks, sub, G, waves, method, (double)bh * seq_len * 512.0 / 1048576.0, cand)) continue;
                built = true;
                if (!best.ok || cand.cycles < best.cycles) { best = cand; best.ok = true; }
            }
        }
    }
    if (!best.ok) return false;
    // short launches take the map that spares the prologue a dependent scalar load; long ones the XCD-contiguous map
    best.contig = force_contig >= 0 ? force_contig != 0 : best.tiles_per_simd * best.sub >= 12;
    out = best;
    return true;
}

template <int kSub, int kWaves, bool kContig>
static int launch_variant(const Params16 &p, int64_t bh, void *stream) {
    static std::atomic<int> attr_set[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const void *fn = reinterpret_cast<const void *>(fwd16_kernel<kSub, kWaves, kContig>);
    constexpr int kLds = lds_bytes(kSub, kWaves);
    if (kLds > 65536 && (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire))) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 64) attr_set[dev].store(1, std::memory_order_release);
    }
    void *args[] = {const_cast<Params16 *>(&p)};
    const dim3 grid = kContig ? dim3((unsigned)p.per_xcd * 8u) : dim3((unsigned)p.wgs_per_bh, (unsigned)bh);
    return (int)hipLaunchKernel(fn, grid, dim3(kWaves * 64), args, kLds, static_cast<hipStream_t>(stream));
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
    p.gshift = plan.granule == 2 ? 1 : 0;
    p.heads_magic = 0xffffffffu / (unsigned)num_heads;
    p.wgs_magic = 0xffffffffu / (unsigned)plan.wgs_per_bh;
    memcpy(p.table, plan.table, sizeof(p.table));
    const int key = (plan.sub == 2 ? 4 : 0) | (plan.waves == 8 ? 2 : 0) | (plan.contig ? 1 : 0);
    switch (key) {
        case 0: return launch_variant<1, 4, false>(p, bh, stream);
        case 1: return launch_variant<1, 4, true>(p, bh, stream);
        case 2: return launch_variant<1, 8, false>(p, bh, stream);
        case 3: return launch_variant<1, 8, true>(p, bh, stream);
        case 4: return launch_variant<2, 4, false>(p, bh, stream);
        case 5: return launch_variant<2, 4, true>(p, bh, stream);
        case 6: return launch_variant<2, 8, false>(p, bh, stream);
        default: return launch_variant<2, 8, true>(p, bh, stream);
    }
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
