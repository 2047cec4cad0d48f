// cmhsa_fwd.hip -- fp32 causal multi-head self-attention forward for MI355X (gfx950 / CDNA4).
//
// Written from scratch for wave64 + fp32 MFMA; it replaces the reference's CUDA kernel and
// launcher (reference torch_extensions/causal_multihead_self_attention/
// causal_multihead_self_attention.cu:64-462, "CMHSA.cu" below) behind the C ABI of include/cmhsa.h.
//
// Decomposition (see DESIGN.md section 3 for the why and the measurements):
//   * work item = one 32-row query block of one (batch, head); its KV range [0, q_block_end) is cut into
//     32-key units (causal early exit: nothing past the diagonal is touched, only the diagonal unit is masked);
//   * a workgroup has W = 8 (one per CU) or 4 (two per CU) wave64s and holds ONE long query block, a PAIR of
//     a long and a short one (analytic plan, pack_plan()) or up to FOUR blocks or parts of blocks listed in a table
//     (cmhsa_plan.h); a block's waves split its units round-robin (split-KV inside the workgroup), so even B=1,
//     T=1024 fills the chip with one round of equally long workgroups.  Large grids run as ONE-WAVE workgroups:
//     a wave runs whole blocks, one or two, with no merge at all and O straight from its accumulators;
//   * S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x2_f32: with S^T the query sits on the lane, so
//     softmax statistics are per-lane scalars and the S^T accumulator registers are, as they stand, the B
//     operand of the PV product (no LDS round trip, no lane shuffles);
//   * K: coalesced buffer loads -> staging registers (one unit ahead) -> the wave's private, XOR-swizzled
//     LDS unit -> ds_read_b128 fragments; V: straight from global into the MFMA A-operand layout, one
//     unit ahead; the query block is fetched once per workgroup into shared LDS.  The main loop has no
//     barrier and no branch inside the MFMA chains;
//   * online softmax in the log2 domain (Q pre-scaled by log2(e)/sqrt(64)), lazy rescale, normalisation
//     deferred to the merge: the partial (m, l, O) of a block's waves are combined through LDS behind the
//     kernel's second and last barrier.
//
// Function computed per (batch b, head h), exactly CMHSA.cu:64-405 / the naive formula of
// reference causal_self_attention.py:173-194:
//     O[b, :, 64h:64h+64] = softmax(mask(Q_h K_h^T) / 8) V_h ,  key j <= query i.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include <type_traits>

#include "cmhsa.h"
#include "cmhsa_common.h"   // constants, Params, the segment decode shared by kernel and host
#include "cmhsa_plan.h"     // host side: which workgroup computes what (plan families, cost models)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// LDS: [kWaves][8 KiB] each wave's K unit (later its merge image) | [kSegs][8 KiB] pre-scaled query blocks |
//      [kWaves][32 x (m, l)] softmax statistics for the merge | [8] "this workgroup arrived last" flags (split plans)
constexpr int lds_floats_q(int waves) { return waves * kUnitFloats; }
constexpr int lds_floats_ml(int waves, int segs) { return (waves + segs) * kUnitFloats; }
constexpr int lds_floats_flags(int waves, int segs) { return lds_floats_ml(waves, segs) + waves * 64; }
// (one-wave workgroups keep nothing but their K unit in LDS: no shared query blocks, no merge)
constexpr int lds_bytes(int waves, int segs) { return waves == 1 ? kUnitFloats * 4 : (lds_floats_flags(waves, segs) + 8) * 4; }   // <= 130 KiB (8 waves, 8 segments), 98 KiB (8, 4), 65 KiB (4), 8 KiB (1)
constexpr float kScaleLog2 = 0.18033688011112042f;        // log2(e) / sqrt(64)
constexpr int kSc1 = 16;                                   // buffer load/store aux bits on gfx950: 16 = sc1 (agent-coherent: stores write through, loads bypass the CU's L1)
constexpr float kRescaleSum = 1.8446744e19f;              // 2^64: a row sum of P above it triggers the rescale path

// Diagnostic build only (-DCMHSA_STAMPS, tools/Makefile target cmhsa_bench_stamps): s_memtime stamps of
// one workgroup's phases go to a buffer of their own; no output value depends on them.  The shipped
// library is built without it and executes no stamp.
#ifdef CMHSA_STAMPS
__device__ unsigned long long *g_stamp_buf = nullptr;
__device__ int g_stamp_block = 0;
__device__ unsigned long long *g_block_times = nullptr;   // [blockIdx][2]: s_memrealtime (100 MHz) at entry / exit
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
// `enable` = false gives num_records = 0: every load through it returns 0 and touches no memory, which
// keeps the software pipeline branch-free past the last unit.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t unit_rsrc(const float *head_base, int64_t rs, int row0, int n_rows,
                                                            bool enable = true) {
    const int valid = min(kKeyUnit, n_rows - row0);       // >= 1 for every unit the kernel touches
    const int bytes = (((valid - 1) * (int)rs + kHeadDim) * 4) & -(int)enable;     // mask, not select: no scalar branch
    // row0 >= 0 and rs <= 2^24 (host check): one 32 x 32 -> 64-bit multiply instead of the generic 64-bit one
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(head_base + (uint64_t)(uint32_t)row0 * (uint32_t)rs), 0, bytes, 0x00020000);
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

// one float4 of the output.  Plain stores: write-through (sc1) stores of O measured +10 us on BASELINE config 2,
// non-temporal ones +0.45 us (tools/ab_run.sh).
__device__ __forceinline__ void store_o(float *ptr, f32x4 v, bool in_range) {
    if (in_range) *reinterpret_cast<f32x4 *>(ptr) = v;
}

// kSegs = 2: the analytic plan (one block or a pair per workgroup); 4: an explicit table (Params::seg_table).
// kAcquire: the last-arriving workgroup of a split query block runs an agent-scope acquire before it reads the other
// parts (workspace in a __device__ array: the no-allocation entry point); without it every load of a part is an sc1 load
// of sc1-stored, drained bytes in hipMalloc memory, one workgroup per CU (MI355X guide, visibility table row 1).
// kWaves = 1: a workgroup is ONE wave that runs its (up to two) query blocks one after the other, whole -- no split-KV,
// no shared LDS, no barrier, O straight from the accumulators.  For grids of many rounds: every wave slot of the chip is
// refilled on its own, and with the pairing n <-> nqb+1-n every wave runs the same nqb+1 units.
// kList (one-wave workgroups only): the wave runs a LIST of up to 16 whole query blocks of SEVERAL (b,h)s, one after the
// other, with the K / V prefetch stream and the query prefetch running on across every boundary.  The lists come from the
// host (Params::seg_table: 8 words = 16 half-word entries per list, `units | (b,h) within the period << 8`, 0 = end).  Blocks
// of at most list_rev_max units are taken in DESCENDING order, from the diagonal unit down to unit 0: the waves of a team
// whose (short) blocks END together then read the same K / V unit at the same time, as the waves whose (long) blocks START
// together do in ascending order.  The order is a function of the block's length alone, so a block's result does not depend
// on the list it is on (equal prompts in a batch give bit-equal rows).  A
// PERIOD of list_p consecutive (b,h)s is dealt to list_q waves so that every wave carries (almost) the same number of
// units and the whole launch is ONE round of equally long waves -- no round quantisation, one prologue per wave
// instead of one per 17-unit job.  Grid = (8, list_q, periods / 8): x is the XCD, so a period's waves share an L2.
template <int kWaves, int kSegs, bool kAcquire, bool kList = false>
__global__ __launch_bounds__(kWaves * 64, 2) void cmhsa_fwd_kernel(Params p) {
    static_assert(!kList || (kWaves == 1 && kSegs == 2 && !kAcquire), "job lists: one-wave workgroups");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;      // which k-slot of the MFMA operand this lane feeds
#ifdef CMHSA_STAMPS
    const int lin_block = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const bool stamp_on = g_stamp_buf != nullptr && lin_block == g_stamp_block && lane == 0;
    unsigned long long *stamp_ptr = g_stamp_buf + wave * 128;
    int stamp_i = 4;
    STAMP(0);
    if (g_block_times && threadIdx.x == 0) g_block_times[2 * lin_block] = __builtin_amdgcn_s_memrealtime();
#endif
    const int l31 = lane & 31;       // query (B/C column) or key / d-pair (A row) index of this lane

    // Pull every kernel argument into SGPRs NOW, with one wait: left alone, hipcc sinks the scalar loads into
    // the blocks that use them -- five dependent round trips to a cold kernarg segment (~2 k cycles, measured)
    // before the first global load could be issued.
    // (the table variant's eight words of this workgroup ride in the same batch: their address needs blockIdx.y only)
    unsigned tw[2 * kSegs];
    if constexpr (kSegs > 2) {
#pragma unroll
        for (int i = 0; i < 2 * kSegs; ++i) tw[i] = p.seg_table[blockIdx.y * (2 * kSegs) + i];
        if constexpr (kSegs > 4)
            asm volatile("" ::"s"(tw[8]), "s"(tw[9]), "s"(tw[10]), "s"(tw[11]), "s"(tw[12]), "s"(tw[13]), "s"(tw[14]), "s"(tw[15]));
        asm volatile("" ::"s"(tw[0]), "s"(tw[1]), "s"(tw[2]), "s"(tw[3]), "s"(tw[4]), "s"(tw[5]), "s"(tw[6]), "s"(tw[7]),
                     "s"(p.q), "s"(p.k), "s"(p.v), "s"(p.o), "s"(p.q_rs), "s"(p.k_rs), "s"(p.v_rs), "s"(p.o_rs));
    } else
    asm volatile("" ::"s"(p.q), "s"(p.k), "s"(p.v), "s"(p.o), "s"(p.q_rs), "s"(p.k_rs), "s"(p.v_rs), "s"(p.o_rs));
    unsigned lw[8];                             // kList: this wave's list, as it stands in the table
    if constexpr (kList) {
#pragma unroll
        for (int i = 0; i < 8; ++i) lw[i] = p.seg_table[blockIdx.y * 8 + i];
        asm volatile("" ::"s"(lw[0]), "s"(lw[1]), "s"(lw[2]), "s"(lw[3]), "s"(lw[4]), "s"(lw[5]), "s"(lw[6]), "s"(lw[7]), "s"(p.list_p), "s"(p.list_periods), "s"(p.list_rev_max));
    }
    asm volatile("" ::"s"(p.q_bs), "s"(p.k_bs), "s"(p.v_bs), "s"(p.o_bs), "s"(p.seq_len), "s"(p.num_heads), "s"(p.nqb),
                 "s"(p.grid_x), "s"(p.heads_magic), "s"(p.chain_magic), "s"(p.chain), "s"(p.n_paired), "s"(p.n_alone), "s"(p.use_table));
    STAMP(124);

    // ---- grid -> (batch*head, workgroup j of that head).  Workgroups are dispatched in linear order, x fastest,
    //      round-robin over the 8 XCDs: with grid = (8, wgs_per_bh, bh/8) every XCD keeps its own (b,h)s (K/V
    //      stay in that XCD's L2), with grid = (bh, wgs_per_bh, 1) otherwise; either way j = 0 (heaviest) first.
    int bh = blockIdx.z * p.grid_x + blockIdx.x;
    const int j = blockIdx.y;
    if constexpr (kWaves == 1 && !kList) {
        if (bh >= p.bh_count) return;            // padding of the longest-first grid (no barrier in this variant)
    }
    // kList: the list's entries whose (b,h) exists (the last period may be short), packed into the lanes of one register:
    // job i = lane i, wave-uniform and dynamic; lanes past the last job hold 0 = "no job"
    unsigned jobs_v = 0;
    int period_bh0 = 0;
    if constexpr (kList) {
        const int period = blockIdx.z * 8 + blockIdx.x;
        if (period >= p.list_periods) return;   // padding of the grid's z (no barrier in this variant)
        period_bh0 = period * p.list_p;
        const unsigned limit = (unsigned)(p.bh_count - period_bh0);
        int njobs = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned e = (lw[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            if ((e & 255u) != 0u && (e >> 8) < limit) {
                jobs_v = lane == njobs ? e : jobs_v;
                ++njobs;
            }
        }
        if (njobs == 0) return;                  // (a list whose every (b,h) is past the end)
        bh = period_bh0 + (int)((unsigned)__builtin_amdgcn_readlane((int)jobs_v, 0) >> 8);
    }
    // ---- workgroup j -> up to kSegs query blocks ("segments"), this wave -> (segment, local wave index) ----
    int seg_n[kSegs];                           // units (= query block index + 1) of each segment's query block; 0 = none
    int seg_ulo[kSegs], seg_cnt[kSegs];          // the segment covers units [ulo, ulo + cnt) of it (the whole block: 0, n)
    unsigned seg_mask[kSegs], seg_x[kSegs];      // waves working on it (bit w = wave w); table word 1 (0 = not a part)
    if constexpr (kList) {
#pragma unroll
        for (int i = 0; i < kSegs; ++i) { seg_n[i] = 0; seg_ulo[i] = 0; seg_cnt[i] = 0; seg_mask[i] = 0; seg_x[i] = 0; }
        seg_n[0] = (int)((unsigned)__builtin_amdgcn_readlane((int)jobs_v, 0) & 255u);
    } else
    decode_segments<kSegs>(tw, j, p.nqb, p.n_paired, p.n_alone, p.chain, p.chain_magic, seg_n, seg_ulo, seg_cnt, seg_mask, seg_x);
    // this wave -> (segment, index among the segment's waves); defaults = idle wave: barriers and row work only
    int u_first = 0, u_end = 0, qb = -1, seg_waves = 1, seg = 0, n_seg = 0;
    bool active = false;
#pragma unroll
    for (int i = 0; i < kSegs; ++i) {
        if ((seg_mask[i] >> wave) & 1u) {
            seg = i;
            seg_waves = __builtin_popcount(seg_mask[i]);
            u_first = seg_ulo[i] + __builtin_popcount(seg_mask[i] & ((1u << wave) - 1u));
            u_end = seg_ulo[i] + seg_cnt[i];
            qb = seg_n[i] - 1;                   // the diagonal unit; causal early exit: units past it are never touched (CMHSA.cu:133-136)
            active = true;
        }
        n_seg += seg_n[i] ? 1 : 0;
    }
    if constexpr (kWaves == 1) active = true;    // (its segments are set up pass by pass below; the masks are not used)
    const int b = udiv_magic(bh, p.num_heads, p.heads_magic), h = bh - b * p.num_heads;
    const int N = p.seq_len;

    const float *qg = p.q + (uint64_t)(uint32_t)b * (uint64_t)p.q_bs + (uint32_t)h * kHeadDim;
    const float *kg = p.k + (uint64_t)(uint32_t)b * (uint64_t)p.k_bs + (uint32_t)h * kHeadDim;
    const float *vg = p.v + (uint64_t)(uint32_t)b * (uint64_t)p.v_bs + (uint32_t)h * kHeadDim;
    // kList: the two jobs after the current one (units; 0 = none) -- the prefetch stream looks two units ahead, and a job
    // may be one unit long -- and where each finds its (b,h)
    struct JobRef { int n; bool rev; const float *q, *k, *v; float *o; };
    auto job_ref = [&](int i) {
        const unsigned e = (unsigned)__builtin_amdgcn_readlane((int)jobs_v, i);
        const int jbh = period_bh0 + (int)(e >> 8);
        const int jb = udiv_magic(jbh, p.num_heads, p.heads_magic), jh = jbh - jb * p.num_heads;
        JobRef r;
        r.n = (int)(e & 255u);
        r.rev = r.n <= p.list_rev_max;
        r.q = p.q + (uint64_t)(uint32_t)jb * (uint64_t)p.q_bs + (uint32_t)jh * kHeadDim;
        r.k = p.k + (uint64_t)(uint32_t)jb * (uint64_t)p.k_bs + (uint32_t)jh * kHeadDim;
        r.v = p.v + (uint64_t)(uint32_t)jb * (uint64_t)p.v_bs + (uint32_t)jh * kHeadDim;
        r.o = p.o + (uint64_t)(uint32_t)jb * (uint64_t)p.o_bs + (uint32_t)jh * kHeadDim;
        return r;
    };
    JobRef job1 = {0, false, qg, kg, vg, nullptr}, job2 = job1;
    bool cur_rev = false;                        // the current job takes its units in descending order
    if constexpr (kList) {
        job1 = job_ref(1); job2 = job_ref(2);
        cur_rev = seg_n[0] <= p.list_rev_max;
    }

    float *wlds = lds + wave * kUnitFloats;      // private to this wave: its current K unit (8 KiB); later its merge image
    float *qlds_all = lds + lds_floats_q(kWaves);                      // shared: the segments' query blocks,
    float *qlds = qlds_all + seg * kUnitFloats;                        // pre-scaled, K-style swizzle
    float *ml_all = lds + lds_floats_ml(kWaves, kSegs);                       // [wave][query][m, l]

    // per-lane byte offsets inside a unit
    const int sub = lane >> 4, chunk = lane & 15;
    const int koff = (sub * (int)p.k_rs + chunk * 4) * 4, kstep = 16 * (int)p.k_rs;   // K: coalesced rows -> LDS

    // ---- prologue.  One memory round trip ahead of the first MFMA: every wave requests its first K unit, then
    //      its share of the query blocks, and waits for both together.  Each segment's query block is fetched
    //      ONCE: its 8 4-row groups go round-robin over ALL waves of the workgroup, are scaled by
    //      log2(e)/sqrt(64) and parked in LDS; every wave reads its B-operand fragments from there.
    //      (Inactive waves and absent segments use zero-record descriptors: no memory touched, no branch.
    //      Giving the query blocks to waves 0-3 only, so that the barrier opens earlier, measured no faster.) ----
    f32x4 kst[8];                    // K staging registers (always one unit ahead of LDS)
    // (one-wave workgroups fetch their query fragments straight from global memory, below.  The same in 8-wave
    //  workgroups -- no staging, no barrier here -- measured +0.6 us on BASELINE config 2: 64 strided 16-B pieces per load)
    constexpr bool kDirectQ = kWaves == 1;
    if constexpr (!kDirectQ) {
    load_unit(unit_rsrc(kg, p.k_rs, u_first * kKeyUnit, N, active), koff, kstep, kst);
    {
        constexpr int kQIters = kSegs * 8 / kWaves;
        f32x4 qst[kQIters];
#pragma unroll
        for (int i = 0; i < kQIters; ++i) {
            // wave < kWaves <= 8: segment and row group of (wave + i * kWaves) split at compile time
            const int sg = (i * kWaves) >> 3, row = 4 * (wave + ((i * kWaves) & 7)) + sub;
            // plain global loads, two VALU instructions of address arithmetic each (a buffer descriptor per segment cost
            // ~30 scalar instructions in front of the barrier).  Rows past the end of a ragged last block, and absent
            // segments, read a valid row instead of zeros: their results are never stored.  q_rs <= 2^24 (host check).
            const int grow = min((max(seg_n[sg], 1) - 1) * kQBlock + row, N - 1);
            const float *src = qg + (uint64_t)(uint32_t)grow * (uint32_t)p.q_rs + chunk * 4;
            qst[i] = *reinterpret_cast<const f32x4 *>(src);
        }
        STAMP(121);
#pragma unroll
        for (int i = 0; i < kQIters; ++i) {
            const int sg = (i * kWaves) >> 3, row = 4 * (wave + ((i * kWaves) & 7)) + sub;
            *reinterpret_cast<f32x4 *>(qlds_all + sg * kUnitFloats + row * kHeadDim + ((chunk ^ (row & 15)) << 2)) = qst[i] * kScaleLog2;
        }
    }
    store_unit<true>(wlds, lane, kst);
    STAMP(122);
    __syncthreads();                 // query blocks complete in LDS
    STAMP(123);
    }

    // O^T accumulators.  MFMA row i of o0 / o1 is d = 2i / 2i+1, so that one dwordx2 per lane fetches both
    // A operands of a PV k-step and each half-wave reads one whole contiguous 256-B V row:
    //   o0[r] <-> d = 2*(acc_row(r) + 4*half),  o1[r] <-> d + 1,  query = l31.
    f32x16 o0 = {0}, o1 = {0};
    float m_run = -INFINITY;         // running max (log2 domain), same in both lane halves
    float l_run = 0.f;               // running sum over THIS lane half's keys only (merged at the end)

    // one-wave workgroups: the wave runs segment 0 whole, then segment 1 (n_after units; 0 = none).  Its K/V prefetch
    // stream runs on ACROSS the boundary: the units after the last one of a block are the first ones of the next.
    int n_after = 0;
    if constexpr (kWaves == 1) {
        u_first = 0; u_end = seg_n[0]; qb = u_end - 1; seg_waves = 1;
        n_after = seg_n[1];
    }
    if (active) {
        const int vrs4 = (int)p.v_rs * 4;
        const int voff = half * 4 * vrs4 + l31 * 8;                                       // V: straight into MFMA layout
        const float *krow0 = wlds + l31 * kHeadDim;
        const int sw = l31 & 15;
        auto kfrag = [&](int c) {
            return *reinterpret_cast<const f32x4 *>(krow0 + (((2 * c + half) ^ sw) << 2));
        };

        // All per-lane load offsets are computed once: VALU instructions are NOT free beside fp32 MFMAs (they
        // share the SIMD's fp32 datapath: measured cost = sum of both), so the main loop carries as few as possible.
        int koffs[8], voffs[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) koffs[i] = koff + i * kstep;
#pragma unroll
        for (int r = 0; r < 16; ++r) voffs[r] = voff + acc_row(r) * vrs4;
        f32x2 vreg[16];              // V of the current unit, MFMA A-operand layout: [k-step r] = {d even, d odd}
        float qf[32];                // Q^T fragment: B operand of S^T = K Q^T; k-step 4c+j uses d = 8c + 4*half + j
        if constexpr (kDirectQ) {
            // this pass's first K unit and the query block's fragments straight from global memory in operand layout
            // (lane = query: eight 16-B pieces of its row), one round trip for both
            load_unit(unit_rsrc(kg, p.k_rs, (kList && cur_rev ? u_end - 1 : u_first) * kKeyUnit, N), koff, kstep, kst);
            const __amdgpu_buffer_rsrc_t qsrc = unit_rsrc(qg, p.q_rs, qb * kQBlock, N);
            f32x4 qt[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                qt[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qsrc, (l31 * (int)p.q_rs + 8 * c + 4 * half) * 4, 0, 0));
            store_unit<true>(wlds, lane, kst);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) qf[4 * c + j] = qt[c][j] * kScaleLog2;
        } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(qlds + l31 * kHeadDim + (((2 * c + half) ^ sw) << 2));
#pragma unroll
            for (int j = 0; j < 4; ++j) qf[4 * c + j] = t[j];
        }
        }
        f32x4 kcur = kfrag(0), knext = kfrag(1);
        STAMP(1);

        // descriptor of the unit this wave processes d steps after unit u (zero records when there is none).  One-wave
        // workgroups: past the end of the block it is a unit of the wave's NEXT block.
        //  With job lists the next block is another (b,h)'s as a rule, and the one after it is reached when the next is a single unit.
        auto ahead = [&](const float *base, int64_t rs, int u, int d, bool is_v = false) {
            int t = u + d * seg_waves;
            bool en = t < u_end;
            if constexpr (kList) {               // (u counts the job's units in the order it takes them)
                int n = u_end;
                bool rev = cur_rev;
                if (!en) {
                    t -= u_end; base = is_v ? job1.v : job1.k; en = t < job1.n; n = job1.n; rev = job1.rev;
                    if (!en) { t -= job1.n; base = is_v ? job2.v : job2.k; en = t < job2.n; n = job2.n; rev = job2.rev; }
                }
                if (rev) t = max(n - 1 - t, 0);
            } else if constexpr (kWaves == 1) {
                if (!en) { t -= u_end; en = t < n_after; }
            }
            return unit_rsrc(base, rs, t * kKeyUnit, N, en);
        };

        // One unit = QK chain, softmax, PV chain.  `first` (compile time): the wave's first unit, whose QK chain
        // also carries the loads the steady state issues one unit ahead -- V(u) and K(u + seg_waves) -- so that
        // the first MFMA does not queue behind 24 more buffer loads per wave (a CU's L1 serves 64 B/clk).
        //  `last` (compile time, one-wave workgroups): the last unit of a block that another block follows: the query
        //  fragments of that next block are requested right behind the QK chain -- the registers of this block's fragments
        //  are free from there on -- and land under the softmax and the PV chain (requested after the chain, at the block
        //  boundary, they cost the wave a whole memory round trip per block: 6-8 k cycles, measured).
        f32x4 qt[8];                 // raw query fragments of the wave's next block
        auto unit = [&](const int u, auto first, auto last) __attribute__((always_inline)) {
            constexpr bool kFirst = decltype(first)::value, kLast = decltype(last)::value;
#ifdef CMHSA_STAMPS
            STAMP(stamp_i);
#endif
            // ---- S^T[key][query] = sum_d K[key][d] * Q[query][d] : 32 chained MFMAs.  K fragments are read
            //      two steps (512 MFMA cycles) ahead. ----
            f32x16 s = {0};
            const int ua = kList && cur_rev ? u_end - 1 - u : u;       // the unit's index in its block
            __amdgpu_buffer_rsrc_t vsrc0, ksrc1;
            if constexpr (kFirst) {
                vsrc0 = unit_rsrc(vg, p.v_rs, ua * kKeyUnit, N);
                ksrc1 = ahead(kg, p.k_rs, u, 1);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 kk = kcur;
                kcur = knext;
                if (c + 2 < 8) knext = kfrag(c + 2);
                if constexpr (kFirst) {
                    vreg[2 * c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(vsrc0, voffs[2 * c], 0, 0));
                    vreg[2 * c + 1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(vsrc0, voffs[2 * c + 1], 0, 0));
                    kst[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ksrc1, koffs[c], 0, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[j], qf[4 * c + j], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 1);
#endif
            if constexpr (kLast) {
                const int n_next = kList ? job1.n : n_after;
                const __amdgpu_buffer_rsrc_t qsrc = unit_rsrc(kList ? job1.q : qg, p.q_rs, max(n_next - 1, 0) * kQBlock, N, n_next != 0);
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    qt[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qsrc, (l31 * (int)p.q_rs + 8 * c + 4 * half) * 4, 0, 0));
            }
            // ---- causal mask: only the diagonal unit has key > query pairs (CMHSA.cu:228, :249) ----
            if (ua == qb) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc_row(r) + 4 * half > l31) s[r] = -INFINITY;
            }

            // ---- online softmax, log2 domain; every processed unit has >=1 unmasked key per row.
            //      Speculative: P = exp2(S - m_run) is computed against the OLD running max without looking at the
            //      unit's own max (16 max + a lane swap saved per unit; VALU work is paid in full beside fp32 MFMAs).
            //      fp32 has the exponent range: as long as no row sum exceeds kRescaleSum = 2^64 nothing can
            //      overflow later (<= 2^64 * 32 keys * units * |V|), and a stale max costs no precision (floating point
            //      is scale-free).  Only if some row's sum is above the threshold -- or not finite: always on a
            //      wave's first unit, where m_run = -inf -- the max is computed, O and l are rescaled and P is redone.
            // the two waves of a SIMD: the one in its softmax (VALU) phase yields to the one feeding the matrix pipe
            // (s_setprio 0 here, 2 again below: measured -1.5 % on BASELINE config 2, neutral on long grids)
            if constexpr (!kFirst) __builtin_amdgcn_s_setprio(0);    // (the first unit, which every wave reaches together, stays high: measured)
            __builtin_amdgcn_sched_barrier(0);           // (s_setprio is no scheduling barrier of its own)
            f32x16 pv;
            f32x2 psum2 = {0.f, 0.f};                 // packed fp32 ops: two elements per VALU issue slot
            const bool fresh = kFirst || (kWaves == 1 && u == 0);     // a block's first unit goes straight to the max path below
            if (!fresh) {
                const f32x2 m2 = {m_run, m_run};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 t = {s[r], s[r + 1]};
                    t -= m2;
                    t[0] = __builtin_amdgcn_exp2f(t[0]);
                    t[1] = __builtin_amdgcn_exp2f(t[1]);
                    psum2 += t;
                    pv[r] = t[0];
                    pv[r + 1] = t[1];
                }
            }
            float psum = psum2[0] + psum2[1];
            if (fresh || !__all(psum < kRescaleSum)) {           // NaN and inf compare false: they take this path too
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                mx = xhalf_max(mx);
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // 0 on the first unit (m_run = -inf)
                l_run *= alpha;
                m_run = m_new;
                if (!fresh) {                  // (a block's first unit: O is zero already -- 32 multiplies that each wait for a gap
                                               //  in the SIMD partner's MFMA stream)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o0[r] *= alpha;
                        o1[r] *= alpha;
                    }
                }
                psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
                    psum += pv[r];
                }
            }
            l_run += psum;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(2);
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 2);
#endif
            // ---- O^T[d][query] += sum_key V[key][d] * P[query][key] : 32 MFMAs, all operands in registers.
            //      k-step r pairs keys acc_row(r) (lanes 0-31) and acc_row(r)+4 (lanes 32-63), which is exactly
            //      where accumulator register r of S^T holds them: P feeds the B operand as it stands.
            //      Interleaved with the chain (program order = issue order, one item per MFMA pair):
            //        steps 0-7  : staged K(u+1) registers -> the wave's LDS unit.  ONE buffer is enough: every
            //                     fragment of K(u) was read before the QK chain's last MFMAs were issued, and a
            //                     wave's LDS operations execute in order (a second buffer only cost address VALU:
            //                     measured -3 % time without it)
            //        steps 8-15 : K(u+2) global -> staging registers;  steps 13,14: first K(u+1) fragments ----
            //      Everything is unconditional (past the last unit the descriptors have zero records and the
            //      LDS unit receives zeros nobody reads), so the chain stays one basic block.
            const __amdgpu_buffer_rsrc_t ksrc2 = ahead(kg, p.k_rs, u, 2);
            const __amdgpu_buffer_rsrc_t vsrc1 = ahead(vg, p.v_rs, u, 1, true);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r < 8) {
                    const int row = 4 * r + sub;
                    *reinterpret_cast<f32x4 *>(wlds + row * kHeadDim + ((chunk ^ (row & 15)) << 2)) = kst[r];
                } else {
                    kst[r - 8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ksrc2, koffs[r - 8], 0, 0));
                    if (r == 13) kcur = kfrag(0);
                    if (r == 14) knext = kfrag(1);
                }
                if (r == 8) {   // all K(u+1) stores are queued: order them before this wave's later LDS reads
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[r][0], pv[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[r][1], pv[r], o1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // V(u+1), k-step r: its operand registers are free as soon as the two MFMAs above have issued;
                // the load has the whole next QK chain + softmax to land
                vreg[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(vsrc1, voffs[r], 0, 0));
            }
#ifdef CMHSA_STAMPS
            STAMP(stamp_i + 3);
            stamp_i += 4;
#endif
        };

        __builtin_amdgcn_s_setprio(2);
        unit(u_first, std::true_type{}, std::false_type{});      // every active wave owns at least one unit (a segment has no more waves than units)
        if constexpr (kWaves > 1) {
            for (int u = u_first + seg_waves; u < u_end; u += seg_waves) unit(u, std::false_type{}, std::false_type{});
            l_run = xhalf_sum(l_run);
        } else {
            float *__restrict__ og = p.o + (uint64_t)(uint32_t)b * (uint64_t)p.o_bs + (uint32_t)h * kHeadDim;
            int u = 1;
            int ji = 0;                          // kList: index of the current job in the list
            for (;;) {
                for (; u + 1 < u_end; ++u) unit(u, std::false_type{}, std::false_type{});
                // the block's last unit, with the next block's query fragments requested inside it -- unless the block was
                // the kernel's first and a single unit (its only unit ran above): then they are requested here
                const bool prefetched = u < u_end;
                if (prefetched) unit(u, std::false_type{}, std::true_type{});
#ifdef CMHSA_STAMPS
                STAMP(stamp_i);          // a block boundary shows as a row of its own: (chain issued, O stores issued, fragments scaled, 0)
#endif
                // the whole block was this wave's: normalise and store, lane = query, 4 consecutive d per store (the
                // merge's pairs).  The next block's first K unit is in LDS and its first V unit in flight already; its
                // query fragments are requested BEFORE the stores (the registers of this block's are free now)
                const int n_next = kList ? job1.n : n_after;
                if (n_next && !prefetched) {
                    const __amdgpu_buffer_rsrc_t qsrc = unit_rsrc(kList ? job1.q : qg, p.q_rs, (n_next - 1) * kQBlock, N);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        qt[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qsrc, (l31 * (int)p.q_rs + 8 * c + 4 * half) * 4, 0, 0));
                }
                const float inv = 1.0f / xhalf_sum(l_run);
                const int row = qb * kQBlock + l31;
#pragma unroll
                for (int pi = 0; pi < 8; ++pi) {
                    const f32x4 t = {o0[2 * pi], o1[2 * pi], o0[2 * pi + 1], o1[2 * pi + 1]};
                    const int col = 2 * (2 * (pi & 1) + 8 * (pi >> 1) + 4 * half);
                    store_o(og + (int64_t)row * p.o_rs + col, t * inv, row < N);
                }
#ifdef CMHSA_STAMPS
                STAMP(stamp_i + 1);
#endif
                if (!n_next) break;
                // fresh accumulators; m_run = -inf sends the block's first unit down the max path (its row sums are not finite)
#pragma unroll
                for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
                m_run = -INFINITY;
                l_run = 0.f;
                u_end = n_next; qb = n_next - 1; n_after = 0; u = 0;
                if constexpr (kList) {           // the list moves up by one: the bases every later load and store goes through
                    kg = job1.k; vg = job1.v; og = job1.o; cur_rev = job1.rev;
                    job1 = job2;
                    job2 = job_ref(ji + 3);
                    ++ji;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) qf[4 * c + j] = qt[c][j] * kScaleLog2;
#ifdef CMHSA_STAMPS
                STAMP(stamp_i + 2);
                if (stamp_on && stamp_i + 3 < 128) stamp_ptr[stamp_i + 3] = 0;
                stamp_i += 4;
#endif
            }
        }
    }
#ifdef CMHSA_STAMPS
    STAMP(2);
#endif
    if constexpr (kWaves > 1) {

    // ---- merge the split-KV partials of each segment: ONE barrier, no transposition.
    //      Every active wave publishes its un-normalised O accumulators in their native MFMA layout, as 8
    //      float4 "pairs" per lane -- pair pi = {o0[2pi], o1[2pi], o0[2pi+1], o1[2pi+1]} = 4 consecutive d
    //      of the lane's query -- plus its (m, l).  After the barrier the 8 (segment, pair) items per segment go
    //      round-robin over ALL waves: an item reads the pair of every wave of its segment, weights it with
    //      exp2(m_w - m_tot), divides by the merged l and stores one float4 per lane straight to O. ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (active) {
        f32x4 *pub = reinterpret_cast<f32x4 *>(wlds);
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) {
            const f32x4 t = {o0[2 * pi], o1[2 * pi], o0[2 * pi + 1], o1[2 * pi + 1]};
            pub[pi * 64 + lane] = t;
        }
    }
    // EVERY wave publishes its (m, l) -- an idle wave (no segment) its initial (-inf, 0): the merge reads all kWaves
    // slots and multiplies l by a weight of 0 for waves outside the segment, so a slot must never hold leftover LDS
    // bits of an earlier kernel (inf or NaN patterns: 0 * NaN would poison the whole row block).  The idle wave's O
    // image is zero already: its K unit was stored from a zero-record descriptor in the prologue.
    if (lane < 32) *reinterpret_cast<f32x2 *>(ml_all + wave * 64 + 2 * lane) = f32x2{m_run, l_run};
    __syncthreads();
    STAMP(120);
    {
        float *__restrict__ og = p.o + (uint64_t)(uint32_t)b * (uint64_t)p.o_bs + (uint32_t)h * kHeadDim;
        // Items (segment, pair) with item = 8 * segment + pair go round-robin over the waves, so a wave's items share
        // their pair index (8 waves: pair = wave; 4 waves: pair = wave and wave + 4): the image of every wave is read
        // ONCE per pair and serves all segments -- each publishing wave belongs to exactly one of them.
#pragma unroll
        for (int pc = 0; pc < 8 / kWaves; ++pc) {
            const int pi = wave + pc * kWaves;
            float mv[kWaves], lv[kWaves];
            u32x4 ov[kWaves];
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const f32x2 ml = *reinterpret_cast<const f32x2 *>(ml_all + w * 64 + 2 * l31);
                mv[w] = ml[0];
                lv[w] = ml[1];
                ov[w] = *reinterpret_cast<const u32x4 *>(lds + w * kUnitFloats + (pi * 64 + lane) * 4);
            }
            const int col = 2 * (2 * (pi & 1) + 8 * (pi >> 1) + 4 * half);       // = 2 * (acc_row(2 pi) + 4 half)
#pragma unroll
            for (int sg = 0; sg < kSegs; ++sg) {
                if (sg >= n_seg) break;                                          // wave-uniform
                // wave-uniform bit masks, not selects or branches, pick the segment's waves
                float mw[kWaves];
                float m_tot = -INFINITY;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const unsigned mine = 0u - ((seg_mask[sg] >> w) & 1u);
                    mw[w] = __uint_as_float((__float_as_uint(mv[w]) & mine) | (~mine & 0xff800000u));   // -inf when not mine
                    m_tot = fmaxf(m_tot, mw[w]);
                }
                float l_tot = 0.f;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const float f = __builtin_amdgcn_exp2f(mw[w] - m_tot);      // 0 for the other segments' waves
                    l_tot += lv[w] * f;
                    acc += __builtin_bit_cast(f32x4, ov[w]) * f;
                }
                if (kSegs > 2 && seg_x[sg] != 0) {                               // wave-uniform
                    // ONE PART of a query block whose keys are split across workgroups: the un-normalised (O, m, l)
                    // goes to the workspace in the layout of the LDS images (a wave's store covers eight whole 128-B
                    // lines), write-through (sc1), so that no release fence is needed before the arrival counter
                    const int slot = (int)(seg_x[sg] & 255u) + (int)((seg_x[sg] >> 8) & 15u);
                    float *img = p.ws_parts + ((int64_t)bh * p.parts_per_bh + slot) * kPartFloats;
                    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(img, 0, kPartFloats * 4, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), prs, (pi * 64 + lane) * 16, 0, kSc1);
                    if (pi == 0 && lane < 32)
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{m_tot, l_tot}), prs, kUnitFloats * 4 + lane * 8, 0, kSc1);
                } else {
                    acc *= 1.0f / l_tot;
                    const int row = (seg_n[sg] - 1) * kQBlock + l31;
                    store_o(og + (int64_t)row * p.o_rs + col, acc, row < N);
                }
            }
        }
        // ---- query blocks split across workgroups: count this workgroup in; the one that arrives LAST combines.
        //      No workgroup ever waits for another (no spin, no residency assumption): every part's workgroup stores
        //      its partial (sc1, above), every storing wave drains its stores, the workgroup meets at a barrier, ONE
        //      lane per part adds to the block's arrival counter (agent scope, returning) and the workgroup whose add
        //      returns parts-1 reads all parts back (sc1 loads straight to registers; behind an agent-scope acquire
        //      when kAcquire) and combines them in PART ORDER, whoever it is -- the result bits do not depend on the
        //      arrival order.  The last arriver also puts the counter back to 0 for the next launch. ----
        if constexpr (kSegs > 2) {
            unsigned any_part = 0;
#pragma unroll
            for (int sg = 0; sg < kSegs; ++sg) any_part |= seg_x[sg];
            if (any_part) {                                                      // workgroup-uniform
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // EVERY storing wave: its sc1 stores have left
                __syncthreads();
                unsigned *flags = reinterpret_cast<unsigned *>(lds + lds_floats_flags(kWaves, kSegs));
                if (threadIdx.x < kSegs) {
                    const int t = threadIdx.x;
                    unsigned x = 0;
#pragma unroll
                    for (int i = 0; i < kSegs; ++i) x = t == i ? seg_x[i] : x;
                    unsigned last = 0;
                    if (x) {
                        unsigned *cnt = p.ws_counters + (int64_t)bh * p.splits_per_bh + (x >> 16);
                        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        last = old + 1u == ((x >> 12) & 15u) ? 1u : 0u;
                        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    flags[t] = last;
                }
                if (kAcquire && wave == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
#pragma unroll
                for (int sg = 0; sg < kSegs; ++sg) {
                    if (seg_x[sg] == 0) continue;                                // wave-uniform
                    if (__builtin_amdgcn_readfirstlane(flags[sg]) == 0) continue;    // another workgroup finishes this block
                    const int first = (int)(seg_x[sg] & 255u), nparts = (int)((seg_x[sg] >> 12) & 15u);
                    const float *base = p.ws_parts + ((int64_t)bh * p.parts_per_bh + first) * kPartFloats;
#pragma unroll
                    for (int pc = 0; pc < 8 / kWaves; ++pc) {
                        const int pi = wave + pc * kWaves;
                        float pm[kMaxParts], pl[kMaxParts];
                        f32x4 po[kMaxParts];
#pragma unroll
                        for (int q = 0; q < kMaxParts; ++q) {
                            pm[q] = -INFINITY; pl[q] = 0.f; po[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (q < nparts) {                                    // wave-uniform
                                const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                                    const_cast<float *>(base + (int64_t)q * kPartFloats), 0, kPartFloats * 4, 0x00020000);
                                const f32x2 ml = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(prs, kUnitFloats * 4 + l31 * 8, 0, kSc1));
                                pm[q] = ml[0];
                                pl[q] = ml[1];
                                po[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, (pi * 64 + lane) * 16, 0, kSc1));
                            }
                        }
                        float m_tot = pm[0];
#pragma unroll
                        for (int q = 1; q < kMaxParts; ++q) m_tot = fmaxf(m_tot, pm[q]);
                        float l_tot = 0.f;
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q = 0; q < kMaxParts; ++q) {
                            const float f = __builtin_amdgcn_exp2f(pm[q] - m_tot);  // 0 for absent parts (m = -inf)
                            l_tot += pl[q] * f;
                            acc += po[q] * f;
                        }
                        acc *= 1.0f / l_tot;
                        const int col = 2 * (2 * (pi & 1) + 8 * (pi >> 1) + 4 * half);
                        const int row = (seg_n[sg] - 1) * kQBlock + l31;
                        store_o(og + (int64_t)row * p.o_rs + col, acc, row < N);
                    }
                }
            }
        }
    }
    }   // kWaves > 1
#ifdef CMHSA_STAMPS
    STAMP(3);
    if (g_block_times && threadIdx.x == 0) g_block_times[2 * lin_block + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

std::atomic<int> g_attr_set[9][64];  // per variant and device: dynamic-LDS limit raised once, not per call

template <int kWaves, int kSegs, bool kAcquire, bool kList = false>
hipError_t launch(const Params &p, dim3 grid, void *stream) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<int> *flag = (dev >= 0 && dev < 64) ? &g_attr_set[kList ? 8 : kWaves == 1 ? 5 : kSegs == 8 ? 6 + kAcquire : kAcquire ? 4 : (kWaves == 8) * 2 + (kSegs > 2)][dev] : nullptr;
    if (!flag || !flag->load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(cmhsa_fwd_kernel<kWaves, kSegs, kAcquire, kList>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(kWaves, kSegs));
        if (e != hipSuccess) return e;
        if (flag) flag->store(1, std::memory_order_release);
    }
    // hipLaunchKernel returns this launch's own status: an error left behind by an unrelated earlier HIP call on
    // this thread is neither reported as ours nor cleared (hipGetLastError would do both)
    void *args[] = {const_cast<Params *>(&p)};
    return hipLaunchKernel(reinterpret_cast<const void *>(cmhsa_fwd_kernel<kWaves, kSegs, kAcquire, kList>), grid, dim3(kWaves * 64), args,
                           lds_bytes(kWaves, kSegs), static_cast<hipStream_t>(stream));
}

// 1 = the occupancy query says one workgroup of the 8-wave table variant (4 or 8 segments) per CU on this device; cached.
bool one_workgroup_per_cu(int dev, bool eight_segments) {
    static std::atomic<int> cached[2][64];                // 0 = not asked yet, 1 = yes, 2 = no
    std::atomic<int> *slot = (dev >= 0 && dev < 64) ? &cached[eight_segments ? 1 : 0][dev] : nullptr;
    int v = slot ? slot->load(std::memory_order_relaxed) : 0;
    if (!v) {
        int blocks = 0;
        const hipError_t e = eight_segments
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cmhsa_fwd_kernel<8, 8, false>, 512, lds_bytes(8, 8))
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, cmhsa_fwd_kernel<8, 4, false>, 512, lds_bytes(8, 4));
        v = (e == hipSuccess && blocks == 1) ? 1 : 2;
        if (slot) slot->store(v, std::memory_order_relaxed);
    }
    return v == 1;
}

// ---- the no-allocation entry point's scratch: a file-scope array in device memory.  In HOST code the name `scratch`
// is only a stand-in (not a device address): its address on each device comes from hipGetSymbolAddress, once per
// device, and only that pointer is ever handed to a kernel.  Zero-initialised at module load, like every arrival
// counter the kernel leaves behind.
static __device__ __align__(256) unsigned char scratch[CMHSA_NOALLOC_SCRATCH_BYTES];
static_assert(CMHSA_NOALLOC_SCRATCH_BYTES > kWsCounterBytes + kPartFloats * 4 && CMHSA_NOALLOC_SCRATCH_BYTES % 256 == 0, "scratch size");

std::mutex g_scratch_mu;
void *g_scratch_ptr[64];                         // per device: each GPU holds its own copy of the array at its own address

int scratch_of_current_device(void **ptr) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return CMHSA_ERR_TOO_LARGE;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));            // with device `dev` current: ITS copy of the array
        if (e != hipSuccess) return (int)e;
        if (!p || ((uintptr_t)p & 255)) return CMHSA_ERR_ALIGNMENT;
        g_scratch_ptr[dev] = p;
    }
    *ptr = g_scratch_ptr[dev];
    return CMHSA_OK;
}

// clock probe (cmhsa_probe_shader_clock): 4 independent MFMA chains per wave keep the matrix pipe saturated
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long *out4, int iters, float seed) {
    f32x16 a0 = {seed}, a1 = {seed + 1.f}, a2 = {seed + 2.f}, a3 = {seed + 3.f};
    const float x = seed + threadIdx.x, y = seed - threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    const float keep = a0[0] + a1[1] + a2[2] + a3[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out4[0] = t0; out4[1] = t1; out4[2] = r0; out4[3] = r1 + (keep == 12345.678f ? 1 : 0);   // keeps the chains alive
    }
}

}  // namespace

extern "C" {

int cmhsa_probe_shader_clock(unsigned long long *out4_dev, int iters, void *stream) {
    if (!out4_dev) return CMHSA_ERR_NULL_POINTER;
    if (iters <= 0 || ((uintptr_t)out4_dev & 7)) return CMHSA_ERR_BAD_SHAPE;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    float seed = 1.0f;
    void *args[] = {&out4_dev, &iters, &seed};
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(clock_probe_kernel), dim3((unsigned)(cus > 0 ? cus : 256) * 2),
                                         dim3(256), args, 0, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CMHSA_OK : (int)e;
}


const char *cmhsa_status_string(int status) {
    switch (status) {
        case CMHSA_OK: return "success";
        case CMHSA_ERR_NULL_POINTER: return "a tensor pointer is NULL";
        case CMHSA_ERR_BAD_SHAPE: return "batch, seq_len and num_heads must be non-negative and num_heads positive";
        case CMHSA_ERR_HEAD_DIM: return "Head dimension must be 64.";
        case CMHSA_ERR_ALIGNMENT: return "pointers must be 16-byte aligned and row/batch strides multiples of 4 for float4 loads to work";
        case CMHSA_ERR_TOO_LARGE: return "problem too large for 32-bit tile indexing";
        case CMHSA_ERR_WORKSPACE: return "workspace too small or not ordinary device memory (cmhsa_workspace_init: hipMalloc'd memory of at "
                                         "least 16 KiB, not managed / host-mapped; cmhsa_fp32_fwd_noalloc: this shape's plan needs more than "
                                         "the static CMHSA_NOALLOC_SCRATCH_BYTES; use cmhsa_fp32_fwd_ws with a workspace, or cmhsa_fp32_fwd)";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown cmhsa status";
    }
}

int cmhsa_version(void) { return (1 << 16) | 0; }

#ifdef CMHSA_STAMPS
// diagnostic build only: buf = device buffer of 8 waves x 128 stamps, block = blockIdx.x to stamp
int cmhsa_debug_set_stamps(unsigned long long *buf, int block) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_block), &block, sizeof(block));
    if (e == hipSuccess) e = (hipError_t)cmhsa_t16::debug_set_stamps(buf, block);
    return (int)e;
}
int cmhsa_debug_set_block_times(unsigned long long *buf) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_block_times), &buf, sizeof(buf));
    if (e == hipSuccess) e = (hipError_t)cmhsa_t16::debug_set_block_times(buf);
    return (int)e;
}
#endif

double cmhsa_algorithmic_flops(int batch, int seq_len, int num_heads, int head_dim) {
    return 2.0 * batch * num_heads * head_dim * (double)seq_len * ((double)seq_len + 1.0);
}

}  // extern "C"

namespace {

// Validation shared by every forward entry point; on success the shape in blocks and (b,h)s.
int check_forward_args(const float *q, const float *k, const float *v, float *o,
                       int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                       int64_t &q_bs, int64_t &k_bs, int64_t &v_bs, int64_t &o_bs,
                       int batch, int seq_len, int num_heads, int head_dim, int64_t &nqb, int64_t &bh, bool &empty) {
    empty = false;
    if (batch < 0 || seq_len < 0 || num_heads <= 0) return CMHSA_ERR_BAD_SHAPE;
    if (head_dim != kHeadDim) return CMHSA_ERR_HEAD_DIM;
    if (batch == 0 || seq_len == 0) { empty = true; return CMHSA_OK; }
    if (!q || !k || !v || !o) return CMHSA_ERR_NULL_POINTER;
    if (batch == 1) q_bs = k_bs = v_bs = o_bs = 0;
    const uintptr_t ptr_bits = (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o;
    const int64_t stride_bits = q_rs | k_rs | v_rs | o_rs | q_bs | k_bs | v_bs | o_bs;
    if ((ptr_bits & 15) || (stride_bits & 3)) return CMHSA_ERR_ALIGNMENT;
    if (stride_bits < 0) return CMHSA_ERR_BAD_SHAPE;             // a negative stride (sign bit set in the OR of all)
    nqb = ((int64_t)seq_len + kQBlock - 1) / kQBlock;
    bh = (int64_t)batch * num_heads;
    // 32-bit row arithmetic in the kernel: the software pipeline forms row numbers up to (nqb + 2 * 8 waves) * 32 and
    // multiplies none of them by a stride in 32 bits (row offsets are 64-bit), so the bound is on the row number itself
    if ((nqb + 16) * kKeyUnit > 0x7fffffffLL || bh > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;
    // 32-bit byte offsets inside a 32-row unit (buffer loads): 32 rows * row_stride * 4 B must fit
    const int64_t max_rs = q_rs > k_rs ? q_rs : k_rs;
    if ((max_rs > v_rs ? max_rs : v_rs) > (1LL << 24)) return CMHSA_ERR_TOO_LARGE;
    return CMHSA_OK;
}

// workspace: null = none (plans that split no block).  acquire: see the kernel template parameter.
int forward(const float *q, const float *k, const float *v, float *o,
            int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
            int64_t q_batch_stride, int64_t k_batch_stride, int64_t v_batch_stride, int64_t o_batch_stride,
            int batch, int seq_len, int num_heads, int head_dim, void *workspace, size_t workspace_bytes,
            bool reject_short_workspace, bool acquire, void *stream) {
    int64_t nqb = 0, bh = 0;
    bool empty = false;
    const int rc = check_forward_args(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride,
                                      q_batch_stride, k_batch_stride, v_batch_stride, o_batch_stride,
                                      batch, seq_len, num_heads, head_dim, nqb, bh, empty);
    if (rc != CMHSA_OK || empty) return rc;
    if (workspace && ((uintptr_t)workspace & 255)) return CMHSA_ERR_ALIGNMENT;

    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    const int cus = device_cu_count(dev);
    Plan plan = cached_plan(bh, seq_len, cus, workspace != nullptr);
    if (plan_workspace_bytes(plan, bh) > workspace_bytes) {
        if (reject_short_workspace) return CMHSA_ERR_WORKSPACE;
        plan = cached_plan(bh, seq_len, cus, false);              // a short workspace selects the plan that needs none
    }
    if (plan.tiles16) {                                           // the 16-row-block family: its own kernel and arguments (cmhsa_fwd16.hip)
        const int e16 = cmhsa_t16::launch(plan.t16, q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride,
                                          q_batch_stride, k_batch_stride, v_batch_stride, o_batch_stride, bh, seq_len, num_heads, stream);
        return e16 == (int)hipSuccess ? CMHSA_OK : e16;
    }
    const PackPlan &pl = plan.pp;
    const int waves = pl.waves;
    dim3 grid;
    const int grid_rc = plan_grid(plan, bh, grid);
    if (grid_rc != CMHSA_OK) return grid_rc;

    Params p;
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.q_rs = q_row_stride; p.k_rs = k_row_stride; p.v_rs = v_row_stride; p.o_rs = o_row_stride;
    p.q_bs = q_batch_stride; p.k_bs = k_batch_stride; p.v_bs = v_batch_stride; p.o_bs = o_batch_stride;
    p.seq_len = seq_len;
    p.num_heads = num_heads;
    p.nqb = (int)nqb;
    p.grid_x = (int)grid.x;
    p.bh_count = (int)bh;
    p.heads_magic = 0xffffffffu / (unsigned)num_heads;
    p.chain_magic = 0xffffffffu / (unsigned)pl.chain;
    p.chain = pl.chain; p.n_paired = pl.n_paired; p.n_alone = pl.n_alone;
    p.use_table = plan.use_table ? 1 : 0;
    const bool split = plan.parts_per_bh > 0;
    p.ws_counters = split ? static_cast<unsigned *>(workspace) : nullptr;
    p.ws_parts = split ? reinterpret_cast<float *>(static_cast<char *>(workspace) + kWsCounterBytes) : nullptr;
    p.parts_per_bh = plan.parts_per_bh;
    p.splits_per_bh = plan.splits_per_bh;
    if (plan.use_table) memcpy(p.seg_table, plan.table, sizeof(unsigned) * (size_t)plan.n_wgs * 2 * plan.segs);
    p.list_p = plan.list_p;
    p.list_rev_max = plan.list_rev_max;
    p.list_periods = plan.list_p ? (int)((bh + plan.list_p - 1) / plan.list_p) : 0;
    if (plan.list_p) memcpy(p.seg_table, plan.table, sizeof(unsigned) * (size_t)plan.list_q * 8);

    // The hand-off of a split block without an acquire (sc1 stores, drained, one counter add, sc1 loads: MI355X guide,
    // visibility table row 1) is measured for ONE workgroup per CU.  That holds by the LDS footprint of the 8-wave table
    // variants (98 / 130 KiB of 160); it is ASKED of the runtime once per device and variant, and if the answer is not 1
    // (another device, another LDS size) the merging workgroup runs the agent-scope acquire after all.
    if (split && !acquire) acquire = !one_workgroup_per_cu(dev, plan.segs == 8);
    hipError_t e;
    if (plan.list_p) e = launch<1, 2, false, true>(p, grid, stream);
    else if (waves == 1) e = launch<1, 2, false>(p, grid, stream);
    else if (!p.use_table) e = waves == 8 ? launch<8, 2, false>(p, grid, stream) : launch<4, 2, false>(p, grid, stream);
    else if (waves == 4) e = launch<4, 4, false>(p, grid, stream);
    else if (plan.segs == 8) e = split && acquire ? launch<8, 8, true>(p, grid, stream) : launch<8, 8, false>(p, grid, stream);
    else e = split && acquire ? launch<8, 4, true>(p, grid, stream) : launch<8, 4, false>(p, grid, stream);
    return e == hipSuccess ? CMHSA_OK : (int)e;     // launch-configuration errors only; no synchronisation (CMHSA.cu:450)
}

}  // namespace

extern "C" {

int cmhsa_fp32_fwd(const float *q, const float *k, const float *v, float *o,
                   int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                   int64_t q_batch_stride, int64_t k_batch_stride, int64_t v_batch_stride, int64_t o_batch_stride,
                   int batch, int seq_len, int num_heads, int head_dim, void *stream) {
    return forward(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_batch_stride, k_batch_stride,
                   v_batch_stride, o_batch_stride, batch, seq_len, num_heads, head_dim, nullptr, 0, false, false, stream);
}

int cmhsa_fp32_fwd_ws(const float *q, const float *k, const float *v, float *o,
                      int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                      int64_t q_batch_stride, int64_t k_batch_stride, int64_t v_batch_stride, int64_t o_batch_stride,
                      int batch, int seq_len, int num_heads, int head_dim,
                      void *workspace, size_t workspace_bytes, void *stream) {
    return forward(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_batch_stride, k_batch_stride,
                   v_batch_stride, o_batch_stride, batch, seq_len, num_heads, head_dim,
                   workspace_bytes ? workspace : nullptr, workspace ? workspace_bytes : 0, false, false, stream);
}

int cmhsa_fp32_fwd_noalloc(const float *q, const float *k, const float *v, float *o,
                           int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                           int64_t q_batch_stride, int64_t k_batch_stride, int64_t v_batch_stride, int64_t o_batch_stride,
                           int batch, int seq_len, int num_heads, int head_dim, void *stream) {
    void *ptr = nullptr;
    const int rc = scratch_of_current_device(&ptr);
    if (rc != CMHSA_OK) return rc;
    // a plan that needs more than the static array is REJECTED (CMHSA_ERR_WORKSPACE), not replaced; the workgroup that
    // merges a split block runs an agent-scope acquire first (the array is not hipMalloc memory)
    return forward(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, q_batch_stride, k_batch_stride,
                   v_batch_stride, o_batch_stride, batch, seq_len, num_heads, head_dim, ptr, CMHSA_NOALLOC_SCRATCH_BYTES,
                   true, true, stream);
}

size_t cmhsa_workspace_bytes(void) { return kWsMaxBytes; }

int cmhsa_workspace_init(void *workspace, size_t workspace_bytes, void *stream) {
    if (!workspace) return CMHSA_ERR_NULL_POINTER;
    if ((uintptr_t)workspace & 255) return CMHSA_ERR_ALIGNMENT;
    if (workspace_bytes < kWsCounterBytes) return CMHSA_ERR_WORKSPACE;
    // The hand-off between the workgroups that share a split block (sc1 stores, drained, one agent-scope counter add, sc1
    // loads, no fence) is sufficient for ordinary device memory -- coarse-grained hipMalloc memory, where an sc1 store
    // writes through to the memory side and an sc1 load misses the reader's L2 -- and is measured for that only (MI355X
    // guide, visibility table row 1).  Managed, fine-grained or host-mapped memory is refused here, once, instead of
    // silently merging stale partials.
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, workspace) != hipSuccess) {
        (void)hipGetLastError();                                     // (an unregistered pointer: not device memory either)
        return CMHSA_ERR_WORKSPACE;
    }
    if (attr.type != hipMemoryTypeDevice || attr.isManaged) return CMHSA_ERR_WORKSPACE;
    const hipError_t e = hipMemsetAsync(workspace, 0, kWsCounterBytes, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

size_t cmhsa_fp32_fwd_workspace_bytes(int batch, int seq_len, int num_heads) {
    if (batch <= 0 || seq_len <= 0 || num_heads <= 0) return 0;
    const int64_t nqb = ((int64_t)seq_len + kQBlock - 1) / kQBlock, bh = (int64_t)batch * num_heads;
    if ((nqb + 16) * kKeyUnit > 0x7fffffffLL || bh > 0x7fffffffLL) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    return plan_workspace_bytes(cached_plan(bh, seq_len, device_cu_count(dev), true), bh);
}

int cmhsa_plan_describe(int batch, int seq_len, int num_heads, int with_workspace, char *buf, int buf_len) {
    if (!buf || buf_len <= 0) return CMHSA_ERR_NULL_POINTER;
    buf[0] = 0;
    if (batch <= 0 || seq_len <= 0 || num_heads <= 0) return CMHSA_ERR_BAD_SHAPE;
    const int64_t nqb = ((int64_t)seq_len + kQBlock - 1) / kQBlock, bh = (int64_t)batch * num_heads;
    if ((nqb + 16) * kKeyUnit > 0x7fffffffLL || bh > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    const int cus = device_cu_count(dev);
    const Plan pl = cached_plan(bh, seq_len, cus, with_workspace != 0);
    if (pl.tiles16) {
        snprintf(buf, (size_t)buf_len, "%s: %d-wave workgroups, %d per (b,h) x %lld (b,h) on %d CUs, %d query blocks per (b,h), at most %d tiles "
                 "per SIMD and %d jobs per wave, %s grid, workspace 0 B, estimate %.0f cycles", pl.family, pl.t16.waves, pl.t16.wgs_per_bh,
                 (long long)bh, cus, pl.t16.blocks, pl.t16.tiles_per_simd, pl.t16.max_jobs, pl.t16.contig ? "XCD-contiguous" : "2-D", pl.cycles);
        return CMHSA_OK;
    }
    if (pl.list_p) {
        const long long periods = (long long)((bh + pl.list_p - 1) / pl.list_p);
        snprintf(buf, (size_t)buf_len, "%s: 1-wave workgroups, %d lists per period of %d (b,h) x %lld periods on %d CUs, %d query blocks per "
                 "(b,h), at most %d units and %d jobs per list, workspace 0 B, estimate %.0f cycles", pl.family, pl.list_q, pl.list_p, periods,
                 cus, (int)nqb, pl.list_units, pl.list_jobs, pl.cycles);
        return CMHSA_OK;
    }
    int n = snprintf(buf, (size_t)buf_len, "%s: %d-wave workgroups, %d per (b,h) x %lld (b,h) on %d CUs, chain cap %d, SIMD cap %d, "
                     "%d of %lld query blocks per (b,h) split across workgroups, workspace %zu B, estimate %.0f cycles",
                     pl.family, pl.pp.waves, pl.pp.wgs_per_bh, (long long)bh, cus, pl.pp.chain, pl.simd_cap, pl.splits_per_bh,
                     (long long)nqb, plan_workspace_bytes(pl, bh), pl.cycles);
    if (pl.parts_per_bh > 0 && n > 0 && n < buf_len && dev >= 0)
        snprintf(buf + n, (size_t)(buf_len - n), one_workgroup_per_cu(dev, pl.segs == 8)
                 ? "; hand-off without acquire (occupancy query: one workgroup per CU)"
                 : "; hand-off with agent-scope acquire (occupancy query: not one workgroup per CU)");
    return CMHSA_OK;
}

int cmhsa_plan_dump(int batch, int seq_len, int num_heads, int with_workspace, int cus, int *out, int cap) {
    if (batch <= 0 || seq_len <= 0 || num_heads <= 0) return CMHSA_ERR_BAD_SHAPE;
    const int64_t nqb = ((int64_t)seq_len + kQBlock - 1) / kQBlock, bh = (int64_t)batch * num_heads;
    if ((nqb + 16) * kKeyUnit > 0x7fffffffLL || bh > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;
    if (cus <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        cus = device_cu_count(dev);
    }
    const Plan plan = cached_plan(bh, seq_len, cus, with_workspace != 0);
    if (plan.tiles16) {
        // the 16-key-tile family: header as below with out[1] = job words per wave, out[8] = tiles on the busiest SIMD,
        // out[9] = query blocks, out[10] = 2, out[12] = 16-row halves per block, out[13] = XCD-contiguous grid, out[14] = tiles per count of a job word; then, per
        // workgroup of a (b,h) and per job list (run order), its job words as they are (cmhsa_tiles16.h)
        const cmhsa_t16::Plan16 &t = plan.t16;
        const int J = cmhsa_t16::kWordsPerWg / t.waves;
        const int64_t total = bh * t.wgs_per_bh, need16 = 16 + (int64_t)t.wgs_per_bh * cmhsa_t16::kWordsPerWg;
        if (!out || cap < need16) return (int)need16;
        const int header16[16] = {t.waves, J, t.wgs_per_bh, t.contig ? (int)((total + 7) / 8 * 8) : t.wgs_per_bh, t.contig ? 1 : (int)bh, 1, 0, 0,
                                  t.tiles_per_simd, t.blocks, 2, 0, t.sub, t.contig ? 1 : 0, t.granule, 0};
        memcpy(out, header16, sizeof(header16));
        for (int64_t i = 0; i < need16 - 16; ++i) out[16 + i] = (int)t.table[i];
        return (int)need16;
    }
    dim3 grid;
    const int grid_rc = plan_grid(plan, bh, grid);
    if (grid_rc != CMHSA_OK) return grid_rc;
    if (plan.list_p) {
        // job lists: header as below with out[2] = 0, out[8] = units of the longest list, out[10] = 3, out[12] = (b,h)s per
        // period, out[13] = lists per period (= grid y), out[14] = jobs of the longest list, out[15] = blocks of at most that many units are taken in descending order;
        // then 8 words per list as the
        // kernel reads them (16 half-word entries: units | (b,h) within the period << 8; 0 = end)
        const int64_t need_l = 16 + (int64_t)plan.list_q * 8;
        if (!out || cap < need_l) return (int)need_l;
        const int header_l[16] = {1, 2, 0, (int)grid.x, (int)grid.y, (int)grid.z, 0, 0, plan.list_units, (int)nqb, 3, 0,
                                  plan.list_p, plan.list_q, plan.list_jobs, plan.list_rev_max};
        memcpy(out, header_l, sizeof(header_l));
        for (int i = 0; i < plan.list_q * 8; ++i) out[16 + i] = (int)plan.table[i];
        return (int)need_l;
    }
    const int segs = plan.use_table ? plan.segs : 2;
    const int64_t need = 16 + (int64_t)plan.pp.wgs_per_bh * segs * 5;
    if (need > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;
    if (!out || cap < need) return (int)need;
    const int header[16] = {plan.pp.waves, segs, plan.pp.wgs_per_bh, (int)grid.x, (int)grid.y, (int)grid.z, plan.parts_per_bh,
                            plan.splits_per_bh, plan.simd_cap, (int)nqb, plan.use_table ? 1 : 0, plan.longest_first, plan.pp.chain,
                            plan.pp.n_paired, plan.pp.n_alone, (int)(plan_workspace_bytes(plan, bh) / 256)};
    memcpy(out, header, sizeof(header));
    const unsigned chain_magic = 0xffffffffu / (unsigned)plan.pp.chain;
    for (int j = 0; j < plan.pp.wgs_per_bh; ++j) {
        int *rec = out + 16 + (int64_t)j * segs * 5;
        int n[kMaxSeg], lo[kMaxSeg], cnt[kMaxSeg];
        unsigned mask[kMaxSeg], x[kMaxSeg];
        if (plan.use_table && segs == 8) {
            decode_segments<8>(plan.table + j * 16, j, (int)nqb, 0, 0, 1, 0, n, lo, cnt, mask, x);
        } else if (plan.use_table) {
            int n4[4], lo4[4], cnt4[4];
            unsigned mask4[4], x4[4];
            decode_segments<4>(plan.table + j * 8, j, (int)nqb, 0, 0, 1, 0, n4, lo4, cnt4, mask4, x4);
            for (int i = 0; i < 4; ++i) { n[i] = n4[i]; lo[i] = lo4[i]; cnt[i] = cnt4[i]; mask[i] = mask4[i]; x[i] = x4[i]; }
        } else {
            int n2[2], lo2[2], cnt2[2];
            unsigned mask2[2], x2[2];
            decode_segments<2>(nullptr, j, (int)nqb, plan.pp.n_paired, plan.pp.n_alone, plan.pp.chain, chain_magic, n2, lo2, cnt2, mask2, x2);
            for (int i = 0; i < 2; ++i) { n[i] = n2[i]; lo[i] = lo2[i]; cnt[i] = cnt2[i]; mask[i] = mask2[i]; x[i] = x2[i]; }
        }
        for (int i = 0; i < segs; ++i) { rec[5 * i] = n[i]; rec[5 * i + 1] = lo[i]; rec[5 * i + 2] = cnt[i]; rec[5 * i + 3] = (int)mask[i]; rec[5 * i + 4] = (int)x[i]; }
    }
    return (int)need;
}

int cmhsa_fp32_fwd_2d(const float *q, const float *k, const float *v, float *o,
                      int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                      int seq_len, int d_model, int num_heads, void *stream) {
    if (num_heads <= 0 || d_model < 0 || d_model % num_heads != 0) return CMHSA_ERR_BAD_SHAPE;
    return cmhsa_fp32_fwd(q, k, v, o, q_row_stride, k_row_stride, v_row_stride, o_row_stride, 0, 0, 0, 0,
                          1, seq_len, num_heads, d_model / num_heads, stream);
}

int cmhsa_fp32_fwd_packed_qkv(const float *qkv, float *o, int64_t qkv_row_stride, int64_t o_row_stride,
                              int64_t qkv_batch_stride, int64_t o_batch_stride,
                              int batch, int seq_len, int num_heads, int head_dim,
                              void *workspace, size_t workspace_bytes, void *stream) {
    if (num_heads <= 0) return CMHSA_ERR_BAD_SHAPE;
    if (head_dim != kHeadDim) return CMHSA_ERR_HEAD_DIM;
    const int64_t d_model = (int64_t)num_heads * kHeadDim;
    if (qkv_row_stride < 3 * d_model) return CMHSA_ERR_BAD_SHAPE;             // a row must hold Q | K | V
    const float *k = qkv ? qkv + d_model : nullptr, *v = qkv ? qkv + 2 * d_model : nullptr;
    return cmhsa_fp32_fwd_ws(qkv, k, v, o, qkv_row_stride, qkv_row_stride, qkv_row_stride, o_row_stride,
                             qkv_batch_stride, qkv_batch_stride, qkv_batch_stride, o_batch_stride,
                             batch, seq_len, num_heads, head_dim, workspace, workspace_bytes, stream);
}

}  // extern "C"
