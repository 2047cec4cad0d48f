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
//     a long and a short one (analytic plan, pack_plan()) or up to FOUR blocks listed in a table (choose_plan());
//     a block with n units gets ceil(n/c) waves, which split its units round-robin (split-KV inside the
//     workgroup), so even B=1, T=1024 fills the chip with one round of equally long workgroups;
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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kHeadDim = 64;            // CMHSA.cu:432
constexpr int kQBlock = 32;             // query rows per workgroup (one MFMA N-tile)
constexpr int kKeyUnit = 32;            // keys per unit (one MFMA M-tile)
// split-KV ways inside a workgroup = waves per workgroup (template parameter kWaves):
//   8 -> one workgroup per CU (two of its waves per SIMD; 224 VGPRs allow no more): shortest chain per query block
//   4 -> two workgroups per CU: prologue/merge of one overlaps the main loop of the other
constexpr int kUnitFloats = kKeyUnit * kHeadDim;          // 2048 floats = 8 KiB
constexpr int kMaxSeg = 4;                                // query blocks ("segments") one workgroup can hold (template kSegs: 2 or 4)
constexpr int kMaxTableWgs = 64;                          // workgroups per (b,h) an explicit segment table can describe
constexpr int kTableWords = 2 * kMaxSeg;                  // table words per workgroup: two per segment (see Params::seg_table)
constexpr int kMaxParts = 4;                              // parts (on different workgroups) a query block's keys may be split into
// One part's un-normalised result in the workspace: O image [8 pairs][64 lanes] float4 (the merge's native layout)
// followed by (m, l) of the 32 queries.  33 whole 256-B lines.
constexpr int kPartFloats = kUnitFloats + 64;
constexpr size_t kWsCounterBytes = 16384;                 // head of the workspace: one arrival counter per split query block
constexpr size_t kWsMaxParts = 1024;
constexpr size_t kWsMaxBytes = kWsCounterBytes + kWsMaxParts * kPartFloats * 4;       // = CMHSA_WORKSPACE_BYTES (include/cmhsa.h)
static_assert(kWsMaxBytes == CMHSA_WORKSPACE_BYTES, "include/cmhsa.h and the kernel disagree on the workspace size");
// LDS: [kWaves][8 KiB] each wave's K unit (later its merge image) | [kSegs][8 KiB] pre-scaled query blocks |
//      [kWaves][32 x (m, l)] softmax statistics for the merge | [4] "this workgroup arrived last" flags (split plans)
constexpr int lds_floats_q(int waves) { return waves * kUnitFloats; }
constexpr int lds_floats_ml(int waves, int segs) { return (waves + segs) * kUnitFloats; }
constexpr int lds_floats_flags(int waves, int segs) { return lds_floats_ml(waves, segs) + waves * 64; }
// (one-wave workgroups keep nothing but their K unit in LDS: no shared query blocks, no merge)
constexpr int lds_bytes(int waves, int segs) { return waves == 1 ? kUnitFloats * 4 : (lds_floats_flags(waves, segs) + 4) * 4; }   // <= 98 KiB (8 waves), 65 KiB (4), 8 KiB (1)
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

struct Params {
    const float *q, *k, *v;
    float *o;
    int64_t q_rs, k_rs, v_rs, o_rs;     // row strides (floats)
    int64_t q_bs, k_bs, v_bs, o_bs;     // batch strides (floats)
    int seq_len;                        // N
    int num_heads;                      // H
    int nqb;                            // ceil(N / 32)
    int grid_x;                         // gridDim.x: 8 when (b,h)s are pinned to XCDs, else bh (see the kernel)
    int bh_count;                       // B*H (one-wave workgroups, longest-first grid: gridDim.x is B*H rounded up to 8)
    unsigned heads_magic, chain_magic;  // floor((2^32-1) / num_heads), floor((2^32-1) / chain): udiv_magic()
    // packing of query blocks into workgroups (pack_plan()):
    int chain;                          // c: max units any wave processes (>= ceil(nqb / kWaves))
    int n_paired;                       // R: query blocks 1..R units long are paired n <-> R+1-n
    int n_alone;                        // nqb - R: the longest query blocks get a workgroup each
    // explicit plan (use_table != 0; choose_plan()): workgroup j of every (b,h) holds up to four segments, two table
    // words each (seg_table[j * 8 + 2 i], [.. + 1]; a segment with n = 0 is absent, present ones come first):
    //   word 0: n (units of the query block = block index + 1) | first unit << 8 | unit count << 16 | wave mask << 24
    //           -- the waves of the mask take the units [first, first + count) round-robin; count < n = the block's keys
    //           are split across workgroups and this segment is one PART of it
    //   word 1: 0 for a whole block, else  first part slot of the block | part index << 8 | parts << 12 | split id << 16
    int use_table;
    // cross-workgroup split (table plans with parts only; null otherwise): a part's workgroup stores its un-normalised
    // (O, m, l) to ws_parts[(bh * parts_per_bh + slot)] and counts itself in ws_counters[bh * splits_per_bh + split id];
    // the workgroup that arrives last combines the parts in part order and writes O
    float *ws_parts;
    unsigned *ws_counters;
    int parts_per_bh, splits_per_bh;
    unsigned seg_table[kMaxTableWgs * kTableWords];
};

// Packing plan.  A query block with n units gets g(n) = ceil(n / c) waves, so that no wave runs more than c units
// (c >= ceil(nqb / kWaves), so that the longest block fits in one workgroup).  The R = min(nqb, c*(kWaves-1))
// shortest blocks are paired n <-> R+1-n: g(n) + g(R+1-n) <= kWaves (== for R = c*(kWaves-1)), i.e. a paired
// workgroup has R+1 units in total; the nqb-R longest blocks each need all kWaves waves.  One round of
// equally long workgroups instead of a tail of short ones that each hold a whole CU.
struct PackPlan { int waves, chain, n_paired, n_alone, wgs_per_bh; };
inline PackPlan pack_plan(int nqb, int waves, int chain = 0) {
    PackPlan pl;
    pl.waves = waves;
    pl.chain = chain > 0 ? chain : (nqb + waves - 1) / waves;
    // (one-wave workgroups: every block is paired, n <-> nqb+1-n, and a wave runs its two blocks one after the other)
    pl.n_paired = waves == 1 ? nqb : (int)min((int64_t)nqb, (int64_t)pl.chain * (waves - 1));
    pl.n_alone = nqb - pl.n_paired;
    pl.wgs_per_bh = pl.n_alone + (pl.n_paired + 1) / 2;
    return pl;
}

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

// one float4 of the output.  Plain stores: write-through (sc1) stores of O measured +10 us on BASELINE config 2,
// non-temporal ones +0.45 us (tools/ab_run.sh).
__device__ __forceinline__ void store_o(float *ptr, f32x4 v, bool in_range) {
    if (in_range) *reinterpret_cast<f32x4 *>(ptr) = v;
}

// floor(n / d) for wave-uniform n, d >= 1 with magic = floor((2^32 - 1) / d): the estimate mulhi(n, magic) is
// the quotient or one below it for every 32-bit n, so one correction step makes it exact.  A handful of
// scalar instructions instead of the ~30-instruction float-reciprocal sequence hipcc emits for `/`.
__host__ __device__ __forceinline__ int udiv_magic(int n, int d, unsigned magic) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned q = __umulhi((unsigned)n, magic);
#else
    const unsigned q = (unsigned)(((uint64_t)(unsigned)n * magic) >> 32);
#endif
    return (int)(q + (((unsigned)n - q * (unsigned)d) >= (unsigned)d ? 1u : 0u));
}

// Workgroup j of a (b,h) -> its segments.  ONE definition for the kernel and for the host (cmhsa_plan_dump(), which the
// CPU tests use to check that every plan covers every unit of every query block exactly once).
//   kSegs > 2: explicit plan, `tw` = the workgroup's eight table words (Params::seg_table);
//   else the analytic plan: the n_alone longest blocks alone, the others paired n <-> n_paired+1-n (the middle block of an
//   odd count alone); a block with n units gets ceil(n / chain) consecutive waves.
template <int kSegs>
__host__ __device__ __forceinline__ void decode_segments(const unsigned *tw, int j, int nqb, int n_paired, int n_alone, int chain,
                                                         unsigned chain_magic, int (&seg_n)[kSegs], int (&seg_ulo)[kSegs],
                                                         int (&seg_cnt)[kSegs], unsigned (&seg_mask)[kSegs], unsigned (&seg_x)[kSegs]) {
    if (kSegs > 2) {
#pragma unroll
        for (int i = 0; i < kSegs; ++i) {
            const unsigned w0 = tw[2 * i];
            seg_n[i] = w0 & 255;
            seg_ulo[i] = (w0 >> 8) & 255;
            seg_cnt[i] = (w0 >> 16) & 255;
            seg_mask[i] = w0 >> 24;
            seg_x[i] = tw[2 * i + 1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < kSegs; ++i) seg_n[i] = 0;
        if (j < n_alone) {
            seg_n[0] = nqb - j;
        } else {
            const int t = j - n_alone;
            seg_n[0] = n_paired - t;
            if (t + 1 != seg_n[0]) seg_n[1] = t + 1;   // (middle block of an odd count: alone with half the waves)
        }
        int w0 = 0;
#pragma unroll
        for (int i = 0; i < kSegs; ++i) {
            const int g = seg_n[i] ? udiv_magic(seg_n[i] + chain - 1, chain, chain_magic) : 0;
            seg_mask[i] = ((1u << g) - 1u) << w0;
            seg_ulo[i] = 0;
            seg_cnt[i] = seg_n[i];
            seg_x[i] = 0;
            w0 += g;
        }
    }
}

// kSegs = 2: the analytic plan (one block or a pair per workgroup); 4: an explicit table (Params::seg_table).
// kAcquire: the last-arriving workgroup of a split query block runs an agent-scope acquire before it reads the other
// parts (workspace in a __device__ array: the no-allocation entry point); without it every load of a part is an sc1 load
// of sc1-stored, drained bytes in hipMalloc memory, one workgroup per CU (MI355X guide, visibility table row 1).
// kWaves = 1: a workgroup is ONE wave that runs its (up to two) query blocks one after the other, whole -- no split-KV,
// no shared LDS, no barrier, O straight from the accumulators.  For grids of many rounds: every wave slot of the chip is
// refilled on its own, and with the pairing n <-> nqb+1-n every wave runs the same nqb+1 units.
template <int kWaves, int kSegs, bool kAcquire>
__global__ __launch_bounds__(kWaves * 64, 2) void cmhsa_fwd_kernel(Params p) {
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
    unsigned tw[kTableWords];
    if constexpr (kSegs > 2) {
#pragma unroll
        for (int i = 0; i < kTableWords; ++i) tw[i] = p.seg_table[blockIdx.y * kTableWords + i];
        asm volatile("" ::"s"(tw[0]), "s"(tw[1]), "s"(tw[2]), "s"(tw[3]), "s"(tw[4]), "s"(tw[5]), "s"(tw[6]), "s"(tw[7]),
                     "s"(p.q), "s"(p.k), "s"(p.v), "s"(p.o), "s"(p.q_rs), "s"(p.k_rs), "s"(p.v_rs), "s"(p.o_rs));
    } else
    asm volatile("" ::"s"(p.q), "s"(p.k), "s"(p.v), "s"(p.o), "s"(p.q_rs), "s"(p.k_rs), "s"(p.v_rs), "s"(p.o_rs));
    asm volatile("" ::"s"(p.q_bs), "s"(p.k_bs), "s"(p.v_bs), "s"(p.o_bs), "s"(p.seq_len), "s"(p.num_heads), "s"(p.nqb),
                 "s"(p.grid_x), "s"(p.heads_magic), "s"(p.chain_magic), "s"(p.chain), "s"(p.n_paired), "s"(p.n_alone), "s"(p.use_table));
    STAMP(124);

    // ---- grid -> (batch*head, workgroup j of that head).  Workgroups are dispatched in linear order, x fastest,
    //      round-robin over the 8 XCDs: with grid = (8, wgs_per_bh, bh/8) every XCD keeps its own (b,h)s (K/V
    //      stay in that XCD's L2), with grid = (bh, wgs_per_bh, 1) otherwise; either way j = 0 (heaviest) first.
    const int bh = blockIdx.z * p.grid_x + blockIdx.x;
    const int j = blockIdx.y;
    if constexpr (kWaves == 1) {
        if (bh >= p.bh_count) return;            // padding of the longest-first grid (no barrier in this variant)
    }
    // ---- workgroup j -> up to kSegs query blocks ("segments"), this wave -> (segment, local wave index) ----
    int seg_n[kSegs];                           // units (= query block index + 1) of each segment's query block; 0 = none
    int seg_ulo[kSegs], seg_cnt[kSegs];          // the segment covers units [ulo, ulo + cnt) of it (the whole block: 0, n)
    unsigned seg_mask[kSegs], seg_x[kSegs];      // waves working on it (bit w = wave w); table word 1 (0 = not a part)
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

    const float *qg = p.q + b * p.q_bs + (int64_t)h * kHeadDim;
    const float *kg = p.k + b * p.k_bs + (int64_t)h * kHeadDim;
    const float *vg = p.v + b * p.v_bs + (int64_t)h * kHeadDim;

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
            const int rg = wave + i * kWaves, sg = rg >> 3, row = 4 * (rg & 7) + sub;    // sg is a compile-time constant
            const int sq0 = (max(seg_n[sg], 1) - 1) * kQBlock;
            const __amdgpu_buffer_rsrc_t qsrc = unit_rsrc(qg, p.q_rs, sq0, N, seg_n[sg] != 0);
            qst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qsrc, (row * (int)p.q_rs + chunk * 4) * 4, 0, 0));
        }
        STAMP(121);
#pragma unroll
        for (int i = 0; i < kQIters; ++i) {
            const int rg = wave + i * kWaves, sg = rg >> 3, row = 4 * (rg & 7) + sub;
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
            load_unit(unit_rsrc(kg, p.k_rs, u_first * kKeyUnit, N), koff, kstep, kst);
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
        auto ahead = [&](const float *base, int64_t rs, int u, int d) {
            int t = u + d * seg_waves;
            bool en = t < u_end;
            if constexpr (kWaves == 1) {
                if (!en) { t -= u_end; en = t < n_after; }
            }
            return unit_rsrc(base, rs, t * kKeyUnit, N, en);
        };

        // One unit = QK chain, softmax, PV chain.  `first` (compile time): the wave's first unit, whose QK chain
        // also carries the loads the steady state issues one unit ahead -- V(u) and K(u + seg_waves) -- so that
        // the first MFMA does not queue behind 24 more buffer loads per wave (a CU's L1 serves 64 B/clk).
        auto unit = [&](const int u, auto first) __attribute__((always_inline)) {
            constexpr bool kFirst = decltype(first)::value;
#ifdef CMHSA_STAMPS
            STAMP(stamp_i);
#endif
            // ---- S^T[key][query] = sum_d K[key][d] * Q[query][d] : 32 chained MFMAs.  K fragments are read
            //      two steps (512 MFMA cycles) ahead. ----
            f32x16 s = {0};
            __amdgpu_buffer_rsrc_t vsrc0, ksrc1;
            if constexpr (kFirst) {
                vsrc0 = unit_rsrc(vg, p.v_rs, u * kKeyUnit, N);
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
            // ---- causal mask: only the diagonal unit has key > query pairs (CMHSA.cu:228, :249) ----
            if (u == qb) {
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
            if constexpr (!kFirst) {                  // (the first unit goes straight to the max path below)
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
            if (kFirst || !__all(psum < kRescaleSum)) {          // NaN and inf compare false: they take this path too
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                mx = xhalf_max(mx);
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // 0 on the first unit (m_run = -inf)
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
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
            const __amdgpu_buffer_rsrc_t vsrc1 = ahead(vg, p.v_rs, u, 1);
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
        unit(u_first, std::true_type{});         // every active wave owns at least one unit (a segment has no more waves than units)
        if constexpr (kWaves > 1) {
            for (int u = u_first + seg_waves; u < u_end; u += seg_waves) unit(u, std::false_type{});
            l_run = xhalf_sum(l_run);
        } else {
            float *__restrict__ og = p.o + b * p.o_bs + (int64_t)h * kHeadDim;
            int u = 1;
            for (;;) {
                for (; u < u_end; ++u) unit(u, std::false_type{});
                // the whole block was this wave's: normalise and store, lane = query, 4 consecutive d per store (the
                // merge's pairs).  The next block's first K unit is in LDS and its first V unit in flight already; its
                // query fragments are requested BEFORE the stores (the registers of this block's are free now)
                const int n_next = n_after;
                f32x4 qt[8];
                if (n_next) {
                    const __amdgpu_buffer_rsrc_t qsrc = unit_rsrc(qg, p.q_rs, (n_next - 1) * kQBlock, N);
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
                if (!n_next) break;
                // fresh accumulators; m_run = -inf sends the block's first unit down the max path (its row sums are not finite)
#pragma unroll
                for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
                m_run = -INFINITY;
                l_run = 0.f;
                u_end = n_next; qb = n_next - 1; n_after = 0; u = 0;
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) qf[4 * c + j] = qt[c][j] * kScaleLog2;
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
        float *__restrict__ og = p.o + b * p.o_bs + (int64_t)h * kHeadDim;
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
                    const unsigned x = t == 0 ? seg_x[0] : t == 1 ? seg_x[1] : t == 2 ? seg_x[2] : seg_x[3];
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

std::atomic<int> g_attr_set[6][64];  // per variant and device: dynamic-LDS limit raised once, not per call

template <int kWaves, int kSegs, bool kAcquire>
hipError_t launch(const Params &p, dim3 grid, void *stream) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<int> *flag = (dev >= 0 && dev < 64) ? &g_attr_set[kWaves == 1 ? 5 : kAcquire ? 4 : (kWaves == 8) * 2 + (kSegs > 2)][dev] : nullptr;
    if (!flag || !flag->load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(cmhsa_fwd_kernel<kWaves, kSegs, kAcquire>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(kWaves, kSegs));
        if (e != hipSuccess) return e;
        if (flag) flag->store(1, std::memory_order_release);
    }
    // hipLaunchKernel returns this launch's own status: an error left behind by an unrelated earlier HIP call on
    // this thread is neither reported as ours nor cleared (hipGetLastError would do both)
    void *args[] = {const_cast<Params *>(&p)};
    return hipLaunchKernel(reinterpret_cast<const void *>(cmhsa_fwd_kernel<kWaves, kSegs, kAcquire>), grid, dim3(kWaves * 64), args,
                           lds_bytes(kWaves, kSegs), static_cast<hipStream_t>(stream));
}

// ---- plan selection (host) ---------------------------------------------------------------------------------------
// A plan says which query blocks (or parts of them) each workgroup of a (b,h) holds and which of its waves work on
// each.  Three families:
//   analytic  pack_plan(): a block alone or a pair n <-> R+1-n; decoded in the kernel from three integers; any size;
//   chain     table plan: every block gets ceil(n / c) consecutive waves, blocks packed greedily (longest first into
//             the feasible workgroup with the fewest units), up to 4 per workgroup;
//   packed    table plan from pack_blocks(): a cap L on the units any SIMD runs (waves w and w+4 of an 8-wave workgroup
//             share one), blocks placed wave by wave under that cap; with a workspace, a block may be cut into two
//             PARTS held by different workgroups (its keys split in two ranges, merged through the workspace by the
//             workgroup that finishes last) -- the only way to put fewer units on a SIMD than the longest block has
//             per SIMD of one CU, and to use every CU when the blocks do not divide evenly.
// Table plans describe at most 64 workgroups per (b,h) and blocks of at most 255 units.
// Grids that fit the chip in ONE round are where the choice matters (measured, tools/sweep_plans.sh): among the
// single-round candidates the cheapest by
//     cycles = kFixed + (units on the busiest SIMD) * kUnit  [+ kHandoff in a workgroup that holds a part]
// wins; an 8-wave workgroup owns its CU (two of its waves per SIMD), two 4-wave workgroups share one.  Grids of
// several rounds run within +-3 % of each other under every plan tried, so they keep the analytic plan with 4-wave
// workgroups at the smallest c (two co-resident workgroups hide each other's prologue and merge).
constexpr double kFixedCycles = 8000.0, kUnitCycles = 5000.0, kHandoffCycles = 5000.0;

struct Plan {
    PackPlan pp;                                 // waves, and for analytic / chain plans the chain cap
    bool use_table = false;
    int n_wgs = 0;                               // workgroups per (b,h)
    int parts_per_bh = 0, splits_per_bh = 0;     // > 0: some blocks are split across workgroups (needs a workspace)
    int simd_cap = 0;                            // packed plans: L
    bool single_round = false;                   // the whole grid is resident at once
    int longest_first = 0;                       // G > 0: one-wave workgroups, one block each; (b,h)s in groups of G (a multiple of 8), within a group longest blocks first
    double cycles = 0.0;
    const char *family = "analytic";
    unsigned table[kMaxTableWgs * kTableWords];  // table plans: Params::seg_table
};

inline unsigned seg_word0(int n, int lo, int cnt, unsigned mask) { return (unsigned)n | ((unsigned)lo << 8) | ((unsigned)cnt << 16) | (mask << 24); }

// units on the busiest SIMD of a workgroup holding blocks n[0..k): block i gets ceil(n_i/c) consecutive waves, its
// units go round-robin over them; waves w and w+4 share a SIMD
int busiest_simd(const int *n, int k, int c, int waves) {
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, w = 0;
    for (int i = 0; i < k; ++i) {
        if (!n[i]) continue;
        const int g = (n[i] + c - 1) / c;
        for (int t = 0; t < g && w < 8; ++t, ++w) load[w] = n[i] / g + (t < n[i] % g ? 1 : 0);
    }
    int m = 0;
    for (int sidx = 0; sidx < 4; ++sidx) m = max(m, load[sidx] + (waves == 8 ? load[sidx + 4] : 0));
    return m;
}

void price(Plan &pl, const int (*wg_blocks)[kMaxSeg], int n_wgs, int64_t bh, int cus) {
    const int waves = pl.pp.waves;
    const int64_t wgs = bh * n_wgs;
    const bool shared = waves == 4 && wgs > cus;                    // a second 4-wave workgroup on the same CU
    double longest = 0.0;
    for (int j = 0; j < n_wgs; ++j) {
        const int l = busiest_simd(wg_blocks[j], kMaxSeg, pl.pp.chain, waves);
        longest = fmax(longest, kFixedCycles + (shared ? 2.0 : 1.0) * l * kUnitCycles);
    }
    pl.n_wgs = n_wgs;
    pl.single_round = wgs <= (int64_t)cus * (waves == 8 ? 1 : 2);
    pl.cycles = longest;                 // meaningful for single-round grids only, the only ones compared
}

bool chain_table_plan(Plan &pl, int nqb, int waves, int c, int64_t bh, int cus) {
    if (nqb > 255) return false;
    int blocks[kMaxTableWgs][kMaxSeg] = {}, units[kMaxTableWgs] = {}, free_waves[kMaxTableWgs], nseg[kMaxTableWgs] = {};
    int n_wgs = 0;
    for (int n = nqb; n >= 1; --n) {
        const int g = (n + c - 1) / c;
        int pick = -1;
        for (int j = 0; j < n_wgs; ++j)
            if (free_waves[j] >= g && nseg[j] < kMaxSeg && (pick < 0 || units[j] < units[pick])) pick = j;
        if (pick < 0) {
            if (n_wgs == kMaxTableWgs) return false;
            pick = n_wgs++;
            free_waves[pick] = waves;
        }
        blocks[pick][nseg[pick]++] = n;
        units[pick] += n;
        free_waves[pick] -= g;
    }
    // heaviest workgroups first (blockIdx.y = 0 is dispatched first): selection sort, n_wgs <= 64
    for (int i = 0; i < n_wgs; ++i) {
        int m = i;
        for (int j = i + 1; j < n_wgs; ++j) if (units[j] > units[m]) m = j;
        if (m != i) {
            for (int t = 0; t < kMaxSeg; ++t) { const int x = blocks[i][t]; blocks[i][t] = blocks[m][t]; blocks[m][t] = x; }
            const int x = units[i]; units[i] = units[m]; units[m] = x;
        }
    }
    pl.pp = pack_plan(nqb, waves, c);
    pl.pp.wgs_per_bh = n_wgs;
    pl.use_table = true;
    pl.family = "chain table";
    for (int j = 0; j < n_wgs; ++j) {
        int w0 = 0;
        for (int t = 0; t < kMaxSeg; ++t) {
            const int n = blocks[j][t], g = n ? (n + c - 1) / c : 0;
            pl.table[j * kTableWords + 2 * t] = n ? seg_word0(n, 0, n, ((1u << g) - 1u) << w0) : 0u;
            pl.table[j * kTableWords + 2 * t + 1] = 0u;
            w0 += g;
        }
    }
    price(pl, blocks, n_wgs, bh, cus);
    return true;
}

void analytic_plan(Plan &pl, int nqb, int waves, int c, int64_t bh, int cus) {
    pl.pp = pack_plan(nqb, waves, c);
    pl.use_table = false;
    pl.family = "analytic";
    // priced from its closed form: the alone blocks nqb, nqb-1, ..., then the pairs (R - t, t + 1)
    std::vector<int> flat((size_t)pl.pp.wgs_per_bh * kMaxSeg, 0);
    int (*blocks)[kMaxSeg] = reinterpret_cast<int (*)[kMaxSeg]>(flat.data());
    for (int j = 0; j < pl.pp.wgs_per_bh; ++j) {
        if (j < pl.pp.n_alone) blocks[j][0] = nqb - j;
        else { const int t = j - pl.pp.n_alone; blocks[j][0] = pl.pp.n_paired - t; if (t + 1 != blocks[j][0]) blocks[j][1] = t + 1; }
    }
    price(pl, blocks, pl.pp.wgs_per_bh, bh, cus);
}

// Experiments only (tests/helpers/libcmhsa_hip_planenv.so and tools/ build the library with -DCMHSA_PLAN_ENV):
// CMHSA_WAVES=4|8, CMHSA_CHAIN=c, CMHSA_TABLE=0|1 force an analytic / chain plan, CMHSA_WAVES=1 one-wave workgroups
// (CMHSA_NOWP=1: never those; CMHSA_LPT=0: pairs of blocks only, CMHSA_LPT=G: one block per wave, longest first in groups
// of G (b,h)s), CMHSA_PACKED=L forces a packed plan
// with SIMD cap L (CMHSA_SPLIT=0|1: without / with blocks split across workgroups), CMHSA_VERBOSE prints the choice.
// The shipped library reads no environment variable.
#ifdef CMHSA_PLAN_ENV
int env_int(const char *name, int fallback) { const char *e = getenv(name); return e ? atoi(e) : fallback; }
#else
constexpr int env_int(const char *, int fallback) { return fallback; }
#endif

// ---- packed plans: 8-wave workgroups, one per CU -------------------------------------------------------------------
// Work items = whole query blocks, or the two parts of a block cut at a unit boundary.  Items go, longest first (with a
// little seeded jitter, so that a failed attempt is followed by a different one), into the fullest workgroup that
// takes them, else into a new one.  "Takes" = there is a set of g free wave slots such that, with the item's units
// dealt round-robin to those waves in ascending wave order (what the kernel does), no SIMD (slots s and s+4) runs more
// than `cap` units: cap = L, or L-1 in a workgroup that holds a part -- the hand-off through the workspace costs about
// a unit of time.  g runs from ceil(cnt / ceil(cap/2)) (prefer two waves per SIMD over one long chain) upwards.
struct PackItem { int n, lo, cnt, split; double key; };          // split: -1 = whole block, else index of the split block
struct PackWg {
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned used = 0;                           // wave slots taken
    int nseg = 0, total = 0;
    bool partial = false;
    int seg_n[kMaxSeg], seg_lo[kMaxSeg], seg_cnt[kMaxSeg], seg_split[kMaxSeg];
    unsigned seg_mask[kMaxSeg];
};

struct Rng {                                     // xorshift64*: deterministic, seeded per shape
    uint64_t s;
    double next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return (double)((s * 2685821657736338717ull) >> 11) / 9007199254740992.0; }
};

// best set of g free slots of `wg` for an item of cnt units under `cap`; 0 if none
unsigned best_slots(const PackWg &wg, int cnt, int g, int cap) {
    unsigned best = 0;
    int best_score = 1 << 30;
    const unsigned free_mask = ~wg.used & 0xffu;
    for (unsigned m = 1; m < 256; ++m) {
        if ((m & free_mask) != m || __builtin_popcount(m) != g) continue;
        int load[8], i = 0, score = 0;
        bool ok = true;
        for (int w = 0; w < 8; ++w) {
            load[w] = wg.load[w];
            if ((m >> w) & 1u) { load[w] = cnt / g + (i < cnt % g ? 1 : 0); ++i; }
        }
        for (int sidx = 0; sidx < 4 && ok; ++sidx) {
            const int sum = load[sidx] + load[sidx + 4];
            if (sum > cap) ok = false;
            const bool touched = ((m >> sidx) | (m >> (sidx + 4))) & 1u;
            const bool full_pair = (((wg.used | m) >> sidx) & 1u) && (((wg.used | m) >> (sidx + 4)) & 1u);
            if (touched) score += full_pair ? (cap - sum) : (cap - sum) + cap;     // prefer completing a SIMD tightly
        }
        if (ok && score < best_score) { best_score = score; best = m; }
    }
    return best;
}

// wave_share: preferred units per wave as a fraction of the SIMD cap: 1/2 = two equal waves per SIMD; up to 1 = one
// long chain per SIMD (fewer waves per block, for plans with few workgroups per (b,h), where wave slots are scarce)
bool pack_attempt(int nqb, int L, bool allow_split, double psplit, double wave_share, Rng &rng, int g_max, std::vector<PackWg> &wgs, int &n_splits) {
    const int Lp = L - 1;
    std::vector<PackItem> items;
    n_splits = 0;
    for (int n = nqb; n >= 1; --n) {
        const bool must = n > 4 * L;
        if (must && (!allow_split || Lp < 1)) return false;
        if (must || (allow_split && Lp >= 1 && n > 8 && rng.next() < psplit)) {
            const int lo_cut = max(1, n - 4 * Lp), hi_cut = min(n - 1, 4 * Lp);
            if (lo_cut > hi_cut) return false;                       // would need more than two parts
            const int cut = lo_cut + (int)(rng.next() * (hi_cut - lo_cut + 1));
            items.push_back(PackItem{n, 0, cut, n_splits, 0.0});
            items.push_back(PackItem{n, cut, n - cut, n_splits, 0.0});
            ++n_splits;
        } else {
            items.push_back(PackItem{n, 0, n, -1, 0.0});
        }
    }
    if (n_splits > 127) return false;
    for (PackItem &it : items) it.key = -(double)it.cnt + 3.0 * rng.next();
    std::stable_sort(items.begin(), items.end(), [](const PackItem &a, const PackItem &b) { return a.key < b.key; });
    wgs.clear();
    for (const PackItem &it : items) {
        const bool part = it.split >= 0;
        // candidates: existing workgroups, fullest first, then a new one
        std::vector<int> order(wgs.size());
        for (size_t j = 0; j < order.size(); ++j) order[j] = (int)j;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wgs[a].total > wgs[b].total; });
        order.push_back(-1);
        bool placed = false;
        for (int j : order) {
            PackWg fresh;
            PackWg &wg = j < 0 ? fresh : wgs[j];
            if (wg.nseg >= kMaxSeg) continue;
            const int cap = (wg.partial || part) ? Lp : L;
            if (part && !wg.partial) {
                bool fits = true;
                for (int sidx = 0; sidx < 4; ++sidx) fits &= wg.load[sidx] + wg.load[sidx + 4] <= Lp;
                if (!fits) continue;
            }
            bool same_block = false;
            for (int t = 0; t < wg.nseg; ++t) same_block |= wg.seg_n[t] == it.n;
            if (same_block) continue;                                   // both parts of a block in one workgroup: pointless
            // wave counts to try: from two waves per SIMD's worth upwards, then -- longer chains, one wave on a SIMD --
            // downwards to the fewest waves the cap allows
            const int wcap = max(1, min(cap, (int)ceil(cap * wave_share))), free_slots = 8 - __builtin_popcount(wg.used);
            const int g_pref = (it.cnt + wcap - 1) / wcap, g_abs = (it.cnt + cap - 1) / cap, g_hi = min(free_slots, it.cnt);
            for (int step = 0; step < 16; ++step) {
                const int g = step < 8 ? g_pref + step : g_pref - 1 - (step - 8);
                if (g > g_hi || g < g_abs || g < 1) continue;
                const unsigned m = best_slots(wg, it.cnt, g, cap);
                if (!m) continue;
                int i = 0;
                for (int w = 0; w < 8; ++w)
                    if ((m >> w) & 1u) { wg.load[w] = it.cnt / g + (i < it.cnt % g ? 1 : 0); ++i; }
                wg.used |= m;
                wg.seg_n[wg.nseg] = it.n; wg.seg_lo[wg.nseg] = it.lo; wg.seg_cnt[wg.nseg] = it.cnt;
                wg.seg_split[wg.nseg] = it.split; wg.seg_mask[wg.nseg] = m;
                ++wg.nseg;
                wg.total += it.cnt;
                wg.partial |= part;
                placed = true;
                break;
            }
            if (placed) {
                if (j < 0) {
                    if ((int)wgs.size() == g_max) {
#ifdef CMHSA_PLAN_ENV
                        if (getenv("CMHSA_VERBOSE") && atoi(getenv("CMHSA_VERBOSE")) > 1)
                            fprintf(stderr, "pack_attempt L=%d: block %d [%d,+%d) would open workgroup %d of %d\n", L, it.n, it.lo, it.cnt, g_max + 1, g_max);
#endif
                        return false;
                    }
                    wgs.push_back(fresh);
                }
                break;
            }
        }
        if (!placed) {
#ifdef CMHSA_PLAN_ENV
            if (getenv("CMHSA_VERBOSE") && atoi(getenv("CMHSA_VERBOSE")) > 1)
                fprintf(stderr, "pack_attempt L=%d: no place for block %d units [%d,+%d) with %zu workgroups open\n", L, it.n, it.lo, it.cnt, wgs.size());
#endif
            return false;
        }
    }
    return true;
}

bool packed_plan(Plan &pl, int nqb, int L, bool allow_split, int64_t bh, int cus, bool eager_split = false) {
    if (nqb > 255 || bh > cus) return false;
    const int g_max = (int)min((int64_t)kMaxTableWgs, (int64_t)cus / bh);
    Rng rng{0x9E3779B97F4A7C15ull ^ ((uint64_t)nqb << 32) ^ ((uint64_t)L << 16) ^ (uint64_t)g_max ^ (allow_split ? 0x5555ull : 0ull)
            ^ ((uint64_t)env_int("CMHSA_SEED", 0) << 40)};
    // Up to four packings that fit are priced and the cheapest is kept (the first on ties): a workgroup's time =
    // kFixed + its busiest SIMD's units * kUnit + kHandoff for the first PART it holds, 0.3 kHandoff for each further one
    // -- three parts in one full workgroup measured +6 % on BASELINE config 2, two parts +1 % (tools/seed_sweep.sh).
    std::vector<PackWg> wgs, cand;
    int n_splits = 0, cand_splits = 0, found = 0;
    double best_max = 0.0, best_sum = 0.0;
    auto price_wgs = [](const std::vector<PackWg> &v, double &worst, double &sum) {
        worst = sum = 0.0;
        for (const PackWg &wg : v) {
            int busiest = 0, parts = 0;
            for (int sidx = 0; sidx < 4; ++sidx) busiest = max(busiest, wg.load[sidx] + wg.load[sidx + 4]);
            for (int t = 0; t < wg.nseg; ++t) parts += wg.seg_split[t] >= 0;
            const double c = kFixedCycles + busiest * kUnitCycles + (parts ? kHandoffCycles * (0.7 + 0.3 * parts) : 0.0);
            worst = fmax(worst, c);
            sum += c;
        }
    };
    // whole blocks first (and parts only where a block cannot fit); then a little optional splitting
    const int tries = allow_split ? 24 : 9;
    for (int t = 0; t < tries && found < 4; ++t) {
        // (eager_split: experiments / tests only -- split blocks although they would fit whole)
        const double psplit = !allow_split ? 0.0 : eager_split ? 0.2 : t < 6 ? 0.0 : 0.04 * (1 + (t - 6) / 6);
        // two equal waves per SIMD first (measured best on BASELINE config 2); longer chains only when that cannot be packed
        const double wave_share = allow_split ? (t < 12 ? 0.5 : t % 2 ? 0.75 : 1.0) : (t < 3 ? 0.5 : t < 6 ? 0.75 : 1.0);
        if (!pack_attempt(nqb, L, allow_split, psplit, wave_share, rng, g_max, cand, cand_splits)) continue;
        if (bh * cand_splits * 2 > (int64_t)kWsMaxParts || bh * cand_splits * 4 > (int64_t)kWsCounterBytes) continue;
        double worst, sum;
        price_wgs(cand, worst, sum);
        if (!found || worst < best_max - 1.0 || (worst < best_max + 1.0 && sum < best_sum - 1.0)) {
            wgs = cand; n_splits = cand_splits; best_max = worst; best_sum = sum;
        }
        ++found;
    }
    if (!found) return false;
    std::stable_sort(wgs.begin(), wgs.end(), [](const PackWg &a, const PackWg &b) { return a.total > b.total; });   // heaviest first
    pl.pp = pack_plan(nqb, 8, (L + 1) / 2);
    pl.pp.wgs_per_bh = (int)wgs.size();
    pl.use_table = true;
    pl.family = n_splits ? "packed table, blocks split across workgroups" : "packed table";
    pl.n_wgs = (int)wgs.size();
    pl.simd_cap = L;
    pl.splits_per_bh = n_splits;
    pl.parts_per_bh = 2 * n_splits;
    pl.single_round = true;
    for (int j = 0; j < pl.n_wgs; ++j) {
        const PackWg &wg = wgs[j];
        for (int t = 0; t < kMaxSeg; ++t) {
            unsigned w0 = 0, w1 = 0;
            if (t < wg.nseg) {
                w0 = seg_word0(wg.seg_n[t], wg.seg_lo[t], wg.seg_cnt[t], wg.seg_mask[t]);
                if (wg.seg_split[t] >= 0)        // first slot | part index << 8 | parts << 12 | split id << 16
                    w1 = (unsigned)(2 * wg.seg_split[t]) | ((wg.seg_lo[t] ? 1u : 0u) << 8) | (2u << 12) | ((unsigned)wg.seg_split[t] << 16);
            }
            pl.table[j * kTableWords + 2 * t] = w0;
            pl.table[j * kTableWords + 2 * t + 1] = w1;
        }
    }
    pl.cycles = best_max;
    return true;
}

int device_cu_count(int dev) {
    static std::atomic<int> cached[64];
    const bool slot = dev >= 0 && dev < 64;
    int n = slot ? cached[dev].load(std::memory_order_relaxed) : 0;
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (slot) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

Plan choose_plan(int64_t bh, int nqb, int cus, bool allow_split) {
    static const int forced_waves = env_int("CMHSA_WAVES", 0);
    static const int forced_chain = env_int("CMHSA_CHAIN", 0);
    static const int forced_table = env_int("CMHSA_TABLE", -1);
    static const int forced_packed = env_int("CMHSA_PACKED", 0);
    static const int forced_split = env_int("CMHSA_SPLIT", -1);
    Plan best;
    bool have = false;
    const bool forced_old = (forced_waves != 0) | (forced_chain != 0) | (forced_table >= 0);
    for (int waves = 8; waves >= 4 && !forced_packed; waves -= 4) {
        if (forced_waves && forced_waves != waves) continue;
        const int c_min = (nqb + waves - 1) / waves;
        // nothing changes any more once every block fits in one wave; nqb > 4096: only the smallest c (pricing is O(nqb))
        const int c_max = forced_chain ? max(forced_chain, c_min) : (nqb > 4096 ? c_min : min(nqb, 4 * c_min + 4));
        for (int c = forced_chain ? c_max : c_min; c <= c_max; ++c) {
            for (int table = 0; table <= 1; ++table) {
                if (forced_table >= 0 && forced_table != table) continue;
                // only single-round grids are compared: skip the others before the O(nqb) pricing
                if (!table && !forced_old && bh * pack_plan(nqb, waves, c).wgs_per_bh > (int64_t)cus * (waves == 8 ? 1 : 2)) continue;
                Plan pl;
                if (table) { if (!chain_table_plan(pl, nqb, waves, c, bh, cus)) continue; }
                else analytic_plan(pl, nqb, waves, c, bh, cus);
                if (!pl.single_round && !forced_old) continue;
                // ties: keep the earlier candidate (8 waves before 4, small c before large, analytic before table)
                if (!have || pl.cycles < best.cycles * 0.98) { best = pl; have = true; }
            }
        }
    }
    // packed plans: the smallest SIMD caps that could hold the work, without and (given a workspace) with split blocks
    if (!forced_old && nqb <= 255 && bh <= cus) {
        const int g_max = (int)min((int64_t)kMaxTableWgs, (int64_t)cus / bh);
        const int64_t total = (int64_t)nqb * (nqb + 1) / 2;
        const int l_min = forced_packed ? forced_packed : (int)max((int64_t)1, (total + 4 * g_max - 1) / (4 * g_max));
        const int l_max = forced_packed ? forced_packed : l_min + 3;
        for (int L = l_min; L <= l_max; ++L) {
            for (int split = 0; split <= (allow_split ? 1 : 0); ++split) {
                if (forced_split >= 0 && forced_split != split && (allow_split || forced_split == 0)) continue;
                Plan pl;
                if (!packed_plan(pl, nqb, L, split != 0, bh, cus, split != 0 && forced_split == 1)) continue;
                if (split && !pl.splits_per_bh) continue;            // the same plan as split = 0
                if (!have || pl.cycles < best.cycles * 0.98) { best = pl; have = true; }
            }
        }
    }
    // Grids of several rounds.  Two families, each priced from a fit to measured launch times (cycles at 2.4 GHz; both
    // fits within 5 % on the 15 shapes of tools/run_w1.sh, B*H = 16..1600, T = 256..8192):
    //   (a) the analytic plan with 4-wave workgroups at the smallest c, two workgroups per CU:
    //           (rounds + 1/4 for the tail) x (12 k + 8 k * c)        rounds = workgroups / (2 * CUs)
    //       e.g. B=4, H=12, T=1024: 68.8 us against 79.8 us for the best single-round candidate (a chain table, c = 24);
    //   (b) ONE-WAVE workgroups, each running the pair n <-> nqb+1-n whole (nqb+1 units): every wave slot (two per SIMD)
    //       is refilled on its own, every job is equally long, and no wave waits at a workgroup's merge barrier for a
    //       sibling that shares its SIMD with a slower neighbour (what holds (a) to ~80 % of the MFMA peak on long
    //       sequences: BASELINE config 5's shard 1108 us under (a), 978 us = 89 % under (b)):
    //           (whole rounds + partial) x (16 k + 9.2 k * (nqb + 1))   rounds = jobs / (8 * CUs),
    //       where a partial round of fill f costs min(1, 1/2 + 3/2 f) of a whole one (the dispatcher does not spread the
    //       stragglers one per SIMD).  (b) wins when the jobs fill the wave slots a whole number of times, or many times.
    {
        Plan multi;
        multi.pp = pack_plan(nqb, 4);
        multi.use_table = false;
        multi.n_wgs = multi.pp.wgs_per_bh;
        const double rounds = (double)bh * multi.n_wgs / (2.0 * cus);
        multi.cycles = (fmax(rounds, 1.0) + 0.25) * (12000.0 + 8000.0 * multi.pp.chain);
        multi.family = "analytic, several rounds";
        if (!have || (rounds > 1.0 && !forced_old && !forced_packed && multi.cycles < best.cycles * 0.98)) { best = multi; have = true; }
    }
    static const int no_one_wave = env_int("CMHSA_NOWP", 0);
    static const int forced_lpt = env_int("CMHSA_LPT", -1);      // -1 = planner's choice, 0 = pairs only, G > 0: longest first in groups of G
    const bool multi_ok = !forced_old && !forced_packed && have;
    {
        Plan wp;
        wp.pp = pack_plan(nqb, 1);
        wp.use_table = false;
        wp.n_wgs = wp.pp.wgs_per_bh;
        const double rounds = (double)bh * wp.n_wgs / (8.0 * cus), whole = floor(rounds + 0.02), f = fmax(0.0, rounds - whole);
        wp.cycles = (whole + (f > 0.02 ? fmin(1.0, 0.5 + 1.5 * f) : 0.0)) * (16000.0 + 9200.0 * (nqb + 1));
        // less than half a round: the first jobs are spread one per SIMD and run alone (5 k cycles per unit; B=5 H=12 T=1024:
        // 73 us against 91 us for the analytic plan)
        if (whole == 0.0 && f <= 0.5) wp.cycles = 10000.0 + 5000.0 * (nqb + 1);
        wp.family = "one wave per pair of query blocks";
        if ((forced_waves == 1 && forced_lpt <= 0) ||
            (!no_one_wave && forced_lpt <= 0 && rounds > 0.3 && multi_ok && wp.cycles < best.cycles * 0.97)) {
            best = wp;
            have = true;
        }
    }
    //   (c) one-wave workgroups with ONE block each, dispatched longest first across ALL (b,h)s (grid x = (b,h) padded to a
    //       multiple of 8, so that a (b,h) stays on one XCD; y = block, longest first): the hardware's in-order dispatch
    //       onto whichever wave slot frees up is then longest-processing-time-first list scheduling -- no round
    //       quantisation, the tail is made of 1-unit jobs.  The jobs of a (b,h) that start together stream the same K/V
    //       units almost in step (L2 hits), but every (b,h) is live at once: only while all K and V fit the 256 MB
    //       infinity cache with room to spare (measured: B=64 H=25 T=512, 420 MB: +22 %; B=16 H=12 T=1024, 100 MB: -15 %).
    //           max( 15 k + 9.8 k * units per wave slot + 8 k * jobs per wave slot,  8 k + u * nqb )
    //       u = 9.2 k cycles per unit for the longest job, less when there are under two jobs per slot (its SIMD neighbour
    //       runs out of work: 8.5 k at 1.5 jobs per slot, measured)
    {
        Plan lf;
        lf.pp = pack_plan(nqb, 1);
        lf.pp.n_paired = 0;
        lf.pp.n_alone = nqb;
        lf.pp.wgs_per_bh = lf.n_wgs = nqb;
        lf.use_table = false;
        const int64_t bh_pad = (bh + 7) / 8 * 8;
        lf.longest_first = (int)min(bh_pad, forced_lpt > 0 ? (int64_t)(forced_lpt + 7) / 8 * 8 : bh_pad);
        const double slots = 8.0 * cus, units = (double)bh_pad * nqb * (nqb + 1) / 2 / slots, jobs = (double)bh_pad * nqb / slots;
        const double longest_unit = jobs >= 2.0 ? 9200.0 : 5000.0 + 4200.0 * fmin(1.0, fmax(0.0, (jobs - 0.5) / 1.2));
        lf.cycles = fmax(15000.0 + 9800.0 * units + 8000.0 * jobs, 8000.0 + longest_unit * nqb);
        lf.family = "one wave per query block, longest first";
        const bool fits_cache = (double)bh * nqb * kQBlock * 512.0 <= 208e6 && bh_pad <= 0x7fffffff;
        if ((forced_waves == 1 && forced_lpt > 0) ||
            (!no_one_wave && forced_lpt < 0 && fits_cache && jobs > 0.98 && multi_ok && lf.cycles < best.cycles * 0.97)) {
            best = lf;
            have = true;
        }
    }
    static const bool verbose = env_int("CMHSA_VERBOSE", 0) != 0;
    if (verbose)
        fprintf(stderr, "cmhsa plan: bh=%lld nqb=%d cus=%d -> %s, waves=%d chain=%d simd cap=%d wgs/bh=%d split blocks/bh=%d est=%.0f cycles\n",
                (long long)bh, nqb, cus, best.family, best.pp.waves, best.pp.chain, best.simd_cap, best.n_wgs, best.splits_per_bh, best.cycles);
#ifdef CMHSA_PLAN_ENV
    if (env_int("CMHSA_VERBOSE", 0) > 1 && best.use_table)
        for (int j = 0; j < best.n_wgs; ++j) {
            fprintf(stderr, "  wg %2d:", j);
            for (int t = 0; t < kMaxSeg; ++t) {
                const unsigned w0 = best.table[j * kTableWords + 2 * t], w1 = best.table[j * kTableWords + 2 * t + 1];
                if (w0 & 255u) fprintf(stderr, "  block %u units [%u,+%u) waves %02x%s", w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24, w1 ? " PART" : "");
            }
            fprintf(stderr, "\n");
        }
#endif
    return best;
}

// choose_plan() prices dozens of candidates; calls repeat the same few shapes, so the result is memoised per
// (shape, CU count of the device the call runs on, workspace available or not)
Plan cached_plan(int64_t bh, int nqb, int cus, bool allow_split) {
    struct Entry { int64_t bh; int nqb, cus; bool allow_split; Plan plan; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry &e : cache)
        if (e.bh == bh && e.nqb == nqb && e.cus == cus && e.allow_split == allow_split) return e.plan;
    if (cache.size() >= 256) cache.clear();
    cache.push_back(Entry{bh, nqb, cus, allow_split, choose_plan(bh, nqb, cus, allow_split)});
    return cache.back().plan;                        // by value: the entry may be gone once the lock is released
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

size_t plan_workspace_bytes(const Plan &pl, int64_t bh) {
    return pl.parts_per_bh ? kWsCounterBytes + (size_t)bh * pl.parts_per_bh * kPartFloats * 4 : 0;
}

// grid of a plan: (8, wgs_per_bh, bh/8) pins each (b,h) to one XCD; (bh, wgs_per_bh, 1) when bh is not a multiple of 8 or
// too large for gridDim.z; longest first: x = (b,h) in groups of G (a multiple of 8, so that a (b,h) stays on one XCD),
// y = block, longest first, z = group.  HIP limits: y, z <= 65535, and x * threads < 2^32.
int plan_grid(const Plan &plan, int64_t bh, dim3 &grid) {
    const PackPlan &pl = plan.pp;
    const bool xcd_map = bh % 8 == 0 && bh / 8 <= 65535 && !plan.longest_first;
    const int64_t lf_groups = plan.longest_first ? (bh + plan.longest_first - 1) / plan.longest_first : 0;
    if (lf_groups > 65535) return CMHSA_ERR_TOO_LARGE;
    grid = dim3(plan.longest_first ? (unsigned)plan.longest_first : xcd_map ? 8u : (unsigned)bh, (unsigned)pl.wgs_per_bh,
                plan.longest_first ? (unsigned)lf_groups : xcd_map ? (unsigned)(bh / 8) : 1u);
    if (pl.wgs_per_bh > 65535 || (int64_t)grid.x * (pl.waves * 64) >= (1LL << 32) ||
        (int64_t)grid.x * grid.z * pl.wgs_per_bh > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;
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
        case CMHSA_ERR_WORKSPACE: return "workspace too small (cmhsa_fp32_fwd_noalloc: this shape's plan needs more than the static "
                                         "CMHSA_NOALLOC_SCRATCH_BYTES; use cmhsa_fp32_fwd_ws with a workspace, or cmhsa_fp32_fwd)";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown cmhsa status";
    }
}

int cmhsa_version(void) { return (1 << 16) | 0; }

#ifdef CMHSA_STAMPS
// diagnostic build only: buf = device buffer of 8 waves x 128 stamps, block = blockIdx.x to stamp
int cmhsa_debug_set_stamps(unsigned long long *buf, int block) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_block), &block, sizeof(block));
    return (int)e;
}
int cmhsa_debug_set_block_times(unsigned long long *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_block_times), &buf, sizeof(buf));
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
    Plan plan = cached_plan(bh, (int)nqb, cus, workspace != nullptr);
    if (plan_workspace_bytes(plan, bh) > workspace_bytes) {
        if (reject_short_workspace) return CMHSA_ERR_WORKSPACE;
        plan = cached_plan(bh, (int)nqb, cus, false);             // a short workspace selects the plan that needs none
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
    if (plan.use_table) memcpy(p.seg_table, plan.table, sizeof(unsigned) * (size_t)plan.n_wgs * kTableWords);

    hipError_t e;
    if (waves == 1) e = launch<1, 2, false>(p, grid, stream);
    else if (!p.use_table) e = waves == 8 ? launch<8, 2, false>(p, grid, stream) : launch<4, 2, false>(p, grid, stream);
    else if (waves == 4) e = launch<4, 4, false>(p, grid, stream);
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
    const hipError_t e = hipMemsetAsync(workspace, 0, kWsCounterBytes, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CMHSA_OK : (int)e;
}

size_t cmhsa_fp32_fwd_workspace_bytes(int batch, int seq_len, int num_heads) {
    if (batch <= 0 || seq_len <= 0 || num_heads <= 0) return 0;
    const int64_t nqb = ((int64_t)seq_len + kQBlock - 1) / kQBlock, bh = (int64_t)batch * num_heads;
    if ((nqb + 16) * kKeyUnit > 0x7fffffffLL || bh > 0x7fffffffLL) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    return plan_workspace_bytes(cached_plan(bh, (int)nqb, device_cu_count(dev), true), bh);
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
    const Plan pl = cached_plan(bh, (int)nqb, cus, with_workspace != 0);
    snprintf(buf, (size_t)buf_len, "%s: %d-wave workgroups, %d per (b,h) x %lld (b,h) on %d CUs, chain cap %d, SIMD cap %d, "
             "%d of %lld query blocks per (b,h) split across workgroups, workspace %zu B, estimate %.0f cycles",
             pl.family, pl.pp.waves, pl.pp.wgs_per_bh, (long long)bh, cus, pl.pp.chain, pl.simd_cap, pl.splits_per_bh,
             (long long)nqb, plan_workspace_bytes(pl, bh), pl.cycles);
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
    const Plan plan = cached_plan(bh, (int)nqb, cus, with_workspace != 0);
    dim3 grid;
    const int grid_rc = plan_grid(plan, bh, grid);
    if (grid_rc != CMHSA_OK) return grid_rc;
    const int segs = plan.use_table ? kMaxSeg : 2;
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
        if (plan.use_table) {
            int n[kMaxSeg], lo[kMaxSeg], cnt[kMaxSeg];
            unsigned mask[kMaxSeg], x[kMaxSeg];
            decode_segments<kMaxSeg>(plan.table + j * kTableWords, j, (int)nqb, 0, 0, 1, 0, n, lo, cnt, mask, x);
            for (int i = 0; i < kMaxSeg; ++i) { rec[5 * i] = n[i]; rec[5 * i + 1] = lo[i]; rec[5 * i + 2] = cnt[i]; rec[5 * i + 3] = (int)mask[i]; rec[5 * i + 4] = (int)x[i]; }
        } else {
            int n[2], lo[2], cnt[2];
            unsigned mask[2], x[2];
            decode_segments<2>(nullptr, j, (int)nqb, plan.pp.n_paired, plan.pp.n_alone, plan.pp.chain, chain_magic, n, lo, cnt, mask, x);
            for (int i = 0; i < 2; ++i) { rec[5 * i] = n[i]; rec[5 * i + 1] = lo[i]; rec[5 * i + 2] = cnt[i]; rec[5 * i + 3] = (int)mask[i]; rec[5 * i + 4] = (int)x[i]; }
        }
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
