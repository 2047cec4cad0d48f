// cmhsa_common.h -- constants, kernel parameters and the segment decode shared by the kernel (cmhsa_fwd.hip) and the
// host-side planner (cmhsa_plan.h).  Included by cmhsa_fwd.hip only (one translation unit: everything here has internal
// linkage).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "cmhsa.h"

namespace {

constexpr int kHeadDim = 64;            // CMHSA.cu:432
constexpr int kQBlock = 32;             // query rows per workgroup (one MFMA N-tile)
constexpr int kKeyUnit = 32;            // keys per unit (one MFMA M-tile)
// split-KV ways inside a workgroup = waves per workgroup (template parameter kWaves):
//   8 -> one workgroup per CU (two of its waves per SIMD; 224 VGPRs allow no more): shortest chain per query block
//   4 -> two workgroups per CU: prologue/merge of one overlaps the main loop of the other
constexpr int kUnitFloats = kKeyUnit * kHeadDim;          // 2048 floats = 8 KiB
constexpr int kMaxSeg = 8;                                // query blocks ("segments") one workgroup can hold (template kSegs: 2, 4 or 8)
constexpr int kMaxTableWgs = 64;                          // workgroups per (b,h) a 4-segment table can describe (an 8-segment table: 32)
constexpr int kTableSize = kMaxTableWgs * 2 * 4;              // words of Params::seg_table: two per segment, 64 x 4 or 32 x 8 segments
constexpr int kMaxParts = 4;                              // parts (on different workgroups) a query block's keys may be split into
// One part's un-normalised result in the workspace: O image [8 pairs][64 lanes] float4 (the merge's native layout)
// followed by (m, l) of the 32 queries.  33 whole 256-B lines.
constexpr int kPartFloats = kUnitFloats + 64;
constexpr size_t kWsCounterBytes = 16384;                 // head of the workspace: one arrival counter per split query block
constexpr size_t kWsMaxParts = 1024;
constexpr size_t kWsMaxBytes = kWsCounterBytes + kWsMaxParts * kPartFloats * 4;       // = CMHSA_WORKSPACE_BYTES (include/cmhsa.h)
static_assert(kWsMaxBytes == CMHSA_WORKSPACE_BYTES, "include/cmhsa.h and the kernel disagree on the workspace size");

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
    // explicit plan (use_table != 0; choose_plan()): workgroup j of every (b,h) holds up to S = four or eight segments
    // (the kernel's kSegs), two table words each (seg_table[j * 2S + 2 i], [.. + 1]; a segment with n = 0 is absent, present
    // ones come first):
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
    // job lists (one-wave workgroups, the kernel's kList): list_p consecutive (b,h)s form a period whose blocks are dealt
    // to the gridDim.y lists of seg_table (8 words = 16 half-word entries each: units | (b,h) within the period << 8; 0 = end)
    int list_p, list_periods;
    int list_rev_max;                   // blocks of at most this many units are taken from the diagonal unit DOWN (0 = none)
    unsigned seg_table[kTableSize];
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
//   kSegs > 2: explicit plan, `tw` = the workgroup's 2 * kSegs table words (Params::seg_table);
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

}  // namespace
