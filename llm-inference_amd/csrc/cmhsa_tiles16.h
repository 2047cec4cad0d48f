// cmhsa_tiles16.h -- internal interface of the "16-key tiles" kernel family (cmhsa_fwd16.hip), used by the launcher and
// planner in cmhsa_fwd.hip / cmhsa_plan.h.  Not part of the C ABI (include/cmhsa.h): every symbol here has hidden
// visibility.
//
// The family: key tiles of 16 rows, query blocks of 16 rows (sub = 1) or of two 16-row halves that
// share every K / V load (sub = 2), v_mfma_f32_16x16x4_f32; a workgroup is four waves (one per SIMD) or eight (two per
// SIMD), one workgroup per CU; every operand goes straight from global memory into MFMA operand registers (no LDS in the
// main loop, no prologue barrier).  A (b,h)'s blocks are dealt to its workgroups so that the workgroups carry (almost)
// the same number of tiles, and inside a workgroup the blocks are laid end to end and cut into four equal runs, one per
// SIMD (McNaughton's wrap-around rule): a wave runs a LIST of jobs -- whole blocks, or pieces of blocks whose other
// pieces are with its neighbours and are merged through LDS at the end.  It is the plan for grids that fit the chip in
// one round (BASELINE config 2 and the B = 1 launches of a generation, config 3).
#pragma once

#include <stddef.h>
#include <stdint.h>

#define CMHSA_HIDDEN __attribute__((visibility("hidden")))

namespace cmhsa_t16 {

constexpr int kMaxWaves = 8;             // waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD)
constexpr int kWordsPerWg = 32;          // job words per workgroup: 4 waves x 8 jobs, or 8 waves x 4 jobs
constexpr int kMaxWgs = 28;              // workgroups per (b,h) the table holds
constexpr int kTableWords = kMaxWgs * kWordsPerWg;
constexpr int kMaxTiles = 255;           // granules of key tiles a job word can count (a granule = one tile, or two for 32-row blocks
                                         // of more than 255 tiles: seq_len <= 4080 with 16-row blocks, <= 8160 with 32-row blocks)

// One job = one run of key tiles of one query block, computed by one wave.
//   bits  0-7   block index b (query rows [16 sub b, 16 sub (b + 1)); its key tiles are 0 .. sub (b + 1) - 1; tile
//               sub b + s is the diagonal tile of the block's half s, later tiles are masked for that half)
//   bits  8-15  first tile of the run        } in granules of Plan16::granule tiles (1, or 2 where a block has more
//   bits 16-23  tiles in the run             } than 255 tiles); 0 tiles = no job: the rest of the wave's list is empty
//   bits 24-26  rank of this piece among the block's pieces (0 = the piece that starts the block)
//   bits 27-30  pieces the block is cut into (1 = whole: the wave stores O itself; up to 8, on consecutive lists)
//   bit  31     (first word of every list of the workgroup) some block of this workgroup is cut: the waves meet at a barrier
// A workgroup's job lists are numbered in RUN ORDER u = 0 .. waves-1: the blocks of the workgroup, laid end to end, are cut
// into four runs, one per SIMD; with eight waves each SIMD's run is halved between its two waves.  Wave w of an 8-wave
// workgroup sits on SIMD w % 4 (waves w and w + 4 share one), so it runs list u = 2 (w % 4) + w / 4; with four waves u = w.
inline unsigned job_word(int block, int lo, int cnt, int rank, int pieces) {
    return (unsigned)block | ((unsigned)lo << 8) | ((unsigned)cnt << 16) | ((unsigned)rank << 24) | ((unsigned)pieces << 27);
}
constexpr unsigned kWgMergesBit = 0x80000000u;

struct Plan16 {
    bool ok = false;
    int sub = 1;                         // 16-row halves per query block: 1 or 2 (kernel template)
    int granule = 1;                     // tiles per count of a job word's first-tile / tiles fields (1 or 2)
    int waves = 4;                       // waves per workgroup: 4 or 8 (kernel template); jobs per wave = 32 / waves
    bool contig = false;                 // workgroup -> (b,h) map: false = grid (workgroup, (b,h)); true = 1-D, every XCD a
                                         // contiguous run of the (b,h)-major list (K / V of a (b,h) through one XCD's L2)
    int blocks = 0;                      // query blocks per (b,h)
    int wgs_per_bh = 0;                  // workgroups per (b,h)
    int tiles_per_simd = 0;              // tiles on the busiest SIMD (what the launch time follows)
    int max_jobs = 0;                    // the longest job list
    double cycles = 0.0;                 // estimate, comparable with the other families' (cmhsa_plan.h)
    unsigned table[kTableWords];         // [workgroup][list u][32 / waves]
};

// host: plan for bh (b,h)s of seq_len rows on `cus` compute units; false = the family does not apply.
// force_sub / force_waves (0 = planner's choice): experiments and tests only.
CMHSA_HIDDEN bool make_plan(int64_t bh, int seq_len, int cus, Plan16 &out, int force_sub = 0, int force_waves = 0, int force_contig = -1);

// host: launch.  Shapes are validated by the caller (cmhsa_fwd.hip check_forward_args); returns a hipError_t as int.
CMHSA_HIDDEN int launch(const Plan16 &plan, const float *q, const float *k, const float *v, float *o,
                        int64_t q_rs, int64_t k_rs, int64_t v_rs, int64_t o_rs,
                        int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                        int64_t bh, int seq_len, int num_heads, void *stream);

#ifdef CMHSA_STAMPS
CMHSA_HIDDEN int debug_set_stamps(unsigned long long *buf, int block);
CMHSA_HIDDEN int debug_set_block_times(unsigned long long *buf);
#endif

}  // namespace cmhsa_t16
