// cmhsa_tiles16.h -- internal interface of the "16-row blocks" kernel family (cmhsa_fwd16.hip), used by the launcher and
// planner in cmhsa_fwd.hip / cmhsa_plan.h.  Not part of the C ABI (include/cmhsa.h): every symbol here has hidden
// visibility.
//
// The family: query blocks of 16 rows, key tiles of 16 rows, v_mfma_f32_16x16x4_f32; a
// workgroup is four waves, one per SIMD; every operand goes straight from global memory into MFMA operand registers
// (no LDS in the main loop, no prologue barrier).  A (b,h)'s blocks are dealt to its workgroups so that the workgroups
// carry (almost) the same number of tiles, and inside a workgroup the blocks are laid end to end and cut into four equal
// runs, one per wave (McNaughton's wrap-around rule): a wave runs a LIST of jobs -- whole blocks, or pieces of blocks
// whose other pieces are with its neighbours and are merged through LDS at the end.  It is the plan for grids that
// fit the chip in one round (BASELINE config 2 and the B = 1 launches of a generation, config 3).
#pragma once

#include <stddef.h>
#include <stdint.h>

#define CMHSA_HIDDEN __attribute__((visibility("hidden")))

namespace cmhsa_t16 {

constexpr int kWaves = 4;                // waves per workgroup
constexpr int kTableWords = 768;         // job words per (b,h): workgroups x 4 waves x jobs per wave (4 or 8)
constexpr int kMaxBlocks = 255;          // 16-row blocks per (b,h) a job word can name (seq_len <= 4080)

// One job = one run of key tiles of one 16-row query block, computed by one wave.
//   bits  0-7   block index (query rows [16 b, 16 b + 16); its key tiles are 0..b, tile b is the diagonal one)
//   bits  8-15  first tile of the run
//   bits 16-23  tiles in the run (0 = no job: the rest of the wave's list is empty)
//   bits 24-25  rank of this piece among the block's pieces (0 = the piece that starts the block)
//   bits 26-28  pieces the block is cut into (1 = whole: the wave stores O itself; up to 4, on consecutive waves)
//   bit  31     (first word of every wave of the workgroup) some block of this workgroup is cut: the waves meet at a barrier
inline unsigned job_word(int block, int lo, int cnt, int rank, int pieces) {
    return (unsigned)block | ((unsigned)lo << 8) | ((unsigned)cnt << 16) | ((unsigned)rank << 24) | ((unsigned)pieces << 26);
}
constexpr unsigned kWgMergesBit = 0x80000000u;

struct Plan16 {
    bool ok = false;
    int n16 = 0;                         // 16-row blocks per (b,h)
    int wgs_per_bh = 0;                  // workgroups per (b,h)
    int jobs_per_wave = 4;               // J: 4 or 8 (kernel template)
    int tiles_per_wave = 0;              // the longest wave's tiles (what the launch time follows)
    int max_jobs = 0;                    // the longest job list
    double cycles = 0.0;                 // estimate, comparable with the other families' (cmhsa_plan.h)
    unsigned table[kTableWords];         // [workgroup][wave][J]
};

// host: plan for bh (b,h)s of n16 blocks on `cus` compute units; false = the family does not apply
CMHSA_HIDDEN bool make_plan(int64_t bh, int n16, int cus, Plan16 &out);

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
