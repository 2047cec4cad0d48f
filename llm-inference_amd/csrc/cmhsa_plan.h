// cmhsa_plan.h -- host side of the attention kernel: which workgroup and which waves compute which 32-key units of
// which 32-row query block (the "plan"), chosen per (B*H, T, CU count) by measured cost models and memoised.
// No device code here.  Included by cmhsa_fwd.hip only (one translation unit: everything has internal linkage).
// cmhsa_plan_dump() exports any plan as integers; tests/test_plan.py checks on the CPU that every unit of every block is
// covered exactly once.
#pragma once

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "cmhsa_common.h"
#include "cmhsa_tiles16.h"

namespace {

// ---- plan selection (host) ---------------------------------------------------------------------------------------
// A plan says which query blocks (or parts of them) each workgroup of a (b,h) holds and which of its waves work on
// each.  Families:
//   analytic      pack_plan(): a block alone or a pair n <-> R+1-n, a block of n units on ceil(n / c) waves; decoded in
//                 the kernel from three integers; any size; 8-wave workgroups (one per CU) or 4-wave (two per CU);
//   chain table   the same wave rule, blocks packed greedily (longest first into the feasible workgroup with the fewest
//                 units), up to 4 per workgroup, handed to the kernel as table words;
//   packed table  pack_attempt(): a cap L on the units any SIMD runs (waves w and w+4 of an 8-wave workgroup share one),
//                 blocks placed wave by wave under that cap; with a workspace, a block may be cut into two PARTS held by
//                 different workgroups (its keys split in two ranges, merged through the workspace by the workgroup that
//                 finishes last) -- the only way to put fewer units on a SIMD than the longest block has per SIMD of one
//                 CU, and to use every CU when the blocks do not divide evenly;
//   wave pairs    one-wave workgroups, each running the blocks n and nqb+1-n whole, one after the other;
//   longest first one-wave workgroups with one block each, the grid ordered by block length across all (b,h)s.
// Table plans describe at most 64 workgroups per (b,h) and blocks of at most 255 units.
// Grids that fit the chip in ONE round (config 2): among the single-round candidates the cheapest by
//     cycles = kFixed + (units on the busiest SIMD) * kUnit  [+ kHandoff in a workgroup that holds a part]
// wins; an 8-wave workgroup owns its CU (two of its waves per SIMD), two 4-wave workgroups share one.  Larger grids:
// the 4-wave analytic plan at the smallest c, wave pairs or longest first, each priced by a fit to measured launch
// times (choose_plan()).
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
    int segs = 4;                                // table plans: segments per workgroup record (4 or 8)
    unsigned table[kTableSize];                  // table plans: Params::seg_table
    int list_p = 0, list_q = 0;                  // > 0: job lists (one-wave workgroups): a period of list_p (b,h)s on list_q waves; table = the lists
    int list_units = 0, list_jobs = 0;           //      units / jobs of the longest list
    int list_rev_max = 0;                        //      blocks of at most this many units are taken in descending order
    bool tiles16 = false;                        // the 16-row-block family (cmhsa_fwd16.hip): everything above but cycles / family is unused
    cmhsa_t16::Plan16 t16;
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

constexpr int kChainSegs = 4;                    // chain tables and the analytic plan's pricing: at most four blocks per workgroup

void price(Plan &pl, const int (*wg_blocks)[kChainSegs], int n_wgs, int64_t bh, int cus) {
    const int waves = pl.pp.waves;
    const int64_t wgs = bh * n_wgs;
    const bool shared = waves == 4 && wgs > cus;                    // a second 4-wave workgroup on the same CU
    double longest = 0.0;
    for (int j = 0; j < n_wgs; ++j) {
        const int l = busiest_simd(wg_blocks[j], kChainSegs, pl.pp.chain, waves);
        longest = fmax(longest, kFixedCycles + (shared ? 2.0 : 1.0) * l * kUnitCycles);
    }
    pl.n_wgs = n_wgs;
    pl.single_round = wgs <= (int64_t)cus * (waves == 8 ? 1 : 2);
    pl.cycles = longest;                 // meaningful for single-round grids only, the only ones compared
}

bool chain_table_plan(Plan &pl, int nqb, int waves, int c, int64_t bh, int cus) {
    if (nqb > 255) return false;
    int blocks[kMaxTableWgs][kChainSegs] = {}, units[kMaxTableWgs] = {}, free_waves[kMaxTableWgs], nseg[kMaxTableWgs] = {};
    int n_wgs = 0;
    for (int n = nqb; n >= 1; --n) {
        const int g = (n + c - 1) / c;
        int pick = -1;
        for (int j = 0; j < n_wgs; ++j)
            if (free_waves[j] >= g && nseg[j] < kChainSegs && (pick < 0 || units[j] < units[pick])) pick = j;
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
            for (int t = 0; t < kChainSegs; ++t) { const int x = blocks[i][t]; blocks[i][t] = blocks[m][t]; blocks[m][t] = x; }
            const int x = units[i]; units[i] = units[m]; units[m] = x;
        }
    }
    pl.pp = pack_plan(nqb, waves, c);
    pl.pp.wgs_per_bh = n_wgs;
    pl.use_table = true;
    pl.segs = kChainSegs;
    pl.family = "chain table";
    for (int j = 0; j < n_wgs; ++j) {
        int w0 = 0;
        for (int t = 0; t < kChainSegs; ++t) {
            const int n = blocks[j][t], g = n ? (n + c - 1) / c : 0;
            pl.table[j * 2 * kChainSegs + 2 * t] = n ? seg_word0(n, 0, n, ((1u << g) - 1u) << w0) : 0u;
            pl.table[j * 2 * kChainSegs + 2 * t + 1] = 0u;
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
    std::vector<int> flat((size_t)pl.pp.wgs_per_bh * kChainSegs, 0);
    int (*blocks)[kChainSegs] = reinterpret_cast<int (*)[kChainSegs]>(flat.data());
    for (int j = 0; j < pl.pp.wgs_per_bh; ++j) {
        if (j < pl.pp.n_alone) blocks[j][0] = nqb - j;
        else { const int t = j - pl.pp.n_alone; blocks[j][0] = pl.pp.n_paired - t; if (t + 1 != blocks[j][0]) blocks[j][1] = t + 1; }
    }
    price(pl, blocks, pl.pp.wgs_per_bh, bh, cus);
}

// Experiments only (tests/helpers/libcmhsa_hip_planenv.so and tools/ build the library with -DCMHSA_PLAN_ENV):
// CMHSA_WAVES=4|8, CMHSA_CHAIN=c, CMHSA_TABLE=0|1 force an analytic / chain plan, CMHSA_SEGS=4|8 the packed tables' segment count, CMHSA_WAVES=1 one-wave workgroups
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

// Planning-time budget: choose_plan() grants a number of slot-set evaluations; when they are used up every further
// packing attempt fails at once and the candidates found so far compete (long sequences on one or two (b,h)s would
// otherwise spend hundreds of milliseconds packing 255 blocks into 64 workgroups for every cap tried).
thread_local int64_t g_pack_budget = 0;

// best set of g free slots of `wg` for an item of cnt units under `cap`; 0 if none
unsigned best_slots(const PackWg &wg, int cnt, int g, int cap) {
    if (g_pack_budget <= 0) return 0;
    g_pack_budget -= 1;
    // the 8-bit masks by population count, ascending (at most C(8,4) = 70 per count)
    static const struct ByCount {
        unsigned char mask[9][70];
        int n[9];
        ByCount() {
            for (int c = 0; c < 9; ++c) n[c] = 0;
            for (unsigned m = 1; m < 256; ++m) { const int c = __builtin_popcount(m); mask[c][n[c]++] = (unsigned char)m; }
        }
    } by_count;
    if (g < 1 || g > 8) return 0;
    unsigned best = 0;
    int best_score = 1 << 30;
    const unsigned free_mask = ~wg.used & 0xffu;
    for (int mi = 0; mi < by_count.n[g]; ++mi) {
        const unsigned m = by_count.mask[g][mi];
        if ((m & free_mask) != m) continue;
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
bool pack_attempt(int nqb, int L, bool allow_split, double psplit, double wave_share, Rng &rng, int g_max, int max_seg, std::vector<PackWg> &wgs, int &n_splits) {
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
        if (g_pack_budget <= 0) return false;
        const bool part = it.split >= 0;
        // candidates: existing workgroups, fullest first, then a new one
        std::vector<int> order(wgs.size());
        for (size_t j = 0; j < order.size(); ++j) order[j] = (int)j;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wgs[a].total > wgs[b].total; });
        order.push_back(-1);
        bool placed = false;
        for (int j : order) {
            g_pack_budget -= 1;                                          // (a visit costs about one slot-set evaluation)
            PackWg fresh;
            PackWg &wg = j < 0 ? fresh : wgs[j];
            if (wg.nseg >= max_seg) continue;
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

// max_seg = 4 or 8 segments per workgroup (8: at most 32 workgroups per (b,h) fit the table; the kernel stages eight query
// blocks, 130 KiB of LDS)
bool packed_plan(Plan &pl, int nqb, int L, bool allow_split, int64_t bh, int cus, bool eager_split = false, int max_seg = 4) {
    if (nqb > 255 || bh > cus) return false;
    const int g_max = (int)min((int64_t)(kTableSize / (2 * max_seg)), (int64_t)cus / bh);
    {   // cannot fit whatever the order: more blocks than segment slots, or more waves needed than there are
        int64_t waves_needed = 0;
        for (int n = 1; n <= nqb; ++n) waves_needed += (n + L - 1) / L;
        if (nqb > (int64_t)g_max * max_seg || waves_needed > (int64_t)g_max * 8) return false;
    }
    g_pack_budget = 20000;                       // slot-set evaluations for this (cap, segments, split) candidate: a few ms
    Rng rng{0x9E3779B97F4A7C15ull ^ ((uint64_t)nqb << 32) ^ ((uint64_t)L << 16) ^ (uint64_t)g_max ^ (allow_split ? 0x5555ull : 0ull) ^ ((uint64_t)(max_seg == 8) << 60)
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
        if (!pack_attempt(nqb, L, allow_split, psplit, wave_share, rng, g_max, max_seg, cand, cand_splits)) continue;
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
    pl.segs = max_seg;
    pl.family = n_splits ? "packed table, blocks split across workgroups" : "packed table";
    pl.n_wgs = (int)wgs.size();
    pl.simd_cap = L;
    pl.splits_per_bh = n_splits;
    pl.parts_per_bh = 2 * n_splits;
    pl.single_round = true;
    for (int j = 0; j < pl.n_wgs; ++j) {
        const PackWg &wg = wgs[j];
        for (int t = 0; t < max_seg; ++t) {
            unsigned w0 = 0, w1 = 0;
            if (t < wg.nseg) {
                w0 = seg_word0(wg.seg_n[t], wg.seg_lo[t], wg.seg_cnt[t], wg.seg_mask[t]);
                if (wg.seg_split[t] >= 0)        // first slot | part index << 8 | parts << 12 | split id << 16
                    w1 = (unsigned)(2 * wg.seg_split[t]) | ((wg.seg_lo[t] ? 1u : 0u) << 8) | (2u << 12) | ((unsigned)wg.seg_split[t] << 16);
            }
            pl.table[j * 2 * max_seg + 2 * t] = w0;
            pl.table[j * 2 * max_seg + 2 * t + 1] = w1;
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

// ---- job lists (the kernel's kList variant): one-wave workgroups that each run a list of whole query blocks of several
// (b,h)s.  A PERIOD of p consecutive (b,h)s is dealt to q lists (waves); every period uses the same lists, so the table
// is q x 16 half-word entries whatever B*H is.  q = wave slots (x rounds) / periods, so that the launch is one round (or
// a whole number of rounds) of waves that all carry (almost) the same number of units.
//   schedule 1 ("teams"): the lists form teams of nqb; a team takes TWO (b,h)s A and B at a time.  With h = nqb / 2, the
//       blocks of more than h units are "long", the others "short".  Waves 0 .. h-1: A's long block nqb-i, then B's short
//       block i+1; the other waves: B's long block, then A's short block (an odd nqb's middle blocks of A and B share one
//       wave) -- nqb+1 units for everyone.  Long blocks START together and are streamed in ascending order, short blocks END
//       together and are streamed in DESCENDING order (the kernel's list_rev_max = h): either way the waves on one (b,h)
//       read the same K / V unit at the same time (L2 hits for all but one; measured: 482 -> 461 us at B64 H25 T512 against
//       every block ascending).  The order depends on the block's length only, never on its place in a list.
//       The last few (b,h)s of the period are dealt block by
//       block, longest first, to the least loaded list, which evens out the ends;
//   schedule 0: every (b,h) in turn, longest block first, to the least loaded list; the last `tail` (b,h)s pooled as above.
struct ListSchedule { int max_units = 0, max_jobs = 0; unsigned short e[64][16]; int len[64], load[64]; };

bool list_schedule(ListSchedule &ls, int p, int q, int nqb, int sched, int tail) {
    for (int l = 0; l < q; ++l) { ls.len[l] = 0; ls.load[l] = 0; }
    auto put = [&](int l, int bh_i, int n) {
        if (ls.len[l] >= 16) return false;
        ls.e[l][ls.len[l]++] = (unsigned short)(n | (bh_i << 8));
        ls.load[l] += n;
        return true;
    };
    auto least = [&]() {
        int best = -1;
        for (int l = 0; l < q; ++l)
            if (ls.len[l] < 16 && (best < 0 || ls.load[l] < ls.load[best])) best = l;
        return best;
    };
    int next_bh = 0;
    if (sched == 1) {
        const int teams = q / nqb;
        if (teams < 1) return false;
        const int want_tail = max(1, (q * 5 / 2 + nqb - 1) / nqb);       // blocks enough for the pooled tail to even out the ends
        const int phases = max(0, (p - want_tail) / (2 * teams));
        const int h = nqb / 2;
        for (int k = 0; k < phases; ++k)
            for (int t = 0; t < teams; ++t) {
                const int a = next_bh, b = next_bh + 1, l0 = t * nqb;
                for (int i = 0; i < h; ++i) {
                    if (!put(l0 + i, a, nqb - i) || !put(l0 + i, b, i + 1)) return false;                       // A long, B short
                    if (!put(l0 + nqb - h + i, b, nqb - i) || !put(l0 + nqb - h + i, a, i + 1)) return false;   // B long, A short
                }
                if (nqb & 1)
                    if (!put(l0 + h, a, h + 1) || !put(l0 + h, b, h + 1)) return false;                         // both middle blocks
                next_bh += 2;
            }
        tail = p - next_bh;
    }
    tail = min(tail, p - next_bh);
    for (; next_bh < p - tail; ++next_bh)
        for (int n = nqb; n >= 1; --n) {
            const int l = least();
            if (l < 0 || !put(l, next_bh, n)) return false;
        }
    for (int n = nqb; n >= 1; --n)
        for (int b = next_bh; b < p; ++b) {
            const int l = least();
            if (l < 0 || !put(l, b, n)) return false;
        }
    ls.max_units = ls.max_jobs = 0;
    for (int l = 0; l < q; ++l) { ls.max_units = max(ls.max_units, ls.load[l]); ls.max_jobs = max(ls.max_jobs, ls.len[l]); }
    return true;
}

// cycles of a launch of job lists (at the 2.4 GHz the other estimates assume): R rounds of waves, two per SIMD, 9.1 k cycles
// per unit and wave, 7 k per block (its boundary is ~150 VALU instructions that each wait for a gap in the SIMD partner's
// MFMA stream, and its first unit takes the max path), one prologue per wave.  Fit to B = 16..64, H = 12 / 25, T = 256..1024
// (profiles/r03_experiments/job_lists_vs_pairs.log), within 3 % where the family is picked.
double list_cycles(int rounds, int max_units, int max_jobs) { return rounds * (16000.0 + 9100.0 * max_units + 7000.0 * max_jobs); }

bool list_plan(Plan &out, int64_t bh, int nqb, int cus, int forced_sched, int forced_p) {
    static const bool list_rev = env_int("CMHSA_LIST_REV", 1) != 0;     // (experiments: 0 = every block in ascending order)
    if (nqb < 2 || nqb > 64 || bh < 8 || cus < 1) return false;
    const int64_t slots = 8LL * cus, jobs = bh * nqb, units_per_bh = (int64_t)nqb * (nqb + 1) / 2;
    const int64_t rounds = max((int64_t)1, (jobs + 13 * slots - 1) / (13 * slots));      // <= 13 jobs per list on average (16 fit)
    const int64_t waves_target = rounds * slots;
    if (bh * units_per_bh < waves_target * (3 * nqb / 2)) return false;                  // lists too short for whole blocks to balance
    if (rounds > 64) return false;
    // candidates: period lengths whose ideal mean list load is lowest (waves are only lost to q = floor(...))
    struct Cand { int p, q; double mean; };
    Cand cands[64];
    int n_cands = 0;
    for (int p = 1; p <= 64 && p <= bh; ++p) {
        if (forced_p > 0 && p != forced_p) continue;
        const int64_t periods = (bh + p - 1) / p;
        const int q = (int)min((int64_t)64, waves_target / periods);
        if (q < 1 || p * nqb > 14 * q) continue;
        cands[n_cands++] = Cand{p, q, (double)p * units_per_bh / q};
    }
    if (!n_cands) return false;
    std::sort(cands, cands + n_cands, [](const Cand &a, const Cand &b) { return a.mean < b.mean || (a.mean == b.mean && a.p < b.p); });
    bool have = false;
    ListSchedule best, ls;
    int best_p = 0, best_q = 0;
    double best_cycles = 0.0;
    for (int ci = 0; ci < n_cands && ci < 6; ++ci) {
        const int p = cands[ci].p, q = cands[ci].q;
        for (int sched = 1; sched >= 0; --sched) {
            if (forced_sched >= 0 ? sched != forced_sched : sched == 0) continue;     // (schedule 0 scatters a (b,h)'s blocks in time: 691 against 470 us at B64 H25 T512)
            for (int tail = 0; tail <= (sched ? 0 : 8); tail += 2) {
                int qq = q;
                if (sched == 1) qq = q / nqb * nqb;                      // whole teams
                if (qq < 1 || !list_schedule(ls, p, qq, nqb, sched, tail)) continue;
                const int64_t periods = (bh + p - 1) / p;
                const int r = (int)((periods * qq + slots - 1) / slots);
                const double c = list_cycles(r, ls.max_units, ls.max_jobs);
                if (!have || c < best_cycles) { best = ls; best_p = p; best_q = qq; best_cycles = c; have = true; }
            }
        }
    }
    if (!have) return false;
    out = Plan();
    out.pp = pack_plan(nqb, 1);
    out.pp.wgs_per_bh = out.n_wgs = 0;
    out.use_table = false;
    out.list_p = best_p;
    out.list_q = best_q;
    out.list_units = best.max_units;
    out.list_jobs = best.max_jobs;
    out.list_rev_max = list_rev ? nqb / 2 : 0;
    const int64_t periods = (bh + best_p - 1) / best_p;
    out.cycles = list_cycles((int)((periods * best_q + slots - 1) / slots), best.max_units, best.max_jobs);
    out.family = "job lists, one wave per list";
    memset(out.table, 0, sizeof(out.table));
    for (int l = 0; l < best_q; ++l)
        for (int i = 0; i < best.len[l]; ++i) out.table[l * 8 + i / 2] |= (unsigned)best.e[l][i] << (16 * (i & 1));
    return true;
}


Plan choose_plan(int64_t bh, int nqb, int seq_len_hint, int cus, bool allow_split) {
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
    // packed plans: the smallest SIMD caps that could hold the work, without and (given a workspace) with split blocks;
    // four segments per workgroup first, then eight (more blocks per workgroup: long sequences on few (b,h)s, where four
    // segments x the workgroups the CUs allow cannot hold every block; the eight-segment kernel stages eight query blocks)
    static const int forced_segs = env_int("CMHSA_SEGS", 0);
    for (int max_seg = 4; max_seg <= 8 && !forced_old && nqb <= 255 && bh <= cus; max_seg += 4) {
        if (forced_segs && forced_segs != max_seg) continue;
        const int g_max = (int)min((int64_t)(kTableSize / (2 * max_seg)), (int64_t)cus / bh);
        if (g_max < 1) continue;
        const int64_t total = (int64_t)nqb * (nqb + 1) / 2;
        const int l_min = forced_packed ? forced_packed : (int)max((int64_t)1, (total + 4 * g_max - 1) / (4 * g_max));
        // caps tried: the four smallest, then -- the greedy packer gives up above ~93 % fill, which long sequences reach
        // (96 blocks on 21 workgroups) -- up to a fifth more in coarser steps, stopping at the first that packs
        int caps[8], n_caps = 0;
        for (int i = 0; i < 4; ++i) caps[n_caps++] = l_min + i;
        for (int pct = 106; pct <= 121 && !forced_packed; pct += 5) {
            const int L = (l_min * pct + 99) / 100;
            if (L > caps[n_caps - 1]) caps[n_caps++] = L;
        }
        if (forced_packed) n_caps = 1;
        bool packed_found = false;
        for (int ci = 0; ci < n_caps && !(packed_found && ci >= 4); ++ci) {
            const int L = caps[ci];
            for (int split = 0; split <= (allow_split ? 1 : 0); ++split) {
                if (forced_split >= 0 && forced_split != split && (allow_split || forced_split == 0)) continue;
                Plan pl;
                if (!packed_plan(pl, nqb, L, split != 0, bh, cus, split != 0 && forced_split == 1, max_seg)) continue;
                if (split && !pl.splits_per_bh) continue;            // the same plan as split = 0
                packed_found = true;
                if (max_seg == 8) pl.cycles += 1500.0;               // twice the query staging, a longer segment decode
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
    //   (e) job lists: one-wave workgroups, each a list of whole blocks of several (b,h)s, ONE round of equally long waves
    //       (list_plan() above).  Many (b,h)s of short sequences: B=64 H=25 T=512 is 6.25 rounds of wave pairs.
    {
        static const int forced_list = env_int("CMHSA_LIST", -1);       // -1 = planner's choice, 0 = never, 1 = wherever it applies
        static const int forced_sched = env_int("CMHSA_LIST_SCHED", -1), forced_p = env_int("CMHSA_LIST_P", 0);
        Plan lp;
        if (forced_list != 0 && multi_ok && forced_waves <= 1 && forced_lpt < 0 && bh <= 0x7fffffff / 64 &&
            list_plan(lp, bh, nqb, cus, forced_sched, forced_p) && (forced_list == 1 || lp.cycles < best.cycles * 0.97)) {
            best = lp;
            have = true;
        }
    }
    //   (d) 16-row blocks on v_mfma_f32_16x16x4_f32 (cmhsa_fwd16.hip): four-wave workgroups, one per CU, every wave a list
    //       of jobs cut so that all SIMDs carry the same number of 16 x 16 tiles; no LDS staging, no prologue barrier.
    //       Grids of one round (bh <= CUs, T <= 4080): a quarter of the 32 x 32 unit's quantum and half its fixed cost.
    {
        static const int forced_t16 = env_int("CMHSA_T16", -1);      // -1 = planner's choice, 0 = never, 1 = wherever it applies
        static const int forced_sub = env_int("CMHSA_T16_SUB", 0), forced_w16 = env_int("CMHSA_T16_WAVES", 0), forced_contig = env_int("CMHSA_T16_CONTIG", -1);
        Plan t;
        if (forced_t16 != 0 && !forced_old && !forced_packed && forced_waves != 1 &&
            cmhsa_t16::make_plan(bh, seq_len_hint, cus, t.t16, forced_sub, forced_w16, forced_contig) &&
            (forced_t16 == 1 || !have || t.t16.cycles < best.cycles * 0.97)) {
            t.tiles16 = true;
            t.pp = pack_plan(nqb, t.t16.waves);
            t.pp.wgs_per_bh = t.n_wgs = t.t16.wgs_per_bh;
            t.single_round = true;
            t.cycles = t.t16.cycles;
            t.family = t.t16.sub == 2 ? "16-key tiles, 32-row blocks, job lists" : "16-key tiles, 16-row blocks, job lists";
            best = t;
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
            for (int t = 0; t < best.segs; ++t) {
                const unsigned w0 = best.table[j * 2 * best.segs + 2 * t], w1 = best.table[j * 2 * best.segs + 2 * t + 1];
                if (w0 & 255u) fprintf(stderr, "  block %u units [%u,+%u) waves %02x%s", w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24, w1 ? " PART" : "");
            }
            fprintf(stderr, "\n");
        }
#endif
    return best;
}

// choose_plan() prices dozens of candidates; calls repeat the same few shapes, so the result is memoised per
// (shape, CU count of the device the call runs on, workspace available or not)
// (seq_len: the 16-row family's block count is ceil(T / 16) -- two per 32-row block, or one fewer; the other families use nqb only)
Plan cached_plan(int64_t bh, int seq_len, int cus, bool allow_split) {
    const int nqb = (seq_len + kQBlock - 1) / kQBlock, n16 = (seq_len + 15) / 16;
    struct Entry { int64_t bh; int nqb, n16, cus; bool allow_split; Plan plan; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry &e : cache)
        if (e.bh == bh && e.nqb == nqb && e.n16 == n16 && e.cus == cus && e.allow_split == allow_split) return e.plan;
    if (cache.size() >= 256) cache.clear();
    cache.push_back(Entry{bh, nqb, n16, cus, allow_split, choose_plan(bh, nqb, seq_len, cus, allow_split)});
    return cache.back().plan;                        // by value: the entry may be gone once the lock is released
}

size_t plan_workspace_bytes(const Plan &pl, int64_t bh) {
    return pl.parts_per_bh ? kWsCounterBytes + (size_t)bh * pl.parts_per_bh * kPartFloats * 4 : 0;
}

// grid of a plan: (8, wgs_per_bh, bh/8) pins each (b,h) to one XCD; (bh, wgs_per_bh, 1) when bh is not a multiple of 8 or
// too large for gridDim.z; longest first: x = (b,h) in groups of G (a multiple of 8, so that a (b,h) stays on one XCD),
// y = block, longest first, z = group.  HIP limits: y, z <= 65535, and x * threads < 2^32.
int plan_grid(const Plan &plan, int64_t bh, dim3 &grid) {
    const PackPlan &pl = plan.pp;
    if (plan.list_p) {                               // job lists: x = XCD, y = list, z = period / 8
        const int64_t periods = (bh + plan.list_p - 1) / plan.list_p, z = (periods + 7) / 8;
        if (z > 65535) return CMHSA_ERR_TOO_LARGE;
        grid = dim3(8u, (unsigned)plan.list_q, (unsigned)z);
        return CMHSA_OK;
    }
    const bool xcd_map = bh % 8 == 0 && bh / 8 <= 65535 && !plan.longest_first;
    const int64_t lf_groups = plan.longest_first ? (bh + plan.longest_first - 1) / plan.longest_first : 0;
    if (lf_groups > 65535) return CMHSA_ERR_TOO_LARGE;
    grid = dim3(plan.longest_first ? (unsigned)plan.longest_first : xcd_map ? 8u : (unsigned)bh, (unsigned)pl.wgs_per_bh,
                plan.longest_first ? (unsigned)lf_groups : xcd_map ? (unsigned)(bh / 8) : 1u);
    if (pl.wgs_per_bh > 65535 || (int64_t)grid.x * (pl.waves * 64) >= (1LL << 32) ||
        (int64_t)grid.x * grid.z * pl.wgs_per_bh > 0x7fffffffLL) return CMHSA_ERR_TOO_LARGE;
    return CMHSA_OK;
}

}  // namespace
