"""Host-side checks of the plans the library chooses (no GPU: cmhsa_plan_dump with an explicit CU count).

A plan says which workgroup and which waves compute which 32-key units of which 32-row query block.  Whatever family
the cost model picks for a shape -- analytic, chain table, packed table with or without blocks split across
workgroups, one-wave workgroups in pairs or longest first -- the kernel is only correct if

  * every unit [0, n) of every query block n = 1..nqb is covered EXACTLY once (whole blocks, or parts that tile it),
  * a workgroup's segments use disjoint sets of its waves, and no segment has more waves than units (the kernel's
    "every active wave owns at least one unit"),
  * table plans stay inside their encoding (n <= 255; 4 segments per workgroup and <= 64 workgroups per (b,h), or 8 and 32),
  * parts carry consistent bookkeeping (part index, part count, slots, split id inside the workspace the plan asks for),
  * the grid covers every (b,h) and stays inside HIP's limits.

The GPU parity tests exercise a few dozen shapes; this sweeps several thousand plans on the CPU.
"""
import itertools
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference_amd"))

import cmhsa  # noqa: E402

WORKSPACE_BYTES = 16384 + 1024 * (2048 + 64) * 4          # CMHSA_WORKSPACE_BYTES (include/cmhsa.h)


def check_tiles16_plan(plan, B, T, H, cus):
    """The 16-key-tile family (csrc/cmhsa_tiles16.h): per workgroup 4 or 8 job lists in run order; a job = a run of key tiles
    of one query block of 16 * sub rows.  Every tile [0, sub * (b + 1)) of every block b exactly once; the pieces of a cut
    block on consecutive lists of ONE workgroup with ranks 0, 1, ... in key order; per list at most one piece that starts
    its block and one later piece (the two LDS slots); the barrier flag on every list of a workgroup that cuts a block, on
    none otherwise; lists packed at the front; one workgroup per CU."""
    tag = (B, T, H, cus)
    sub, W, J = plan["sub"], plan["waves"], plan["jobs_per_wave"]
    nblocks = (T + 16 * sub - 1) // (16 * sub)
    assert sub in (1, 2) and W in (4, 8) and J == 32 // W and plan["blocks"] == nblocks, tag
    assert 1 <= plan["wgs_per_bh"] <= 28 and B * H * plan["wgs_per_bh"] <= cus, tag
    gran = plan["granule"]                                          # a job word counts tiles in granules (8-bit fields)
    assert gran in (1, 2) and sub * nblocks <= 255 * gran and nblocks <= 255 and (gran == 1 or sub == 2), tag
    if plan["contig"]:
        assert plan["grid_x"] % 8 == 0 and plan["grid_x"] >= B * H * plan["wgs_per_bh"] and plan["grid_y"] == 1, tag
    else:
        assert plan["grid_x"] == plan["wgs_per_bh"] and plan["grid_y"] == B * H, tag
    cov = {b: [] for b in range(nblocks)}
    for j, wg in enumerate(plan["workgroups"]):
        assert len(wg) == W, tag
        any_cut = False
        for u, jobs in enumerate(wg):
            assert len(jobs) <= J, (tag, j, u)
            tiles = sum(c for _, _, c, _, _ in jobs)
            if W == 8 and u % 2 == 0:
                tiles += sum(c for _, _, c, _, _ in wg[u + 1])
            if W == 4 or u % 2 == 0:
                assert tiles <= plan["tiles_per_simd"], (tag, j, u)
            later = first = 0
            for blk, lo, cnt, rank, pieces in jobs:
                assert 0 <= blk < nblocks and cnt >= 1 and lo + cnt <= sub * (blk + 1), (tag, j, u, blk, lo, cnt)
                assert lo % gran == 0 and cnt % gran == 0 and cnt // gran <= 255 and lo // gran <= 255, (tag, j, u, blk, lo, cnt)
                assert 1 <= pieces <= W and rank < pieces, (tag, j, u)
                cov[blk].append((lo, cnt, rank, pieces, j, u))
                if pieces > 1:
                    any_cut = True
                    later += rank > 0
                    first += rank == 0
            assert later <= 1 and first <= 1, (tag, j, u)
        assert all(m == any_cut for m in plan["merges"][j]), (tag, j)
        for words in plan["raw_words"][j]:
            counts = [(x >> 16) & 255 for x in words]
            assert counts == sorted(counts, key=lambda c: c == 0), (tag, j)      # no job behind an empty word
    for blk, ps in cov.items():
        ps.sort()
        end = 0
        for k, (lo, cnt, rank, pieces, j, u) in enumerate(ps):
            assert lo == end and rank == k and pieces == len(ps), (tag, blk, ps)
            assert j == ps[0][4] and u == ps[0][5] + k, (tag, blk, ps)
            end = lo + cnt
        assert end == sub * (blk + 1), (tag, blk, ps)
    return plan


def check_list_plan(plan, B, T, H, cus):
    """Job lists (one-wave workgroups that run a list of whole blocks of several (b,h)s): every period of `period_bh`
    consecutive (b,h)s uses the same `lists_per_period` lists; per period every block n = 1..nqb of every (b,h) of it on
    exactly one list; at most 16 jobs per list; the grid (8, lists, ceil(periods / 8)) covers every period; a short last
    period keeps every block of the (b,h)s it has (the kernel drops the entries of the others)."""
    tag = (B, T, H, cus)
    nqb, bh = (T + 31) // 32, B * H
    p, q = plan["period_bh"], plan["lists_per_period"]
    assert plan["waves"] == 1 and plan["nqb"] == nqb and plan["workspace_256"] == 0 and plan["parts_per_bh"] == 0, tag
    assert 1 <= p <= 255 and 1 <= q <= 64 and len(plan["lists"]) == q, tag
    periods = (bh + p - 1) // p
    assert plan["grid_x"] == 8 and plan["grid_y"] == q and plan["grid_z"] == (periods + 7) // 8 <= 65535, tag
    seen = {}
    for l, jobs in enumerate(plan["lists"]):
        assert 1 <= len(jobs) <= 16 == len(plan["raw_words"][l]) * 2, (tag, l)
        assert len(jobs) <= plan["jobs_per_list"] and sum(n for n, _ in jobs) <= plan["units_per_list"], (tag, l)
        for n, b in jobs:
            assert 1 <= n <= nqb and 0 <= b < p and (b, n) not in seen, (tag, l, n, b)
            seen[(b, n)] = l
        es = [(w >> (16 * h)) & 0xffff for w in plan["raw_words"][l] for h in (0, 1)]
        assert all(e == 0 for e in es[len(jobs):]), (tag, l)                     # packed at the front, zero behind
    assert len(seen) == p * nqb, tag                                              # every block of every (b,h) of a period
    assert plan["descending_up_to"] in (0, nqb // 2), tag
    # one round (or a whole number of rounds) of equally long waves: the longest list is within two blocks of the mean
    assert plan["units_per_list"] <= p * nqb * (nqb + 1) / 2 / q + nqb, tag
    return plan


def check_plan(B, T, H, with_ws, cus):
    plan = cmhsa.plan_dump(B, T, H, with_ws, cus)
    tag = (B, T, H, with_ws, cus)
    if plan["family"] == "lists":
        return check_list_plan(plan, B, T, H, cus)
    if plan["family"] == "tiles16":
        assert plan["workspace_256"] == 0 and plan["parts_per_bh"] == 0, tag
        return check_tiles16_plan(plan, B, T, H, cus)
    nqb = (T + 31) // 32
    bh = B * H
    waves = plan["waves"]
    assert plan["nqb"] == nqb, tag
    assert waves in (1, 4, 8), tag
    assert len(plan["workgroups"]) == plan["wgs_per_bh"] == plan["grid_y"], tag
    # grid: every (b,h) gets its workgroups; HIP limits
    assert plan["grid_x"] * plan["grid_z"] >= bh, tag
    assert plan["grid_y"] <= 65535 and plan["grid_z"] <= 65535 and plan["grid_x"] * waves * 64 < 2 ** 32, tag
    if plan["longest_first"]:
        assert waves == 1 and plan["grid_x"] % 8 == 0 and plan["grid_x"] == plan["longest_first"], tag
    else:
        assert plan["grid_x"] * plan["grid_z"] == bh, tag       # only the longest-first grid is padded
    if plan["use_table"]:
        assert plan["segs"] in (4, 8) and plan["wgs_per_bh"] * plan["segs"] <= 256 and nqb <= 255, tag   # the table holds 512 words
    if not with_ws:
        assert plan["parts_per_bh"] == 0 and plan["splits_per_bh"] == 0 and plan["workspace_256"] == 0, tag
    assert plan["workspace_256"] * 256 <= WORKSPACE_BYTES, tag
    if plan["parts_per_bh"]:
        assert plan["workspace_256"] * 256 == 16384 + bh * plan["parts_per_bh"] * (2048 + 64) * 4, tag
        assert bh * plan["splits_per_bh"] * 4 <= 16384, tag       # one u32 arrival counter per split block

    covered = {n: [] for n in range(1, nqb + 1)}                 # block -> [(lo, cnt, part word)]
    for j, wg in enumerate(plan["workgroups"]):
        assert 1 <= len(wg) <= plan["segs"], (tag, j)
        used = 0
        simd = [0, 0, 0, 0]
        for n, lo, cnt, mask, x in wg:
            assert 1 <= n <= nqb and cnt >= 1 and lo >= 0 and lo + cnt <= n, (tag, j, n, lo, cnt)
            covered[n].append((lo, cnt, x))
            if waves == 1:
                assert lo == 0 and cnt == n and x == 0, (tag, j)   # one-wave workgroups run whole blocks, one after the other
                continue
            assert mask and mask < (1 << waves) and not (mask & used), (tag, j, hex(mask), hex(used))
            used |= mask
            g = bin(mask).count("1")
            assert g <= cnt, (tag, j, n, cnt, hex(mask))            # every wave of the segment owns at least one unit
            rank = 0
            for w in range(waves):
                if (mask >> w) & 1:
                    simd[w % 4] += cnt // g + (1 if rank < cnt % g else 0)
                    rank += 1
            if x:
                assert with_ws and plan["use_table"] and cnt < n, (tag, j)
        if plan["simd_cap"]:
            assert max(simd) <= plan["simd_cap"], (tag, j, simd)
        if waves == 1:
            assert len(wg) <= 2, (tag, j)

    slots_seen, ids_seen = set(), {}
    for n, pieces in covered.items():
        pieces.sort()
        assert pieces[0][0] == 0, (tag, n, pieces)
        end = 0
        for lo, cnt, x in pieces:
            assert lo == end, (tag, n, pieces)                      # no gap, no overlap
            end = lo + cnt
        assert end == n, (tag, n, pieces)
        if len(pieces) == 1:
            assert pieces[0][2] == 0, (tag, n)
            continue
        nparts = len(pieces)
        assert nparts <= 4, (tag, n)
        first = {x & 255 for _, _, x in pieces}
        ids = {x >> 16 for _, _, x in pieces}
        assert len(first) == 1 and len(ids) == 1, (tag, n, pieces)  # same first slot, same split id on every part
        assert all(((x >> 12) & 15) == nparts for _, _, x in pieces), (tag, n)
        assert sorted((x >> 8) & 15 for _, _, x in pieces) == list(range(nparts)), (tag, n)   # part order = key order
        f, sid = first.pop(), ids.pop()
        assert sid < plan["splits_per_bh"] and sid not in ids_seen, (tag, n)
        ids_seen[sid] = n
        for q in range(nparts):
            assert f + q < plan["parts_per_bh"] and f + q not in slots_seen, (tag, n)
            slots_seen.add(f + q)
    return plan


SHAPES_T = (1, 31, 32, 33, 64, 100, 255, 256, 500, 512, 777, 1000, 1024, 1536, 2048, 4000, 4096, 8160)


@pytest.mark.parametrize("with_ws", [False, True])
def test_every_plan_covers_every_unit_once(with_ws):
    families = set()
    count = 0
    for B, H, T in itertools.product((1, 2, 3, 5, 8, 16, 64), (1, 3, 12, 16, 25), SHAPES_T):
        if B * H * T > 64 * 25 * 1024:
            continue
        plan = check_plan(B, T, H, with_ws, 256)
        count += 1
        if plan["family"] in ("tiles16", "lists"):
            families.add(plan["family"])
            continue
        families.add((plan["waves"], plan["use_table"], bool(plan["longest_first"]), bool(plan["parts_per_bh"])))
    assert count > 400
    # the sweep reaches every family: analytic 4 / 8 waves, tables, split tables (with a workspace), both one-wave forms
    assert {(4, 0, False, False), (8, 0, False, False), (1, 0, False, False), (1, 0, True, False)} <= families
    assert "tiles16" in families and "lists" in families
    families -= {"tiles16", "lists"}
    assert any(t for _, t, _, _ in families)
    if with_ws:
        assert any(p for _, _, _, p in families)


@pytest.mark.parametrize("cus", [64, 104, 228, 256, 304])
def test_plans_for_other_cu_counts(cus):
    """The planner takes the CU count of the device it runs on; nothing in it may assume 256."""
    for B, H, T in itertools.product((1, 4, 32), (1, 12, 25), (100, 512, 1024, 2500)):
        for with_ws in (False, True):
            check_plan(B, T, H, with_ws, cus)


def test_long_sequences_and_large_batches():
    for B, H, T in ((1, 1, 16384), (1, 3, 12345), (1, 1, 33000), (1, 2, 200000), (512, 16, 64), (4096, 4, 40), (32, 16, 4096), (1, 25, 8192)):
        for with_ws in (False, True):
            check_plan(B, T, H, with_ws, 256)


@pytest.mark.parametrize("env", [{"CMHSA_T16_SUB": "1", "CMHSA_T16_WAVES": "4"}, {"CMHSA_T16_SUB": "1", "CMHSA_T16_WAVES": "8"},
                                 {"CMHSA_T16_SUB": "2", "CMHSA_T16_WAVES": "4"}, {"CMHSA_T16_SUB": "2", "CMHSA_T16_WAVES": "8"}])
def test_forced_tiles16_plans_cover_every_tile_once(env):
    """The planner picks the 16-key-tile family only where it prices best; the plan-forcing TEST build of the library
    (tests/helpers/libcmhsa_hip_planenv.so, CMHSA_T16=1) applies it wherever it fits, in each of its four shapes."""
    import subprocess

    code = r"""
import sys, itertools
sys.path[:0] = [%r, %r]
import cmhsa
cmhsa.LIB_PATH = %r
import test_plan
n = 0
for B, H, T in itertools.product((1, 2, 3, 5, 8, 16), (1, 3, 12, 16, 25), (1, 15, 16, 17, 31, 33, 64, 100, 255, 256, 500, 512, 777, 1000, 1024, 1536, 2048, 4000, 4064, 4097, 6000, 8160)):
    for cus in (256, 104):
        plan = cmhsa.plan_dump(B, T, H, False, cus)
        if plan["family"] == "tiles16":
            test_plan.check_tiles16_plan(plan, B, T, H, cus)
            n += 1
print("CHECKED", n)
assert n > 300
""" % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "llm-inference_amd"), os.path.join(ROOT, "tests", "helpers", "libcmhsa_hip_planenv.so"))
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "CMHSA_T16": "1", **env}, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_baseline_configs_pick_the_documented_families():
    small = cmhsa.plan_dump(1, 256, 12, True, 256)                 # the B = 1 launches of a generation (config 3)
    assert small["family"] == "tiles16" and small["sub"] == 1 and small["waves"] == 4
    c2 = cmhsa.plan_dump(1, 1024, 12, True, 256)
    assert c2["waves"] == 8 and c2["use_table"] and c2["splits_per_bh"] >= 1 and c2["simd_cap"] == 7 and c2["wgs_per_bh"] == 21
    c2_plain = cmhsa.plan_dump(1, 1024, 12, False, 256)
    assert c2_plain["parts_per_bh"] == 0
    for B, H in ((32, 2), (32, 16)):                              # config 5: one GPU's shard, and whole
        c5 = cmhsa.plan_dump(B, 4096, H, True, 256)
        assert c5["waves"] == 1 and not c5["longest_first"] and c5["wgs_per_bh"] == 64
    c4 = cmhsa.plan_dump(64, 512, 25, True, 256)                   # config 4's attention shape on one GPU: one round of job lists
    assert c4["family"] == "lists" and c4["period_bh"] == 25 and c4["lists_per_period"] == 32 and c4["units_per_list"] <= 110
    for T, gran in ((3072, 1), (4096, 2), (6144, 2)):              # single long prompts: 32-row blocks on 16-key tiles, job lists
        lp = cmhsa.plan_dump(1, T, 12, True, 256)
        assert lp["family"] == "tiles16" and lp["sub"] == 2 and lp["waves"] == 8 and lp["granule"] == gran and lp["wgs_per_bh"] == 21
    mid = cmhsa.plan_dump(16, 1024, 12, True, 256)                 # 1.5 rounds of pairs -> longest first
    assert mid["waves"] == 1 and mid["longest_first"] == 192


def test_dump_argument_checks():
    lib = cmhsa.load_library()
    assert lib.cmhsa_plan_dump(0, 128, 4, 0, 256, None, 0) == cmhsa.CMHSA_ERR_BAD_SHAPE
    assert lib.cmhsa_plan_dump(1, 128, 0, 0, 256, None, 0) == cmhsa.CMHSA_ERR_BAD_SHAPE
    need = lib.cmhsa_plan_dump(1, 128, 4, 0, 256, None, 0)
    assert need > 16
    import ctypes
    small = (ctypes.c_int * 4)(7, 7, 7, 7)
    assert lib.cmhsa_plan_dump(1, 128, 4, 0, 256, small, 4) == need and list(small) == [7, 7, 7, 7]   # too small: untouched
