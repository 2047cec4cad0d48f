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


def check_plan(B, T, H, with_ws, cus):
    plan = cmhsa.plan_dump(B, T, H, with_ws, cus)
    tag = (B, T, H, with_ws, cus)
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
        families.add((plan["waves"], plan["use_table"], bool(plan["longest_first"]), bool(plan["parts_per_bh"])))
        count += 1
    assert count > 400
    # the sweep reaches every family: analytic 4 / 8 waves, tables, split tables (with a workspace), both one-wave forms
    assert {(4, 0, False, False), (8, 0, False, False), (1, 0, False, False), (1, 0, True, False)} <= families
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


def test_baseline_configs_pick_the_documented_families():
    c2 = cmhsa.plan_dump(1, 1024, 12, True, 256)
    assert c2["waves"] == 8 and c2["use_table"] and c2["splits_per_bh"] >= 1 and c2["simd_cap"] == 7 and c2["wgs_per_bh"] == 21
    c2_plain = cmhsa.plan_dump(1, 1024, 12, False, 256)
    assert c2_plain["parts_per_bh"] == 0
    for B, H in ((32, 2), (32, 16)):                              # config 5: one GPU's shard, and whole
        c5 = cmhsa.plan_dump(B, 4096, H, True, 256)
        assert c5["waves"] == 1 and not c5["longest_first"] and c5["wgs_per_bh"] == 64
    c4 = cmhsa.plan_dump(64, 512, 25, True, 256)                   # config 4's attention shape on one GPU
    assert c4["waves"] == 1 and not c4["longest_first"]
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
