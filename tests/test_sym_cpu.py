"""CPU checks of the combinatorics behind two device-side index maps (host mirrors in
tests/shard_model.py; the device arithmetic is in csrc/mapn_sym.hip and
csrc/mapn_kernels.hip): the symmetric kernel's meeting schedule must cover every unordered pair of
blocks exactly once, its reduce kernel must read exactly the rows that were written, and flow
mode's row rotation must start on the rank's own slice without changing which chunk a row names."""
import itertools

import numpy as np

import pytest

import shard_model as shard


@pytest.mark.parametrize("nb", list(range(1, 20)) + [128, 129, 512])
def test_symmetric_schedule_covers_every_unordered_block_pair_once(nb):
    seen = {}
    written = {b: set() for b in range(nb)}
    for a, b, d, symmetric in shard.sym_meetings(nb):
        if symmetric:
            key = frozenset((a, b))
            assert a != b and key not in seen, (a, b, d)
            seen[key] = d
            assert (d - 1) not in written[b]                  # one row per (partner block, distance)
            written[b].add(d - 1)
        else:
            assert a == b and d == 0
    assert len(seen) == nb * (nb - 1) // 2                    # every unordered pair of distinct blocks
    assert set(seen) == {frozenset(p) for p in itertools.combinations(range(nb), 2)}
    for b in range(nb):                                       # the reduce kernel reads exactly what was written
        assert sorted(written[b]) == shard.sym_reaction_rows(nb, b), (nb, b)


@pytest.mark.parametrize("nb", [2, 3, 8, 128])
def test_symmetric_schedule_is_balanced(nb):
    per_block = {}
    for a, _, d, _ in shard.sym_meetings(nb):
        per_block[a] = per_block.get(a, 0) + 1
    assert max(per_block.values()) - min(per_block.values()) <= 1     # the half-ring partner: one extra meeting group


@pytest.mark.parametrize("nb,world", [(64, 8), (64, 4), (64, 2), (16, 8), (256, 8), (1024, 8), (24, 3), (48, 6)])
def test_half_ring_group_is_spread_evenly_over_the_ranks(nb, world):
    """which block of a half-ring pair (p, p + nb/2) runs its meetings ALTERNATES with p, so every rank of a sharded
    job runs the same number of meetings (65 536 / 8: 520 per block on average; until round 3 ranks 0 .. 3 ran 528 and ranks 4 .. 7 512 and every
    step waited for the heavy half) -- and each pair is still run by exactly one of its two blocks."""
    nbl, half = nb // world, nb // 2
    per_rank = [0] * world
    for a, b, d, symmetric in shard.sym_meetings(nb):
        per_rank[a // nbl] += 16                              # 16 J-blocks per (block, partner) meeting group
    assert max(per_rank) - min(per_rank) <= (16 if nbl % 2 else 0), per_rank
    if (nb, world) == (64, 8):
        assert per_rank == [8 * 520] * 8                      # 8 blocks a rank, 520 meetings a block on average (528 or 512 each)
    for p in range(half):
        assert shard.sym_runs_half(nb, p) != shard.sym_runs_half(nb, p + half)
        assert shard.sym_runs_half(nb, p) == (p % 2 == 0)
    assert not any(shard.sym_runs_half(nb + 1, a) for a in range(nb + 1))   # odd block counts have no half-ring group


@pytest.mark.parametrize("n,world,waves,sb", [(65536, 8, 16, 16), (65536, 2, 8, 8), (4096, 8, 4, 1), (4096, 4, 8, 2),
                                              (1048576, 8, 8, 8), (3000 * 4, 4, 8, 3), (8192, 2, 16, 4)])
def test_flow_rotation_starts_on_the_own_slice_and_is_a_permutation(n, world, waves, sb):
    tiles = (n + 63) // 64
    for rank in range(world):
        first, count = shard.shard_range(n, rank, world)
        rot = shard.flow_row_rotation(n, first, waves, sb)
        assert 0 <= rot < sb
        logical = [(y + rot) % sb for y in range(sb)]
        assert sorted(logical) == list(range(sb))             # a permutation of the rows: every chunk still computed once
        # the first dispatched row's chunks contain the first tile of the own slice
        t_lo = shard.chunk_tiles(tiles, waves * sb, logical[0] * waves)[0]
        t_hi = shard.chunk_tiles(tiles, waves * sb, logical[0] * waves + waves - 1)[1]
        if tiles >= waves * sb:
            assert t_lo <= first // 64 < max(t_hi, t_lo + 1), (rank, rot, t_lo, t_hi)


def test_chunk_tiles_partition_the_range():
    for tiles, splits in ((1024, 64), (1024, 256), (47, 64), (100, 7), (1, 1)):
        edges = [shard.chunk_tiles(tiles, splits, c) for c in range(splits)]
        assert edges[0][0] == 0 and edges[-1][1] == tiles
        assert all(edges[c][1] == edges[c + 1][0] for c in range(splits - 1))


PLAN_SHAPES = [  # nb, groups per window, parts, taper1, taper2, waves
    (64, 0, 32, None, 0, 4), (64, 0, 38, 28, 4, 4), (64, 0, 64, None, 0, 4), (63, 0, 32, None, 0, 4), (8, 0, 8, None, 0, 4), (8, 1, 4, None, 0, 4),
    (8, 2, 4, None, 0, 4), (9, 2, 4, None, 0, 4), (2, 0, 4, None, 0, 4), (1, 0, 4, None, 0, 4), (3, 0, 2, None, 0, 8), (64, 5, 8, None, 0, 8),
    (256, 0, 32, None, 0, 4), (100, 0, 36, 28, 4, 4), (1024, 57, 8, None, 0, 4), (4096, 16, 2, None, 0, 4)]


@pytest.mark.parametrize("nb,gpw,parts,t1,t2,waves", PLAN_SHAPES)
def test_plan_windows_partition_the_groups_and_waves_carry_equal_cost(nb, gpw, parts, t1, t2, waves):
    """The host-built plan (csrc/mapn_sym_plan.cpp) the kernels read: the windows partition the meeting groups, every wave of a
    launch runs at least 64 steps (so a meeting is cut at most once) and the same COST to within two steps (a step of a block
    against itself and a symmetric step cost the same: the kernel runs one loop for both); tapered parts weigh 4 : 2 : 1."""
    import mapn
    plan = mapn.describe_sym_plan(nb, gpw, parts, t1, t2, waves)
    D, half = (nb - 1) // 2, (nb // 2 if nb % 2 == 0 else 0)
    groups = 1 + D + (1 if half else 0)
    assert plan.groups == groups and plan.nb == nb
    edges = [int(w[0]) for w in plan.windows] + [int(plan.windows[-1][1])]
    assert edges[0] == 0 and edges[-1] == groups
    for k, w in enumerate(plan.windows):
        g0, g1, m0, m1 = (int(x) for x in w)
        assert g1 > g0 and (k == 0 or g0 == int(plan.windows[k - 1][1]))
        assert g1 - max(g0, 1) <= plan.brows and (gpw == 0 or g1 - max(g0, 1) <= gpw)
        has_half = bool(half) and g1 == groups
        assert m0 == 16 * (g1 - g0) and m1 == 16 * (g1 - g0 - (1 if has_half else 0))
        for cls, M in ((0, m0), (1, m1)):
            b = plan.bounds(k, cls).astype(np.int64)
            assert b[0] == 0 and b[-1] == 64 * M and (np.diff(b) >= (64 if M else 0)).all()
            if not M:
                continue
            self_steps = 1024 if g0 == 0 else 0
            cost = np.where(b <= self_steps, 6 * b, 6 * self_steps + 6 * (b - self_steps))   # (csrc/mapn_sym_plan.h: SYM_COST_SELF, SYM_COST_SYM)
            per_wave = np.diff(cost).reshape(parts, waves)
            unit = [4 if s < (parts if t1 is None else t1) else 2 if s < (parts if t1 is None else t1) + t2 else 1 for s in range(parts)]
            norm = per_wave / np.array(unit)[:, None]
            assert norm.max() - norm.min() <= 12 + 1e-9, (norm.min(), norm.max())   # bounds are whole steps: two steps of slack
            assert (per_wave.max(axis=1) - per_wave.min(axis=1) <= 12).all()     # the waves of a workgroup leave together


def test_plan_that_would_leave_a_wave_fewer_than_64_steps_is_refused():
    import mapn
    with pytest.raises(mapn.MapnError, match="< 64"):
        mapn.describe_sym_plan(8, 0, 20, None, 0, 4)           # 64 meetings for 80 waves
    with pytest.raises(mapn.MapnError, match="< 64"):
        mapn.describe_sym_plan(64, 0, 48, 28, 4, 4)            # 16 parts of one unit: 56 steps for the smallest waves
    with pytest.raises(mapn.MapnError):
        mapn.describe_sym_plan(64, 0, 32, 30, 4, 4)            # taper1 + taper2 > parts


XCD_W = [1024, 970, 1010, 1000, 1020, 985, 1024, 990]      # what a calibration returns: the dies' relative speeds


def test_xcd_weighted_parts_are_sized_by_the_speed_of_the_die_they_run_on():
    """mapn_set_sym_xcd_weights, the SPREAD form (xcd_mode 1; what runs where the class-aware form does not apply, e.g. 63 blocks):
    with weights the plan has 16 table sets (class + 2 * (block mod 8)); part s of a block whose
    index is r mod 8 runs on die (r - s) mod 8 and its waves get steps in proportion to that die's speed; equal weights, or a
    launch that does not cover a multiple of 8 blocks, give the default plan."""
    import mapn
    plan = mapn.describe_sym_plan(64, 0, 32, None, 0, 4, xcd_weights=XCD_W, xcd_mode=1)
    assert plan.sets == 16 and plan.xcd_weight == XCD_W and plan.xcd_mode == 1 and plan.wgmap is None
    assert mapn.describe_sym_plan(72, 0, 30, None, 0, 4, xcd_weights=XCD_W).xcd_mode == 1      # parts not a multiple of 4: spread
    assert mapn.describe_sym_plan(8 * 9 + 8, 0, 32, None, 0, 4, xcd_weights=XCD_W, launch_blocks=8, launch_a0=8).xcd_mode == 2
    base = mapn.describe_sym_plan(64, 0, 32, None, 0, 4)
    assert base.sets == 2 and mapn.describe_sym_plan(64, 0, 32, None, 0, 4, xcd_weights=[1000] * 8).sets == 2
    assert mapn.describe_sym_plan(98, 0, 32, None, 0, 4, xcd_weights=XCD_W).sets == 2          # 100 000 bodies: 98 blocks
    assert mapn.describe_sym_plan(64, 0, 32, None, 0, 4, xcd_weights=XCD_W, launch_blocks=4).sets == 2
    for r in range(8):
        for cls in (0, 1):
            b = plan.bounds(0, plan.set_of(cls, r)).astype(np.int64)
            per_part = np.diff(b).reshape(32, 4).sum(axis=1)
            w = np.array([XCD_W[(r - s) % 8] for s in range(32)], np.float64)
            want = w / w.sum() * b[-1]
            assert np.abs(per_part - want).max() <= 4, (r, cls)       # whole steps: a few steps of slack
            assert b[-1] == np.diff(base.bounds(0, cls).astype(np.int64)).sum()


@pytest.mark.parametrize("nb,parts,waves,bias,blocks,a0", [(64, 4, 8, (10, 3), 0, 0), (64, 32, 8, (2, 1), 8, 0), (64, 32, 8, (2, 1), 8, 32), (128, 4, 8, (10, 3), 0, 0),
                                                             (256, 8, 4, (1, 1), 32, 96), (16, 8, 4, (1, 1), 0, 0)])
def test_class_aware_xcd_weights_put_the_heavy_blocks_on_the_fast_dies(nb, parts, waves, bias, blocks, a0):
    """Round 4: the blocks that run the half-ring group (class 0) carry one group more than the others -- 3.1 % at 65 536 bodies --
    and in a one-round launch nothing hides it; the dies' speeds differ by about as much.  With XCD weights the plan therefore maps
    every class-0 block's parts onto four of the dies and every class-1 block's onto the other four -- the 4 : 4 split whose speed
    ratio best matches the classes' work ratio: the faster four for the heavy blocks at 65 536 bodies -- a quarter of the
    block's parts per die, each sized by its die's speed.  The workgroup map is a bijection onto (block, part); workgroup (x, y)
    lands on die x mod 8; the tables stay per class."""
    import mapn
    import shard_model as shard
    plan = mapn.describe_sym_plan(nb, 0, parts, None, 0, waves, xcd_weights=XCD_W, wave_bias=bias, launch_blocks=blocks, launch_a0=a0)
    B = blocks or nb
    assert plan.xcd_mode == 2 and plan.sets == 2 and plan.wgmap.shape == (parts, B, 2)
    # the 4 : 4 split of the dies whose speed ratio best matches the classes' work ratio (groups D + 2 : D + 1), each class's dies fastest first
    import itertools
    D = (nb - 1) // 2
    cost = lambda A: max((D + 2) / sum(XCD_W[d] for d in A), (D + 1) / sum(XCD_W[d] for d in range(8) if d not in A))
    best = min(itertools.combinations(range(8), 4), key=lambda A: (round(cost(A), 12), sum(1 << d for d in A)))
    assert abs(cost(plan.class_die[0]) - cost(best)) < 1e-12 and sorted(plan.class_die[0] + plan.class_die[1]) == list(range(8))
    for c in (0, 1):
        assert [XCD_W[d] for d in plan.class_die[c]] == sorted((XCD_W[d] for d in plan.class_die[c]), reverse=True)
    seen = set()
    for y in range(parts):
        for x in range(B):
            la, s = (int(v) for v in plan.wgmap[y, x])
            cls = shard.sym_block_class(nb, a0 + la)
            assert (la, s) not in seen and la < B and s < parts
            seen.add((la, s))
            assert x % 8 == plan.class_die[cls][s % 4]             # part s of a class-c block runs on the (s mod 4)-th die of class c
    assert len(seen) == B * parts
    base = mapn.describe_sym_plan(nb, 0, parts, None, 0, waves, wave_bias=bias, launch_blocks=blocks, launch_a0=a0)
    for cls in (0, 1):
        b = plan.bounds(0, cls).astype(np.int64)
        assert b[0] == 0 and b[-1] == base.bounds(0, cls)[-1]      # the same steps in total, dealt differently
        per_part = np.diff(b).reshape(parts, waves).sum(axis=1)
        w = np.array([XCD_W[plan.class_die[cls][s % 4]] for s in range(parts)], np.float64)
        assert np.abs(per_part - w / w.sum() * b[-1]).max() <= 2 * waves, (cls, per_part)
    # what it is for: time of a part = steps / speed of its die -- the slowest workgroup of the launch is earlier than with either
    # the default plan (every die holds heavy blocks) or the spread form (which equalises the dies but not the classes)
    def worst(pl, die_of):
        t = 0.0
        for la in range(B):
            cls = shard.sym_block_class(nb, a0 + la)
            b = pl.bounds(0, pl.set_of(cls, la)).astype(np.int64)
            per_part = np.diff(b).reshape(parts, waves).sum(axis=1)
            t = max(t, max(per_part[s] / XCD_W[die_of(pl, la, s, cls)] for s in range(parts)))
        return t
    t_class = worst(plan, lambda pl, la, s, cls: pl.class_die[cls][s % 4])
    t_default = worst(base, lambda pl, la, s, cls: la % 8)
    spread = mapn.describe_sym_plan(nb, 0, parts, None, 0, waves, xcd_weights=XCD_W, wave_bias=bias, launch_blocks=blocks, launch_a0=a0, xcd_mode=1)
    t_spread = worst(spread, lambda pl, la, s, cls: (la - s) % 8)
    assert t_class < t_default and t_class <= t_spread * 1.001, (t_class, t_spread, t_default)


def test_default_sharded_plan_puts_the_heavy_blocks_on_the_odd_dispatch_slots():
    """Round 4: a rank's launch holds one block per die (block x on dispatch slot x mod 8) and its blocks with the half-ring group
    alternate with the others -- all on the even slots for the ranks of the ring's first half, all on the odd ones for the second.
    The odd slots are the faster dies (2 - 3 % on every box measured), so the first-half ranks ran 6 % behind in their heavy blocks
    and were the slower ranks.  Without XCD weights the plan now flips block x <-> x ^ 1 where that puts the heavy blocks on the odd
    slots; unsharded launches (every die holds both classes) and weighted plans (the workgroup map decides) are left alone."""
    import mapn
    import shard_model as shard
    for rank in range(8):
        pl = mapn.describe_sym_plan(64, 0, 32, None, 0, 8, wave_bias=(3, 1), launch_blocks=8, launch_a0=8 * rank)
        heavy_slots = {(la ^ pl.la_flip) % 8 for la in range(8) if shard.sym_block_class(64, 8 * rank + la) == 0}
        assert heavy_slots == {1, 3, 5, 7}, (rank, pl.la_flip, heavy_slots)
        assert pl.la_flip == (1 if rank < 4 else 0)
    assert mapn.describe_sym_plan(64, 0, 4, None, 0, 8, wave_bias=(10, 3)).la_flip == 0                        # unsharded
    assert mapn.describe_sym_plan(27, 0, 32, None, 0, 4, launch_blocks=9).la_flip == 0                         # odd block count: no half-ring group, no classes
    assert mapn.describe_sym_plan(64, 0, 32, None, 0, 8, wave_bias=(2, 1), xcd_weights=XCD_W, launch_blocks=8).la_flip == 0   # weights: the workgroup map decides


def test_wave_bias_gives_the_older_waves_of_a_workgroup_the_larger_share():
    """An 8-wave workgroup's waves 0 .. 3 are the older wave of their SIMDs and are issued first (measured); with a wave bias
    hi : lo they carry hi / lo times the steps of waves 4 .. 7, so that the two waves of a SIMD finish together.  Still one
    linear run of steps per wave, every wave at least 64 steps; a bias with an odd wave count or outside 1 .. 64 is refused."""
    import mapn
    base = mapn.describe_sym_plan(64, 0, 32, None, 0, 8, launch_blocks=8)
    assert base.wave_bias == (1, 1)
    plan = mapn.describe_sym_plan(64, 0, 32, None, 0, 8, launch_blocks=8, wave_bias=(3, 1))
    assert plan.wave_bias == (3, 1) and plan.sets == 2
    for cls in (0, 1):
        b = plan.bounds(0, cls).astype(np.int64)
        assert b[0] == 0 and b[-1] == base.bounds(0, cls)[-1]
        per = np.diff(b).reshape(32, 8)
        assert (per >= 64).all()
        assert np.abs(per[:, :4] - 3 * per[:, 4:].mean()).max() <= 4 and np.ptp(per[:, 4:]) <= 2 and np.ptp(per[:, :4]) <= 2
        assert np.ptp(per.sum(axis=1)) <= 2                                     # the workgroups still carry equal shares
    with pytest.raises(mapn.MapnError, match="< 64"):
        mapn.describe_sym_plan(64, 0, 32, None, 0, 8, wave_bias=(5, 1))        # 1056 steps per workgroup: 44 for the small waves
    with pytest.raises(mapn.MapnError):
        mapn.describe_sym_plan(64, 0, 32, None, 0, 8, wave_bias=(65, 1))
    with pytest.raises(mapn.MapnError):
        mapn.describe_sym_plan(64, 0, 32, None, 0, 8, wave_bias=(0, 1))


@pytest.mark.parametrize("nb,gpw,parts,t1,t2,waves,xw,bias", [sh + (None, (1, 1)) for sh in PLAN_SHAPES[:13]] +
                         [(8, 0, 8, None, 0, 4, XCD_W, (1, 1)), (16, 3, 4, None, 0, 4, XCD_W, (1, 1)), (64, 0, 36, 28, 8, 4, XCD_W, (1, 1)),
                          (64, 0, 32, None, 0, 8, None, (3, 1)), (16, 0, 4, None, 0, 8, XCD_W, (2, 1)), (9, 2, 2, None, 0, 8, None, (1, 3))])
def test_force_kernel_writes_exactly_the_rows_the_reduce_kernel_reads(nb, gpw, parts, t1, t2, waves, xw, bias):
    """force_sym_kernel's bookkeeping replayed from the plan tables: every step of every meeting is run exactly once; a symmetric
    meeting's row is written exactly once (whole, or put together in LDS from two waves of one workgroup, or its first steps when
    it is cut between two workgroups -- then its last steps go to the later workgroup's head row); sym_reduce_integrate_kernel
    reads a meeting's row and, where the split table says so, that very head row -- and nothing else is ever written."""
    import mapn
    plan = mapn.describe_sym_plan(nb, gpw, parts, t1, t2, waves, xcd_weights=xw, wave_bias=bias, xcd_mode=1)   # (the spread form: its 16 table sets are what this test walks)
    assert plan.sets == (16 if xw else 2) and plan.wave_bias == bias
    D, half = (nb - 1) // 2, (nb // 2 if nb % 2 == 0 else 0)
    for k, w in enumerate(plan.windows):
        g0, g1 = int(w[0]), int(w[1])
        rows, heads = {}, {}                                   # (J-block, group) -> writes; (I-block, part) -> meeting it holds
        for a in range(nb if nb <= 16 else 3):                 # every block for small jobs, else a sample holding both classes
            a = a if nb <= 16 else (0, half - 1 if half else 1, nb - 1)[a]
            cls = shard.sym_block_class(nb, a)
            st = plan.set_of(cls, a)
            M = int(w[2 + cls])
            seen = np.zeros((max(M, 1), 64), np.int32)
            lds = {}                                           # (part, wave, which) -> meeting
            for s, wv, m, k0, n in shard.sym_wave_pieces(plan, k, st):
                seen[m, k0:k0 + n] += 1
                g, t = g0 + m // 16, m % 16
                if g == 0:
                    continue
                d = g if g <= D else half
                key = (((a + d) % nb) * 16 + t, g)
                if n == 64:
                    rows[key] = rows.get(key, 0) + 1
                elif k0:                                       # the meeting's last steps
                    assert k0 + n == 64
                    if wv == 0:
                        assert (a, s) not in heads
                        heads[(a, s)] = key
                    else:
                        assert lds.pop((s, wv - 1, 0)) == key  # the previous wave of this workgroup ran its first steps
                        rows[key] = rows.get(key, 0) + 1       # put together after the barrier
                else:                                          # the meeting's first steps
                    if wv == waves - 1:
                        rows[key] = rows.get(key, 0) + 1
                    else:
                        lds[(s, wv, 0)] = key
            assert not lds and (seen[:M] == 1).all()
            # what the reduce kernel reads for the rows this block wrote: the split table names the head row
            sp = plan.split(k, st)
            for m in range(M):
                g, t = g0 + m // 16, m % 16
                if g == 0:
                    assert sp[m] == plan.SPLIT_NONE
                    continue
                d = g if g <= D else half
                key = (((a + d) % nb) * 16 + t, g)
                assert rows.get(key) == 1, (a, m)
                if sp[m] != plan.SPLIT_NONE:
                    assert heads.pop((a, int(sp[m]))) == key
            assert not [h for h in heads if h[0] == a]


@pytest.mark.parametrize("nb,world", [(8, 2), (8, 4), (8, 8), (9, 3), (6, 2), (16, 8), (64, 8), (64, 2), (1024, 8)])
def test_sharded_symmetric_step_sender_and_receiver_sets_agree(nb, world):
    """Gather algorithm 4: rank r stores a reaction row into rank q's receive region exactly when q
    waits for one from r, and together the rows cover every symmetric meeting of the job."""
    masks = [shard.sym_shard_masks(nb, world, r) for r in range(world)]
    for r in range(world):
        for q in range(world):
            assert bool(masks[r][0] >> q & 1) == bool(masks[q][1] >> r & 1)
    nbl = nb // world
    need = {(a // nbl, b // nbl) for a, b, d, sym in shard.sym_meetings(nb) if sym}
    have = {(r, q) for r in range(world) for q in range(world) if masks[r][0] >> q & 1}
    assert need == have
    if world == 8 and nb >= 16:
        assert all(bin(m[0]).count("1") == 5 for m in masks)      # this rank and the four ahead of it on the ring


@pytest.mark.parametrize("nb,world", [(4, 2), (6, 2), (6, 3), (8, 4), (8, 8), (9, 3), (16, 8)])
def test_sharded_symmetric_decomposition_reproduces_the_all_pairs_forces(nb, world):
    """Gather algorithm 4 on paper (float64, tiny blocks): every rank evaluates the meetings of its own blocks
    once, keeps the forces on its bodies and ships the reactions, summed per destination rank, to the owners.
    Own rows + received rows must equal the direct all-pairs sum for every body, and rows may only travel
    where the sender / receiver masks say so."""
    B = 6                                                  # bodies per block (the device uses 1024)
    rng = np.random.default_rng(nb * 100 + world)
    n = nb * B
    x = rng.normal(size=(n, 3)) * 100.0
    soft2 = 25.0

    def pair(i, j):                                        # force on i from j (hlsl:44-57 without the mass)
        r = x[j] - x[i]
        return r * (r @ r + soft2) ** -1.5

    direct = np.array([sum(pair(i, j) for j in range(n)) for i in range(n)])
    nbl = nb // world
    own = np.zeros((n, 3))                                 # a-rows: forces a rank computed for its own bodies
    recv = np.zeros((world, world, n, 3))                  # recv[receiver][sender][body]
    for a, b, d, symmetric in shard.sym_meetings(nb):
        r = a // nbl
        for i in range(a * B, (a + 1) * B):
            for j in range(b * B, (b + 1) * B):
                f = pair(i, j)
                own[i] += f
                if symmetric:
                    recv[b // nbl][r][j] -= f              # the reaction, to the owner of block b
    total = own + recv.sum(axis=1).sum(axis=0)
    np.testing.assert_allclose(total, direct, rtol=1e-9, atol=1e-12)
    for q in range(world):
        send, rcv = shard.sym_shard_masks(nb, world, q)
        for r in range(world):
            used = bool(np.abs(recv[q][r]).max() > 0)
            assert used == bool(rcv >> r & 1), (q, r)


def test_plan_builder_is_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """csrc/mapn_sym_plan.cpp is plain C++ (the one part of the product that is): built with g++ -fsanitize=address,undefined together
    with tests/cpp/plan_sanitize.cpp, which sweeps ~11 000 shapes (block counts 1 ... 1024, parts, tapers, wave biases, the three XCD
    modes with skewed weights, sharded launches) through mapn_sym_plan_describe into arrays of EXACTLY the reported size and checks
    that arrays one word short are refused.  (GPU sanitizers are not available on this pool: DESIGN 6.)"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "plan_sanitize")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(root, "include"),
                        "-I" + os.path.join(root, "multi-adapter-particles_amd", "csrc"), os.path.join(root, "tests", "cpp", "plan_sanitize.cpp"),
                        os.path.join(root, "multi-adapter-particles_amd", "csrc", "mapn_sym_plan.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and ("asan" in b.stderr or "ubsan" in b.stderr or "sanitize" in b.stderr):
        pytest.skip("this g++ has no sanitizer runtime: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "plans built" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
