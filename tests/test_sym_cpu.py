"""CPU checks of the combinatorics behind two device-side index maps (host mirrors in
multi-adapter-particles_amd/shard.py; the device arithmetic is in csrc/mapn_sym.hip and
csrc/mapn_kernels.hip): the symmetric kernel's meeting schedule must cover every unordered pair of
blocks exactly once, its reduce kernel must read exactly the rows that were written, and flow
mode's row rotation must start on the rank's own slice without changing which chunk a row names."""
import itertools

import numpy as np

import pytest

from mapn import shard


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


@pytest.mark.parametrize("meetings,parts,waves,taper", [(528, 32, 4, None), (512, 32, 4, None), (16, 4, 4, None), (48, 12, 4, None), (2064, 32, 4, None),
                                                        (528, 16, 8, None), (40, 7, 4, None), (3, 2, 4, None), (528, 48, 4, (24, 8)), (512, 48, 4, (24, 8)),
                                                        (272, 48, 4, (24, 8)), (144, 48, 4, (24, 8)), (1040, 40, 4, (28, 4))])
def test_wave_deal_covers_every_step_of_every_meeting_once_and_is_balanced(meetings, parts, waves, taper):
    """force_sym_kernel's deal of an I-block's meetings to waves: whole meetings first, the remainder of a
    part shared step-wise.  Every (meeting, travelling-body offset) is run exactly once, and the waves of
    a workgroup all run the same number of steps (nobody waits at the closing barrier).  With tapered parts
    (4 : 2 : 1 units) the late parts are a quarter of the early ones."""
    t1, t2 = taper if taper else (None, 0)
    seen = np.zeros((meetings, 64), np.int32)
    steps = {}
    for s, w, m, rot, n in shard.sym_wave_items(meetings, parts, waves, t1, t2):
        seen[m, rot:rot + n] += 1
        steps[(s, w)] = steps.get((s, w), 0) + n
    assert (seen == 1).all()
    for s in range(parts):
        per_wave = {steps.get((s, w), 0) for w in range(waves)}
        assert len(per_wave) == 1
    bounds = shard.sym_part_bounds(meetings, parts, t1, t2)
    assert bounds[0] == 0 and bounds[-1] == meetings and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
    sizes = [b1 - b0 for b0, b1 in zip(bounds, bounds[1:])]
    if taper is None:
        assert max(sizes) - min(sizes) <= 1                # equal parts differ by at most one meeting
    else:
        big, small = sizes[:t1], sizes[t1 + t2:]
        assert max(big) - min(big) <= 1 and max(small) - min(small) <= 1 and abs(4 * np.mean(small) - np.mean(big)) <= 2


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
