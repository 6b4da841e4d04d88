"""The order-matched restatement of the PARTIALLY ACTIVE step's split form (oracle/mapn_oracle.c, ORDER_MATCHED_SPLIT) against
the oracle proper (reference order, Compute.cpp:1041 semantics) -- on the CPU, with plans computed without a device
(mapn_sym_plan_describe): the frozen bodies' rows + the symmetric plan of the active bodies must add up to the same physics, leave
the frozen tail alone, and respond to the one-sided launch's shape only by rounding."""
import types

import numpy as np
import pytest

import mapn
from oracle import OracleSim, Params


def split_of(n, active, waves, sb):
    return types.SimpleNamespace(active=active, frozen=n - active, frozen_waves=waves, frozen_sb=sb)


@pytest.mark.parametrize("n,num_active,parts,waves,gpw", [(4096, 2100, 4, 4, 0), (6000, 3072, 2, 8, 0), (8192, 4097, 4, 4, 1), (3000, 2048, 2, 4, 0)])
def test_split_restatement_matches_the_reference_order_oracle(oracle, n, num_active, parts, waves, gpw):
    active = oracle.active_bodies(num_active, n)
    assert active < n
    nb = (active + 1023) // 1024
    plan = mapn.compute.describe_sym_plan(nb, groups_per_window=gpw, parts=parts, waves=waves)
    if gpw:
        assert len(plan.windows) > 1
    pos, vel = oracle.initial_state(n, seed=3)
    prm = Params(mass=70000.0 / n)
    ref = OracleSim(oracle, pos, vel, params=prm)
    out = {}
    for fw, fsb in ((8, 2), (4, 1)):
        sim = OracleSim(oracle, pos, vel, params=prm, split_plan=(split_of(n, active, fw, fsb), plan))
        sim.simulate(num_active=num_active, steps=2)
        out[(fw, fsb)] = [p.copy() for p in sim.pos]
    ref.simulate(num_active=num_active, steps=2)
    for key, bufs in out.items():
        for b in (0, 1):
            d = np.linalg.norm(bufs[b][:, :3].astype(np.float64) - ref.pos[b][:, :3], axis=1) / 400.0
            assert d.max() < 1e-6, (key, b, d.max())
            np.testing.assert_array_equal(bufs[b][active:], ref.pos[b][active:])      # the frozen tail: never written
            assert np.abs(bufs[b][:active, 3] - ref.pos[b][:active, 3]).max() <= 1e-4 * ref.pos[b][:active, 3].max()
    a, b = out[(8, 2)][1], out[(4, 1)][1]
    assert np.linalg.norm(a[:, :3].astype(np.float64) - b[:, :3], axis=1).max() / 400.0 < 1e-6      # another cut of the frozen range: rounding only


def test_split_restatement_refuses_shapes_that_do_not_fit(oracle):
    n, active = 4096, 2048
    pos, vel = oracle.initial_state(n, seed=1)
    plan = mapn.compute.describe_sym_plan(3, parts=4, waves=4)          # a plan for 3 blocks: not the 2 blocks of 2048 active bodies
    sim = OracleSim(oracle, pos, vel, params=Params(mass=1.0), split_plan=(split_of(n, active, 8, 1), plan))
    with pytest.raises(AssertionError):
        sim.simulate(num_active=active)


def test_the_cost_model_picks_what_the_sweep_measured(lib):
    """mapn_step_form_describe (product, no device): the form an unsharded all-pairs step takes for (N, num_active), against the measured
    sweep (profiles/r05_partial_active_sweep.txt: one box; where two forms were within 1 % either pick is right and is not listed here).
    0 = one-sided, 1 = full symmetric, 2 = split."""
    ONE, FULL, SPLIT = 0, 1, 2
    table = {65536: {8192: ONE, 24576: SPLIT, 32768: SPLIT, 40960: SPLIT, 49152: SPLIT, 57344: SPLIT, 61440: FULL, 63552: FULL, 65536: FULL},
             262144: {65536: SPLIT, 98304: SPLIT, 131072: SPLIT, 163840: SPLIT, 196608: SPLIT, 229376: SPLIT, 245760: SPLIT, 254272: FULL, 262144: FULL},
             4194304: {2097152: SPLIT, 3145728: SPLIT, 4194304: FULL}}
    for n, rows in table.items():
        for na, want in rows.items():
            assert lib.mapn_step_form_describe(n, na) == want, (n, na)
    # rounding and edges: Compute.cpp:1041 (groups of 64), nothing active, more than N, a job smaller than one block, too few active blocks
    assert lib.mapn_step_form_describe(65536, 32705) == lib.mapn_step_form_describe(65536, 32768) == SPLIT
    assert lib.mapn_step_form_describe(65536, 0) == ONE and lib.mapn_step_form_describe(65536, -3) == ONE
    assert lib.mapn_step_form_describe(65536, 10 ** 9) == FULL and lib.mapn_step_form_describe(1000, 1000) == ONE
    assert lib.mapn_step_form_describe(4096, 3500) == FULL and lib.mapn_step_form_describe(4096, 1000) == ONE      # (tests/test_gpu_sym.py: the small job's forms)
    assert lib.mapn_step_form_describe(0, 5) < 0
    # a pure function: the same answer every time
    assert len({lib.mapn_step_form_describe(100000, 50000) for _ in range(5)}) == 1


@pytest.mark.parametrize("n,world,parts,waves", [(8192, 2, 8, 4), (8192, 4, 8, 4), (9216, 3, 8, 4), (16384, 8, 4, 8), (6144, 2, 4, 4)])
def test_sharded_symmetric_restatement_matches_the_reference_order_oracle(oracle, n, world, parts, waves):
    """ORDER_MATCHED_SHARDED (oracle/mapn_oracle.c): all ranks of the sharded symmetric step restated in one process from per-rank plans
    computed without a device (mapn_sym_plan_describe with the rank's launch_blocks / launch_a0) -- force rows per rank, reactions summed per
    destination, own partial sums + rows received nearest sender first -- against the oracle proper: the same physics to rounding, even and
    odd block counts (the half-ring group alternates between the ranks), one block per rank up to eight."""
    from oracle import step_sym_sharded
    nb, nbl = n // 1024, n // 1024 // world
    plans = [mapn.compute.describe_sym_plan(nb, parts=parts, waves=waves, launch_blocks=nbl, launch_a0=r * nbl) for r in range(world)]
    pos, vel = oracle.initial_state(n, seed=2)
    prm = Params(mass=70000.0 / n)
    p, v = step_sym_sharded(oracle, pos, vel, prm, plans)
    ref = OracleSim(oracle, pos, vel, params=prm); ref.simulate()
    rp, rv = ref.latest
    assert np.linalg.norm(p[:, :3].astype(np.float64) - rp[:, :3], axis=1).max() / 400.0 < 1e-6
    assert np.linalg.norm(v.astype(np.float64) - rv, axis=1).max() / 15.0 < 2e-5
    assert np.abs(p[:, 3] - rp[:, 3]).max() <= 1e-4 * rp[:, 3].max()
    p2, v2 = step_sym_sharded(oracle, pos, vel, prm, plans, threads=3)
    np.testing.assert_array_equal(p, p2); np.testing.assert_array_equal(v, v2)          # fixed-order sums: thread count changes nothing
