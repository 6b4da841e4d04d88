"""Oracle vs an independent numpy restatement (bit-exact), vs the float64 model (tolerance) and
vs the committed golden vectors."""
import os

import numpy as np
import pytest

from oracle import MODE_ALL_PAIRS, MODE_CENTRAL_WELL, OracleSim, Params
from oracle import model_np


@pytest.mark.parametrize("n", [64, 200, 512])
def test_c_oracle_matches_numpy_fp32_bit_exact(oracle, n):
    pos, vel = oracle.initial_state(n, seed=5)
    sim = OracleSim(oracle, pos, vel, params=Params(mass=70000.0 / n))
    sim.simulate(num_active=n)
    p, v = model_np.step_fp32_loop(pos, vel, mass=70000.0 / n)
    na = oracle.active_bodies(n, n)
    np.testing.assert_array_equal(sim.latest[0][:na], p[:na])
    np.testing.assert_array_equal(sim.latest[1][:na], v[:na])


def test_central_well_matches_numpy_bit_exact(oracle):
    pos, vel = oracle.initial_state(1024, seed=2)
    sim = OracleSim(oracle, pos, vel, mode=MODE_CENTRAL_WELL)
    sim.simulate()
    p, v = model_np.step_central_well_fp32(pos, vel)
    np.testing.assert_array_equal(sim.latest[0], p)
    np.testing.assert_array_equal(sim.latest[1], v)


def test_thread_count_does_not_change_result(oracle):
    pos, vel = oracle.initial_state(1000, seed=1)
    a = OracleSim(oracle, pos, vel, params=Params(mass=70.0), threads=1); a.simulate()
    b = OracleSim(oracle, pos, vel, params=Params(mass=70.0), threads=7); b.simulate()
    np.testing.assert_array_equal(a.latest[0], b.latest[0])
    np.testing.assert_array_equal(a.latest[1], b.latest[1])


def test_accel_against_fp64_model(oracle):
    n = 1024
    pos, _ = oracle.initial_state(n, seed=1)
    a32 = oracle.accel_all_pairs(pos, mass=1.0)
    a64 = model_np.accel_fp64(pos, mass=1.0)
    scale = np.linalg.norm(a64, axis=1).max()
    assert np.abs(a32 - a64).max() / scale < 2e-6      # sequential fp32 sum of 1024 terms


def test_slice_step_equals_full_step(oracle):
    """The sharded decomposition is exact: advancing [first, first+count) alone gives the same
    bits as the full step, because each body's sum order does not depend on the slice."""
    n = 768
    pos, vel = oracle.initial_state(n, seed=9)
    prm = Params(mass=70000.0 / n)
    full = OracleSim(oracle, pos, vel, params=prm); full.simulate()
    for first, count in [(0, 256), (256, 256), (512, 256), (100, 37)]:
        p, v = oracle.step_slice(pos, vel, first, count, params=prm)
        np.testing.assert_array_equal(p, full.latest[0][first:first + count])
        np.testing.assert_array_equal(v, full.latest[1][first:first + count])


def test_ping_pong_and_frozen_tail(oracle):
    """Compute.cpp:1022,1034-1035,1003: write buffer idx, read 1-idx, flip; bodies past the
    active count keep whatever the written buffer held."""
    n = 256
    pos, vel = oracle.initial_state(n, seed=1)
    sim = OracleSim(oracle, pos, vel, mode=MODE_CENTRAL_WELL)
    assert sim.buffer_index == 0
    sim.simulate(num_active=100)            # -> 128 bodies advance
    assert sim.buffer_index == 1
    assert not np.array_equal(sim.pos[0][:128], pos[:128])
    np.testing.assert_array_equal(sim.pos[0][128:], pos[128:])     # frozen tail untouched
    np.testing.assert_array_equal(sim.pos[1], pos)                 # read buffer untouched
    sim.simulate(num_active=100)
    assert sim.buffer_index == 0
    np.testing.assert_array_equal(sim.pos[1][128:], pos[128:])


def test_golden_n256(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_n256.npz"))
    pos, vel = oracle.initial_state(256, seed=1)
    np.testing.assert_array_equal(pos, g["pos0"]); np.testing.assert_array_equal(vel, g["vel0"])
    for tag, mode, prm, steps in [("ap", MODE_ALL_PAIRS, Params(mass=70000.0 / 256), [1, 10, 100]),
                                  ("cw", MODE_CENTRAL_WELL, Params(), [1, 10, 100]),
                                  ("aplit", MODE_ALL_PAIRS, Params(), [1, 10])]:
        sim, done = OracleSim(oracle, pos, vel, mode=mode, params=prm), 0
        for s in steps:
            sim.simulate(steps=s - done); done = s
            np.testing.assert_array_equal(sim.latest[0], g[f"{tag}_pos_{s}"])
            np.testing.assert_array_equal(sim.latest[1], g[f"{tag}_vel_{s}"])
    a32 = oracle.accel_all_pairs(pos, mass=1.0)
    scale = np.linalg.norm(g["acc_fp64_unit_mass"], axis=1).max()
    assert np.abs(a32 - g["acc_fp64_unit_mass"]).max() / scale < 1e-6


def test_golden_n4096_config1(oracle, golden_dir):
    """BASELINE config #1: 4 096 bodies, 100 steps, fp32, CPU path."""
    g = np.load(os.path.join(golden_dir, "golden_n4096.npz"))
    n = 4096
    pos, vel = oracle.initial_state(n, seed=1)
    sums = [np.frombuffer(pos.tobytes(), np.uint32).sum(dtype=np.uint64), np.frombuffer(vel.tobytes(), np.uint32).sum(dtype=np.uint64)]
    assert sums == g["init_checksum"].tolist()
    sim = OracleSim(oracle, pos, vel, params=Params(mass=70000.0 / n))
    sim.simulate(steps=1)
    np.testing.assert_array_equal(sim.latest[0], g["pos_1"])
    sim.simulate(steps=99)
    np.testing.assert_array_equal(sim.latest[0], g["pos_100"])
    np.testing.assert_array_equal(sim.latest[1], g["vel_100"])


def test_momentum_and_energy_invariants_stable_regime(oracle):
    """Invariants in the mass = 70000/N regime: total momentum is conserved to rounding (pair
    terms are antisymmetric) and the shells stay bound over 100 steps."""
    n = 512
    pos, vel = oracle.initial_state(n, seed=1)
    sim = OracleSim(oracle, pos, vel, params=Params(mass=70000.0 / n))
    p0 = vel.astype(np.float64).sum(0)
    sim.simulate(steps=100)
    p1 = sim.latest[1].astype(np.float64).sum(0)
    assert np.abs(p1 - p0).max() < 1e-2 * n * 15 * 1e-3        # << total |p| scale n*15
    r = np.linalg.norm(sim.latest[0][:, :3], axis=1)
    assert r.max() < 2000 and np.isfinite(sim.latest[0]).all()


# ---------------------------------------------------------------------------------------------
# diagnostic summation variants: acc64, order-matched, full double

def test_sum_spec_reference_mode_is_the_oracle_proper(oracle):
    from oracle import OracleSim, Params, SumSpec, SUM_REFERENCE
    n = 1000
    pos, vel = oracle.initial_state(n, seed=2)
    prm = Params(mass=70000.0 / n)
    a = OracleSim(oracle, pos, vel, params=prm); a.simulate(steps=3)
    b = OracleSim(oracle, pos, vel, params=prm, sum_spec=SumSpec(SUM_REFERENCE)); b.simulate(steps=3)
    np.testing.assert_array_equal(a.latest[0], b.latest[0]); np.testing.assert_array_equal(a.latest[1], b.latest[1])


def test_f64_step_matches_the_independent_numpy_float64_model(oracle):
    from oracle import OracleSim64, Params
    n = 700
    pos, vel = oracle.initial_state(n, seed=3)
    prm = Params(mass=70000.0 / n)
    s = OracleSim64(oracle, pos, vel, params=prm); s.simulate()
    acc = model_np.accel_fp64(pos, mass=float(prm.mass), soft2=float(prm.soft2))
    x, v = model_np.integrate_fp64(pos, vel, acc, dt=float(prm.dt), damping=float(prm.damping))   # the fp32-rounded constants
    np.testing.assert_allclose(s.latest[0][:, :3], x[:, :3], rtol=1e-13, atol=1e-11)
    np.testing.assert_allclose(s.latest[1], v, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(s.latest[0][:, 3], x[:, 3], rtol=1e-12)


@pytest.mark.parametrize("n,waves,sb", [(1000, 8, 8), (4096, 16, 16), (130, 4, 3), (64, 1, 1), (3000, 8, 1)])
def test_acc64_and_order_matched_agree_with_the_double_step(oracle, n, waves, sb):
    """The three fp32 variants differ from the double step only by fp32 rounding: after one step
    velocities agree to a few 1e-7 of the speed scale; acc64 (summation error removed) is at least
    as close as the reference order.  Ragged N and S > number of tiles (empty chunks) included."""
    from oracle import OracleSim, OracleSim64, Params, SumSpec, SUM_FP64_ACC, SUM_ORDER_MATCHED
    pos, vel = oracle.initial_state(n, seed=4)
    if n % 2:
        pos[n - 1, :3] = [5.0, 6.0, 7.0]
    prm = Params(mass=70000.0 / n)
    truth = OracleSim64(oracle, pos, vel, params=prm); truth.simulate()
    err = {}
    for name, spec in (("ref", None), ("acc64", SumSpec(SUM_FP64_ACC)), ("matched", SumSpec(SUM_ORDER_MATCHED, waves, sb))):
        s = OracleSim(oracle, pos, vel, params=prm, sum_spec=spec); s.simulate()
        err[name] = np.abs(s.latest[1].astype(np.float64) - truth.latest[1]).max() / 15.0
        assert np.abs(s.latest[0][:, :3].astype(np.float64) - truth.latest[0][:, :3]).max() / 400.0 < 2e-7
        np.testing.assert_allclose(s.latest[0][:, 3], truth.latest[0][:, 3], rtol=2e-5, atol=1e-9)
    assert err["ref"] < 5e-7 and err["matched"] < 5e-7 and err["acc64"] <= err["ref"] + 6e-8, err


def test_order_matched_depends_on_the_plan_only_through_rounding(oracle):
    from oracle import OracleSim, Params, SumSpec, SUM_ORDER_MATCHED
    n = 2048
    pos, vel = oracle.initial_state(n, seed=5)
    prm = Params(mass=70000.0 / n)
    outs = []
    for waves, sb in ((8, 4), (4, 8), (1, 32), (1, 1)):
        s = OracleSim(oracle, pos, vel, params=prm, sum_spec=SumSpec(SUM_ORDER_MATCHED, waves, sb)); s.simulate(steps=2)
        outs.append(s.latest[1].copy())
    for o in outs[1:]:
        assert np.abs(o - outs[0]).max() / 15.0 < 3e-7
    assert any(not np.array_equal(o, outs[0]) for o in outs[1:])       # the order does reach the bits
