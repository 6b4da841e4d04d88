"""Known-answer tests of the CPU oracle, derived by hand from the reference text (SURVEY.md 8c,
K1..K8).  The reference holds no tests of its own; these pin the restatement op by op."""
import numpy as np
import pytest

from oracle import MODE_CENTRAL_WELL, OracleSim, Params

f32 = np.float32


def test_k1_pair_term(oracle):
    # nBodyGravityCS.hlsl:44-57: bi=(0,0,0), bj=(3,4,0), mass 70000, particles 1
    a = oracle.pair_term([0, 0, 0], [3, 4, 0, 9], [0, 0, 0, 7])      # .w of both ignored
    d = f32(50.0); inv = f32(1.0) / np.sqrt(d)
    s = f32(70000.0) * (inv * inv * inv)
    assert a.dtype == np.float32
    np.testing.assert_array_equal(a, np.array([f32(3) * s, f32(4) * s, 0], f32))
    np.testing.assert_allclose(a, [593.9697, 791.95966, 0.0], rtol=2e-7)
    assert abs(float(s) - 197.98989873) < 2e-5


def test_k1_particles_multiplier(oracle):
    a1 = oracle.pair_term([0, 0, 0], [3, 4, 0, 0], [0, 0, 0, 0], particles=1)
    a3 = oracle.pair_term([0, 0, 0], [3, 4, 0, 0], [0, 0, 0, 0], particles=3)
    np.testing.assert_allclose(a3, 3 * a1, rtol=2e-7)


def test_k2_self_pair_adds_exact_zero(oracle):
    b = [123.5, -7.25, 400.0, 1.0]
    a = oracle.pair_term([1.5, -2.5, 3.5], b, b)
    np.testing.assert_array_equal(a, np.array([1.5, -2.5, 3.5], f32))   # r = 0 -> adds exactly 0


def test_k3_central_well_step(oracle):
    # hlsl:86-109 with pos=(3,4,0,0), vel=0, dt 0.1, damping 1
    pos = np.tile(np.array([[3, 4, 0, 0]], f32), (64, 1))
    sim = OracleSim(oracle, pos, np.zeros((64, 3), f32), mode=MODE_CENTRAL_WELL)
    sim.simulate()
    p, v = sim.latest
    np.testing.assert_allclose(v[0], [-59.396973, -79.19597, 0.0], rtol=2e-7)
    np.testing.assert_allclose(p[0], [-2.9396973, -3.9195971, 0.0, 989.9496], rtol=3e-7)
    assert abs(float(p[0, 3]) - 989.94949366) < 1e-4


def test_k4_two_body_step(oracle):
    # bodies (0,0,0) and (3,4,0), mass 1, soft2 25, dt 0.1: equal and opposite accelerations
    pos = np.zeros((2, 4), f32); pos[1, :3] = [3, 4, 0]
    sim = OracleSim(oracle, pos, np.zeros((2, 3), f32), params=Params(mass=1.0))
    sim.simulate(num_active=2)
    p, v = sim.latest
    a0 = v[0] / f32(0.1)
    np.testing.assert_allclose(a0, [0.0084852814, 0.011313708, 0.0], rtol=3e-7)
    np.testing.assert_array_equal(v[0], -v[1])                       # momentum exactly antisymmetric
    np.testing.assert_allclose(p[0, 3], 0.014142136, rtol=3e-7)
    np.testing.assert_allclose(p[0, :3], [8.4852814e-5, 1.1313708e-4, 0.0], rtol=3e-7)


def test_k5_fast_rand(oracle):
    # Compute.cpp:599-609 == MSVC rand()
    assert oracle.fast_rand(1, 8) == [41, 18467, 6334, 26500, 19169, 15724, 11478, 29358]
    assert oracle.fast_rand(0, 8) == [38, 7719, 21238, 2437, 8855, 11797, 8365, 32285]


def test_k6_rand_sse(oracle):
    # Compute.cpp:622-661, seed 0
    assert oracle.rand_sse(0, 3) == [[41, 158, 212, 0], [18467, 8430, 13005, 1], [6334, 31659, 363, 7257]]


@pytest.mark.parametrize("n,seed", [(256, 1), (4096, 7), (1000, 3)])
def test_k7_shell_invariants(oracle, n, seed):
    pos, vel = oracle.initial_state(n, seed=seed)
    half = n // 2
    cx = np.where(np.arange(n) < half, 300.0, -300.0)
    rel = pos[:, :3].astype(np.float64) - np.stack([cx, 0 * cx, 0 * cx], 1)
    np.testing.assert_allclose(np.linalg.norm(rel, axis=1), 400.0, rtol=0, atol=2e-4)    # Compute.cpp:695-699
    speed = np.linalg.norm(vel.astype(np.float64), axis=1)
    assert speed.max() <= 15.0 + 1e-4                                                     # |dir x perp| <= 1
    assert speed.min() > 0.0                                                              # = 15*sin(angle(dir, perp))
    radial = (vel.astype(np.float64) * pos[:, :3]).sum(1) / np.linalg.norm(pos[:, :3], axis=1)
    assert np.abs(radial).max() < 1e-4                                                    # tangential
    assert np.all(pos[:, 3] == 0)
    # two well separated groups, different bodies differ
    assert len(np.unique(pos[:, 0])) > 0.99 * n


def test_k7_seed_and_order_independence(oracle):
    a, _ = oracle.initial_state(512, seed=1)
    b, _ = oracle.initial_state(512, seed=2)
    assert not np.array_equal(a, b)
    # per-body seeding: a longer run reproduces the first group's bodies only where the global
    # index AND the centre agree (bodies [0,256) of N=512 vs [0,256) of N=1024 share both)
    c, _ = oracle.initial_state(1024, seed=1)
    np.testing.assert_array_equal(a[:256], c[:256])


def test_k8_cbuffer(oracle):
    p, f = oracle.cbuffer(65536)
    assert p == [65536, 1024, 0, 0]
    assert f.view(np.uint32).tolist() == [0x3DCCCCCD, 0x3F800000, 0, 0]                   # Compute.cpp:545-546


@pytest.mark.parametrize("na,n,expect", [(0, 4096, 0), (-5, 4096, 0), (1, 4096, 64), (64, 4096, 64), (65, 4096, 128),
                                         (4096, 4096, 4096), (5000, 4096, 4096), (100, 100, 100), (90, 100, 100)])
def test_active_rounding(oracle, na, n, expect):
    # Compute.cpp:1041 Dispatch(ceil(numActive/64)) x 64 threads, out-of-range dropped
    assert oracle.active_bodies(na, n) == expect
