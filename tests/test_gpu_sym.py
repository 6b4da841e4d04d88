"""The symmetric all-pairs kernel (csrc/mapn_sym.hip, MAPN_KERNEL_SYMMETRIC): every unordered pair
evaluated once, feeding both bodies.  Same pair term and integrator as the one-sided kernels, another
summation order -- so the same tolerances against the oracle apply (tests/test_gpu_parity.py), plus
what Newton's third law adds: the momentum change of a step is rounding only."""
import numpy as np
import pytest

import mapn
from oracle import OracleSim, Params

pytestmark = pytest.mark.gpu
SPREAD, SPEED = 400.0, 15.0


def draw(c, steps, num_active=None):
    n = c.num_particles if num_active is None else num_active
    for _ in range(steps):
        c.Simulate(n, c.GetFenceValue())


def errs(a, b, scale):
    d = np.linalg.norm(a.astype(np.float64) - b.astype(np.float64), axis=1) / scale
    return d.max(), np.median(d)


@pytest.mark.parametrize("n", [1024, 1025, 2048, 3000, 3072, 4096, 5000, 5120, 8192, 16384, 65536 + 100])
def test_symmetric_one_step_against_the_oracle(oracle, n):
    """Odd and even numbers of 1024-body blocks (the half-ring partner exists only for even counts),
    one block only (nothing symmetric to do), ragged N (the last block is padded with stand-in bodies that
    exert no force), teacher-forced."""
    mass = 70000.0 / n
    pos, vel = oracle.initial_state(n, seed=2)
    if n % 2:
        pos[n - 1, :3] = [10.0, -20.0, 30.0]; vel[n - 1] = [1.0, 2.0, 3.0]     # the generator leaves the odd body at the origin
    sim = OracleSim(oracle, pos, vel, params=Params(mass=mass)); sim.simulate()
    with mapn.Compute(n, mass=mass, seed=2, kernel=mapn.KERNEL_SYMMETRIC, flags=mapn.FLAG_NO_INIT) as c:
        c.upload_state(pos, vel)
        draw(c, 1)
        p, v = c.download_state()
        st = c.kernel_stats()
        assert st.kernel_name.decode() == "force_sym_kernel" and st.epilogue == 3
        assert np.isfinite(p).all() and np.isfinite(v).all()
    rp, rv = sim.latest
    assert errs(p[:, :3], rp[:, :3], SPREAD)[0] < 1e-6
    assert errs(v, rv, SPEED)[0] < 2e-5
    assert np.abs(p[:, 3] - rp[:, 3]).max() <= 1e-4 * rp[:, 3].max()


@pytest.mark.parametrize("n", [65536, 262144])
def test_symmetric_full_size_subset_momentum_and_reproducibility(oracle, n):
    mass = 70000.0 / n
    pos, vel = oracle.initial_state(n, seed=1)
    first = 1234 * 32
    rp, rv = oracle.step_slice(pos, vel, first, 4096, params=Params(mass=mass))
    out = []
    for rep in range(2):
        with mapn.Compute(n, mass=mass, kernel=mapn.KERNEL_SYMMETRIC) as c:
            draw(c, 1)
            p, v = c.download_state()
            assert c.kernel_stats().kernel_name.decode() == "force_sym_kernel"
            draw(c, 4)
            p5, v5 = c.download_state()
            out.append((p, v, p5, v5))
    p, v, p5, v5 = out[0]
    assert errs(p[first:first + 4096, :3], rp[:, :3], SPREAD)[0] < 1e-6
    assert errs(v[first:first + 4096], rv, SPEED)[0] < 2e-5
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)                      # fixed-order reductions: bit-reproducible
    p0 = vel.astype(np.float64).sum(0)
    drift = np.abs(v5.astype(np.float64).sum(0) - p0).max() / (n * SPEED)
    print(f"N={n}: relative momentum drift after 5 symmetric steps {drift:.2e}")
    assert drift < 1e-7 and np.isfinite(v5).all()


def test_symmetric_free_run_matches_golden_and_the_one_sided_kernel(oracle, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "golden_n4096.npz"))
    n = 4096
    with mapn.Compute(n, mass=70000.0 / n, kernel=mapn.KERNEL_SYMMETRIC) as c, mapn.Compute(n, mass=70000.0 / n) as ref:
        draw(c, 100); draw(ref, 100)
        p, v = c.download_state()
        q, _ = ref.download_state()
    mx, med = errs(p[:, :3], g["pos_100"][:, :3], SPREAD)
    print(f"symmetric kernel, 100-step free run N=4096: max |dx|/400 = {mx:.3e}, median = {med:.3e}")
    assert mx < 1e-4 and med < 1e-6
    assert errs(p[:, :3], q[:, :3], SPREAD)[0] < 1e-4


def test_symmetric_context_falls_back_where_the_kernel_does_not_apply(oracle):
    """num_active < N, or N smaller than one 1024-body block: the step runs the scalar-cache kernel (same results
    contract); with all bodies active again the symmetric kernel is back."""
    n = 4096
    pos, vel = oracle.initial_state(n, seed=4)
    prm = Params(mass=70000.0 / n)
    sim = OracleSim(oracle, pos, vel, params=prm)
    with mapn.Compute(n, mass=70000.0 / n, seed=4, kernel=mapn.KERNEL_SYMMETRIC) as c:
        for na, name in ((n, "force_sym_kernel"), (1000, "force_sgpr_kernel"), (n, "force_sym_kernel")):
            sim.simulate(num_active=na); draw(c, 1, num_active=na)
            assert c.kernel_stats().kernel_name.decode() == name
            for b in (0, 1):
                pb, vb = c.download_buffer(b)
                assert errs(pb[:, :3], sim.pos[b][:, :3], SPREAD)[0] < 3e-6
    with mapn.Compute(1000, mass=1.0, kernel=mapn.KERNEL_SYMMETRIC) as c:       # less than one block: one-sided
        draw(c, 1)
        assert c.kernel_stats().kernel_name.decode() == "force_sgpr_kernel"
