"""The north_star's tolerance, gated: 65 536 bodies, 1000 free-running steps, device
vs the CPU oracle on identical initial conditions -- with the statistic NAMED and the
difference ATTRIBUTED.

Five CPU trajectories (tests/parity_report.py): `matched_sym` (the symmetric kernel's order restated from its launch plan), `ref` (the oracle proper: fp32, HLSL order, one
running sum), `acc64` (same pair terms, double accumulation), `matched` (the device's summation
order and fma fusion restated on the CPU; only v_rsq_f32 vs 1/sqrtf is left) and `f64` (the step in
double on double state).  Statements asserted, all on ||dx_i|| / ||x_i|| over the 65 536 bodies:

  T1  device vs ref after 1000 steps: median <= 1e-5, RMS <= 5e-5, >= 99.9 % of the bodies within
      1e-4.  The MAXIMUM over bodies is NOT within 1e-4 (a handful of bodies that passed close to
      another one amplify a 1-ulp difference: SURVEY F4's chaos).  What bounds it is measured IN THE SAME RUN, not a constant
      tuned to earlier observations: the reference-order oracle against its own double-accumulated twin --
      n_over_1e-4(device, ref) <= n_over_1e-4(ref, acc64) + 4 and max(device, ref) <= 1.5 x max(ref, acc64): the bodies beyond
      1e-4 are the ones `ref`'s single fp32 running sum over 65 536 terms puts there (rounds 2 - 4 measured: device vs ref
      7 .. 11 bodies, max 4.1e-4 .. 4.2e-4; ref vs acc64 9 bodies, max 4.3e-4).
  T2  device vs ref after 100 steps: max <= 2e-6 (every body far inside 1e-4).
  T3  attribution: the device is no farther from either yardstick than the reference-order oracle is --
      err(device, acc64) <= 1.5 x err(ref, acc64) for median, RMS AND max at 100 and 1000 steps, and
      err(device, f64) <= 1.5 x err(ref, f64) for median and RMS at 100 steps (the double leg stops
      there: it costs 10x an fp32 leg; its 1000-step numbers are in profiles/r02_parity_1000_65536_all_legs.json).
      I.e. the device-vs-ref difference is the fp32 summation order of `ref` itself (one running sum
      over 65 536 terms), amplified by the dynamics -- not a kernel defect.  Measured: device vs acc64
      after 1000 steps max 9.7e-5 (NO body beyond 1e-4), ref vs acc64 max 4.3e-4 (9 bodies beyond).
  T4  each device kernel vs ITS order-matched oracle (only v_rsq_f32 differs), after 1 and 100 steps: "device1s" vs `matched`, "device" (the
      symmetric kernel) vs `matched_sym`, "device_weighted" vs `matched_symw` -- tighter than T1/T2 by the bounds written below (the
      1000-step rows of these pairs: profiles/rNN_parity_1000_65536_all_legs.json, `tools/evidence.sh parity1000`).

T1-T3 are asserted for BOTH device kernels: "device" = MAPN_KERNEL_AUTO (the symmetric kernel at this size,
csrc/mapn_sym.hip) and "device1s" = the one-sided scalar-cache kernel whose summation order `matched` restates -- and for
"device_weighted": the symmetric kernel under an XCD-WEIGHTED launch plan (fixed lopsided die weights, class-aware), the kind of
plan bench.py's headline number runs (until round 4 only the unweighted plan went through the long legs).

The oracle legs take ~2.5 minutes on the 16 cores the GPU box's container is granted (it shows 256 threads; Oracle.best_threads picks the
count a step runs fastest at): ref 35 s and acc64 85 s for 1000 steps, f64 25 s and the three order-matched legs 3 - 4 s each for 100
(tests/oracle_leg_times.py measures them).
"""
import json
import os

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def report():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_report import run_report
    # (the three ORDER-MATCHED legs stop at 100 steps in this gating run since round 6 -- they restate the device's own order, so what 900 more
    #  steps add is v_rsq_f32's drift, not a new statement; their 1000-step rows are recorded every round by `tools/evidence.sh parity1000`:
    #  profiles/rNN_parity_1000_65536_all_legs.json)
    rep = run_report(65536, (1, 10, 100, 1000), f64_max_steps=100, matched_max_steps=100, log=lambda s: print(s, flush=True))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(rep, open(os.path.join(out, "parity_1000_65536.json"), "w"), indent=1)
    return rep


def _row(rep, steps, a, b):
    from parity_report import row
    return row(rep, steps, a, b)


LEGS = ["device", "device1s", "device_weighted"]     # default plan (symmetric kernel), one-sided kernel, symmetric kernel under an XCD-weighted plan


@pytest.mark.parametrize("leg", LEGS)
def test_t1_device_vs_reference_order_oracle_after_1000_steps(report, leg):
    """leg "device" = the default kernel (the symmetric one at this size), "device1s" = the one-sided one."""
    r = _row(report, 1000, leg, "ref")
    print(leg, "vs ref @1000:", r, report[leg + "_plan"], report.get(leg + "_sym_plan"))
    assert r["median"] <= 1e-5
    assert r["rms"] <= 5e-5
    assert r["frac_within_1e-4"] >= 0.999
    # NOT max <= 1e-4: see the module docstring and BASELINE.md section 4.  The yardstick is the oracle's OWN summation error, same run:
    own = _row(report, 1000, "ref", "acc64")
    print("ref vs acc64 @1000:", own)
    assert r["n_over_1e-4"] <= own["n_over_1e-4"] + 4, (r, own)
    assert r["max"] <= 1.5 * own["max"], (r, own)
    assert report[leg + "_momentum_drift_rel"] < 1e-7


@pytest.mark.parametrize("leg", LEGS)
def test_t2_device_vs_reference_order_oracle_after_100_steps(report, leg):
    for steps, bound in ((1, 5e-7), (10, 5e-7), (100, 2e-6)):
        r = _row(report, steps, leg, "ref")
        assert r["max"] <= bound, (steps, r)
        assert r["frac_within_1e-4"] == 1.0


@pytest.mark.parametrize("leg", LEGS)
@pytest.mark.parametrize("yardstick", ["acc64", "f64"])
def test_t3_device_is_no_farther_from_the_yardsticks_than_the_oracle_is(report, yardstick, leg):
    for steps in ((100, 1000) if yardstick == "acc64" else (100,)):     # the double leg stops at 100 steps (cost)
        dev, ref = _row(report, steps, leg, yardstick), _row(report, steps, "ref", yardstick)
        print(f"{leg} @{steps} vs {yardstick}: device median {dev['median']:.3e} rms {dev['rms']:.3e} max {dev['max']:.3e} | "
              f"ref median {ref['median']:.3e} rms {ref['rms']:.3e} max {ref['max']:.3e}")
        assert dev["median"] <= 1.5 * ref["median"], (steps, dev, ref)
        assert dev["rms"] <= 1.5 * ref["rms"], (steps, dev, ref)
        if yardstick == "acc64":
            # the MAXIMUM too: measured (round 2) device vs acc64 9.7e-5 -- inside 1e-4 -- against 4.3e-4 for the
            # reference-order oracle vs its own double-accumulated twin: the 4.2e-4 of T1 is the ORACLE's summation error
            assert dev["max"] <= 1.5 * ref["max"], (steps, dev, ref)
            if steps == 1000:
                assert dev["n_over_1e-4"] <= ref["n_over_1e-4"], (dev, ref)     # (measured rounds 2 - 4: 0 against 9)


@pytest.mark.parametrize("leg,yardstick", [("device1s", "matched"), ("device", "matched_sym"), ("device_weighted", "matched_symw")])
def test_t4_device_vs_order_matched_oracle_differs_by_rsq_only(report, leg, yardstick):
    r1, r100 = (_row(report, s, leg, yardstick) for s in (1, 100))
    print(leg, "vs", yardstick, ":", r1, r100, report.get("device_sym_plan"))
    assert r1["max"] <= 1.3e-7                 # one step: <= 1 ulp of the position
    assert r100["max"] <= 1e-6 and r100["median"] <= 3e-8
    assert r100["median"] <= _row(report, 100, leg, "ref")["median"]   # tighter than against the reference-order oracle
