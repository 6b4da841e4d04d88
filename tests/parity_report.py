#!/usr/bin/env python3
"""Parity report: the device path against FOUR CPU trajectories started
from identical seeded initial conditions, free-running, statistics after 1, 10, 100 and 1000 steps.

  ref      the oracle proper: fp32, the HLSL's operation order, one running sum over ascending j
           (nBodyGravityCS.hlsl:44-57, :103-108) -- the north_star's "CPU reference";
  acc64    the same fp32 pair terms accumulated in double (summation error removed);
  matched  the ONE-SIDED device kernel's summation ORDER and operation FUSION restated on the CPU (chunked
           sums, fma, mass after the sum) -- what is left against the device is v_rsq_f32 alone;
  matched_sym  the same for the SYMMETRIC kernel (the default one), restated from the launch plan the context
           runs (mapn_get_sym_plan): waves' step ranges, reaction chains, cut meetings, rows, windows;
  matched_symw the same for the "device_weighted" leg: the symmetric kernel under an XCD-weighted (class-aware) plan;
  f64      the whole step in double on double state: the discrete map itself.

Comparing device|ref|matched with acc64 / f64 attributes a device-vs-ref difference: if the device
is no farther from the yardsticks than ref is, the difference is fp32 summation order amplified by
the dynamics, not a kernel defect.  Regime: mass = 70000/N.

Runs on the GPU box (the oracles use all host cores).  Imported by tests/test_parity_1000.py; as a
script it prints one JSON row per (mark, pair) and writes --out.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root


def stats(a_pos, b_pos):
    """Named statistics of the relative position error ||a_i - b_i|| / ||b_i|| over bodies."""
    a = np.asarray(a_pos, np.float64)[:, :3]
    b = np.asarray(b_pos, np.float64)[:, :3]
    dx = np.linalg.norm(a - b, axis=1)
    rel = dx / np.maximum(np.linalg.norm(b, axis=1), 1e-30)
    return {"max": float(rel.max()), "median": float(np.median(rel)), "rms": float(np.sqrt((rel ** 2).mean())),
            "frac_within_1e-4": float((rel < 1e-4).mean()), "n_over_1e-4": int((rel >= 1e-4).sum()),
            "max_over_spread400": float(dx.max() / 400.0)}


WEIGHTED_LEG_XCD_WEIGHTS = (1024, 900, 1000, 950, 1024, 880, 990, 1010)     # relative die speeds of the "device_weighted" leg (1024 = the fastest)


def run_report(n=65536, marks=(1, 10, 100, 1000), seed=1, with_f64=True, f64_max_steps=None, log=print, weighted_leg=True, matched_max_steps=None):
    """matched_max_steps: the three ORDER-MATCHED legs (`matched`, `matched_sym`, `matched_symw`) stop there (None: they run all marks -- what
    `tools/evidence.sh parity1000` records under profiles/ every round).  (the gpu suite's time is mostly these legs; a leg
    costs 30 - 40 s per 1000 steps on the 16 cores the GPU box's container is granted.)"""
    import mapn
    from oracle import Oracle, OracleSim, OracleSim64, Params, SumSpec, SUM_FP64_ACC, SUM_ORDER_MATCHED

    o = Oracle()
    pos, vel = o.initial_state(n, seed=seed)
    prm = Params(mass=70000.0 / n)
    out = {"bodies": n, "seed": seed, "regime": "mass=70000/N, soft2=25, dt=0.1, damping=1", "threads": o.hardware_threads(),
           "marks": list(marks), "timing_s": {}, "rows": []}
    snaps = {}
    sym_plan = None
    # two device legs: "device" = the default kernel (MAPN_KERNEL_AUTO: the symmetric kernel where it applies),
    # "device1s" = the one-sided scalar-cache kernel, whose summation order the `matched` oracle restates
    # "device_weighted" = the symmetric kernel under an XCD-WEIGHTED plan (fixed, deliberately lopsided weights: class-aware where it
    # applies) -- the KIND of plan bench.py's headline number runs (there the weights are the box's calibrated ones), through the same
    # 100 / 1000 steps and the same statements as the default plan
    legs = [("device", mapn.KERNEL_AUTO, None), ("device1s", mapn.KERNEL_SCALAR, None)]
    if weighted_leg:
        legs.append(("device_weighted", mapn.KERNEL_AUTO, list(WEIGHTED_LEG_XCD_WEIGHTS)))
    symw_plan = None
    for leg, kern, xcd_w in legs:
        with mapn.Compute(n, mass=70000.0 / n, seed=seed, kernel=kern) as c:
            assert np.array_equal(c.download_state()[0], pos)
            if xcd_w is not None:
                try:
                    if c.sym_plan().nb % 8:
                        continue                               # (XCD weights need a block count that is a multiple of 8)
                except mapn.MapnError:
                    continue                                   # (the symmetric kernel does not run at this size)
                c.set_sym_xcd_weights(xcd_w)
                symw_plan = c.sym_plan()
                assert symw_plan.xcd_mode != 0 and symw_plan.xcd_weight == xcd_w
                out["device_weighted_sym_plan"] = {"waves": symw_plan.waves, "parts": symw_plan.parts, "xcd_mode": symw_plan.xcd_mode,
                                                   "xcd_weight": list(symw_plan.xcd_weight), "windows": len(symw_plan.windows)}
            done, t0 = 0, time.perf_counter()
            for m in marks:
                for _ in range(m - done):
                    c.Simulate(n, c.GetFenceValue())
                done = m
                snaps[(leg, m)] = c.download_state()
            out["timing_s"][leg] = time.perf_counter() - t0
            st = c.kernel_stats()
            out[leg + "_plan"] = {"kernel": st.kernel_name.decode(), "waves": st.block_x // 64, "sb": st.grid_y, "j_splits": st.j_splits,
                                  "bodies_per_lane": st.bodies_per_lane, "epilogue": st.epilogue}
            if leg == "device1s":
                waves, sb = st.block_x // 64, st.grid_y
            elif st.kernel_name.decode() == "force_sym_kernel" and xcd_w is None:
                sym_plan = c.sym_plan()
                out["device_sym_plan"] = {"waves": sym_plan.waves, "parts": sym_plan.parts, "taper": [sym_plan.taper1, sym_plan.taper2],
                                          "windows": len(sym_plan.windows)}
            v0 = vel.astype(np.float64).sum(0)
            out[leg + "_momentum_drift_rel"] = float(np.abs(snaps[(leg, marks[-1])][1].astype(np.float64).sum(0) - v0).max() / (n * 15.0))
    out["momentum_drift_rel"] = out["device_momentum_drift_rel"]
    sims = {"ref": OracleSim(o, pos, vel, params=prm),
            "acc64": OracleSim(o, pos, vel, params=prm, sum_spec=SumSpec(SUM_FP64_ACC)),
            "matched": OracleSim(o, pos, vel, params=prm, sum_spec=SumSpec(SUM_ORDER_MATCHED, waves, sb))}
    if sym_plan is not None:
        sims["matched_sym"] = OracleSim(o, pos, vel, params=prm, sym_plan=sym_plan)
    if symw_plan is not None:
        sims["matched_symw"] = OracleSim(o, pos, vel, params=prm, sym_plan=symw_plan)
    if with_f64:
        sims["f64"] = OracleSim64(o, pos, vel, params=prm)
    have = set()
    # Every leg on the thread count a step runs FASTEST at on this host (round 6; Oracle.best_threads: a step creates and joins its workers,
    # and the GPU box's container does not have the 256 cores it shows -- 65 536 bodies, ms per step at 64 / 128 / 256 threads: ref 30 / 47 /
    # 42, acc64 78 / 101 / 131, f64 250 / 276 / 456, tests/oracle_leg_times.py; the legs side by side on a quarter of the threads each were
    # tried and took six times as long apiece).  The result of a leg does not depend on its thread count (tests/test_oracle_models.py).
    leg_threads = o.best_threads()
    for name, sim in sims.items():
        sim.threads = leg_threads
        done, t0 = 0, time.perf_counter()
        for m in marks:
            if name == "f64" and f64_max_steps is not None and m > f64_max_steps:
                break                                          # the double leg is ~10x the fp32 legs' cost
            if name.startswith("matched") and matched_max_steps is not None and m > matched_max_steps:
                break
            have.add((name, m))
            sim.simulate(steps=m - done)
            done = m
            p, v = sim.latest
            snaps[(name, m)] = (p.copy(), v.copy())
        out["timing_s"][name] = time.perf_counter() - t0
        log(f"# oracle {name}: {out['timing_s'][name]:.1f} s for {done} steps of {n} bodies on {leg_threads} of {o.hardware_threads()} threads")
    pairs = [("device", "ref"), ("device", "acc64"), ("device1s", "ref"), ("device1s", "matched"), ("device1s", "acc64"),
             ("device", "device1s"), ("ref", "acc64"), ("matched", "acc64")]
    if sym_plan is not None:
        pairs += [("device", "matched_sym"), ("matched_sym", "acc64")]
    if symw_plan is not None:
        pairs += [("device_weighted", "ref"), ("device_weighted", "acc64"), ("device_weighted", "matched_symw"), ("device_weighted", "device")]
    if with_f64:
        pairs += [("device", "f64"), ("device1s", "f64"), ("ref", "f64"), ("matched", "f64"), ("acc64", "f64")]
        if symw_plan is not None:
            pairs += [("device_weighted", "f64")]
    for m in marks:
        for a, b in pairs:
            if any((x, m) not in have and not x.startswith("device") for x in (a, b)):
                continue                                       # (a leg that stopped earlier: f64, the bounded order-matched legs)
            row = {"steps": m, "a": a, "b": b, **stats(snaps[(a, m)][0], snaps[(b, m)][0])}
            dv = np.linalg.norm(snaps[(a, m)][1].astype(np.float64) - snaps[(b, m)][1].astype(np.float64), axis=1)
            row["vel_max_over_15"] = float(dv.max() / 15.0)
            out["rows"].append(row)
            log(json.dumps(row))
    return out


def row(report, steps, a, b):
    for r in report["rows"]:
        if r["steps"] == steps and r["a"] == a and r["b"] == b:
            return r
    raise KeyError((steps, a, b))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=65536)
    ap.add_argument("--steps", default="1,10,100,1000")
    ap.add_argument("--no-f64", action="store_true")
    ap.add_argument("--f64-max-steps", type=int, default=None)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rep = run_report(a.bodies, tuple(int(s) for s in a.steps.split(",")), with_f64=not a.no_f64, f64_max_steps=a.f64_max_steps)
    if a.out:
        json.dump(rep, open(a.out, "w"), indent=1)
