#!/usr/bin/env python3
"""Parity report: device path vs CPU oracle on identical seeded initial conditions,
free-running, statistics after 1, 10, 100 and 1000 steps.  Runs on the GPU box (the oracle uses
all host cores).  Regime: mass = 70000/N."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import mapn  # noqa: E402
from oracle import Oracle, OracleSim, Params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=65536)
ap.add_argument("--steps", default="1,10,100,1000")
ap.add_argument("--out", default="")
a = ap.parse_args()
n = a.bodies
marks = [int(s) for s in a.steps.split(",")]
o = Oracle()
pos, vel = o.initial_state(n, seed=1)
prm = Params(mass=70000.0 / n)
sim = OracleSim(o, pos, vel, params=prm)
rows = []
with mapn.Compute(n, mass=70000.0 / n) as c:
    assert np.array_equal(c.download_state()[0], pos)
    done = 0
    t_cpu = t_gpu = 0.0
    for m in marks:
        t0 = time.perf_counter(); sim.simulate(steps=m - done); t_cpu += time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(m - done):
            c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu(); t_gpu += time.perf_counter() - t0
        done = m
        gp, gv = c.download_state()
        rp, rv = sim.latest
        dx = np.linalg.norm(gp[:, :3].astype(np.float64) - rp[:, :3], axis=1)
        xr = np.maximum(np.linalg.norm(rp[:, :3].astype(np.float64), axis=1), 1e-30)
        dv = np.linalg.norm(gv.astype(np.float64) - rv, axis=1)
        dw = np.abs(gp[:, 3].astype(np.float64) - rp[:, 3])
        row = {"steps": m,
               "pos_rel_to_norm": {"max": float((dx / xr).max()), "median": float(np.median(dx / xr)), "rms": float(np.sqrt(((dx / xr) ** 2).mean()))},
               "pos_rel_to_spread400": {"max": float(dx.max() / 400), "median": float(np.median(dx) / 400), "rms": float(np.sqrt((dx ** 2).mean()) / 400)},
               "vel_rel_to_15": {"max": float(dv.max() / 15), "median": float(np.median(dv) / 15)},
               "w_rel_to_max": {"max": float(dw.max() / rp[:, 3].max())},
               "momentum_drift_rel": float(np.abs(gv.astype(np.float64).sum(0) - vel.astype(np.float64).sum(0)).max() / (n * 15.0)),
               "frac_bodies_within_1e-4_rel": float((dx / xr < 1e-4).mean())}
        rows.append(row)
        print(json.dumps(row), flush=True)
print(f"# N={n} mass=70000/N dt=0.1 soft2=25; oracle {t_cpu:.1f} s on {o.hardware_threads()} threads, device {t_gpu:.2f} s", flush=True)
if a.out:
    json.dump({"bodies": n, "rows": rows, "cpu_seconds": t_cpu, "gpu_seconds": t_gpu, "threads": o.hardware_threads()}, open(a.out, "w"), indent=1)
