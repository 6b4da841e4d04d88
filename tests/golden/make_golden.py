"""Generates the committed golden vectors under tests/golden/ (run from the repo root:
``python tests/golden/make_golden.py``).

The reference holds no golden data for this path and cannot be executed, so
these vectors come from the CPU oracle (oracle/mapn_oracle.c, fp32, exact HLSL op order) and
from the independent float64 numpy model (oracle/model_np.py).  They pin the oracle against
regressions and give the GPU tests fixed targets.  Regime: mass = 70000/N unless noted
(with the literal per-body 70000 all-pairs is chaotic within ~10 steps).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import MODE_ALL_PAIRS, MODE_CENTRAL_WELL, Oracle, OracleSim, Params  # noqa: E402
from oracle import model_np  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(o, pos, vel, mode, params, steps_list):
    sim = OracleSim(o, pos, vel, mode=mode, params=params)
    out, done = {}, 0
    for s in steps_list:
        sim.simulate(steps=s - done)
        done = s
        p, v = sim.latest
        out[s] = (p.copy(), v.copy())
    return out


def main():
    o = Oracle()
    # ---- N = 256: full set ---------------------------------------------------------------
    n = 256
    pos, vel = o.initial_state(n, seed=1)
    d = {"pos0": pos, "vel0": vel}
    stable = Params(mass=70000.0 / n)
    for s, (p, v) in run(o, pos, vel, MODE_ALL_PAIRS, stable, [1, 10, 100]).items():
        d[f"ap_pos_{s}"], d[f"ap_vel_{s}"] = p, v
    for s, (p, v) in run(o, pos, vel, MODE_CENTRAL_WELL, Params(), [1, 10, 100]).items():
        d[f"cw_pos_{s}"], d[f"cw_vel_{s}"] = p, v
    for s, (p, v) in run(o, pos, vel, MODE_ALL_PAIRS, Params(), [1, 10]).items():      # literal mass 70000
        d[f"aplit_pos_{s}"], d[f"aplit_vel_{s}"] = p, v
    d["acc_fp64_unit_mass"] = model_np.accel_fp64(pos, mass=1.0)                       # per-body 1-step table
    np.savez_compressed(os.path.join(OUT, "golden_n256.npz"), **d)

    # ---- N = 4096, 100 steps: BASELINE config #1 -------------------------------------------
    n = 4096
    pos, vel = o.initial_state(n, seed=1)
    res = run(o, pos, vel, MODE_ALL_PAIRS, Params(mass=70000.0 / n), [1, 100])
    np.savez_compressed(os.path.join(OUT, "golden_n4096.npz"),
                        pos_1=res[1][0], vel_1=res[1][1], pos_100=res[100][0], vel_100=res[100][1],
                        init_checksum=np.array([np.frombuffer(pos.tobytes(), np.uint32).sum(dtype=np.uint64),
                                                np.frombuffer(vel.tobytes(), np.uint32).sum(dtype=np.uint64)]))
    # ---- init-state checksums for larger N (no arrays stored) ------------------------------
    sums = {}
    for n in (1000, 65536):
        p, v = o.initial_state(n, seed=1)
        sums[str(n)] = np.array([np.frombuffer(p.tobytes(), np.uint32).sum(dtype=np.uint64),
                                 np.frombuffer(v.tobytes(), np.uint32).sum(dtype=np.uint64)])
    np.savez(os.path.join(OUT, "init_checksums.npz"), **sums)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
