#!/usr/bin/env python3
"""Sweep of the symmetric kernel's launch shape on one GPU (development tool): for each N, every (waves, parts, taper1, taper2, groups per window, wave bias)
of a list through mapn_set_sym_plan, interleaved repeats, ms per step.  Usage: tools/sym_plan_sweep.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapn  # noqa: E402

SHAPES = [(0, 0, 0, 0), (4, 32, 0, 0), (4, 40, 28, 4), (4, 36, 28, 4), (4, 48, 24, 8), (4, 24, 0, 0), (4, 64, 0, 0),
          (8, 16, 0, 0), (8, 20, 14, 2),
          # 8-wave workgroups whose first four waves (the older wave of every SIMD) carry hi : lo of the steps
          (8, 16, 0, 0, 0, (3, 1)), (8, 16, 0, 0, 0, (2, 1)), (8, 16, 0, 0, 0, (5, 2)), (8, 12, 0, 0, 0, (3, 1)), (8, 8, 0, 0, 0, (3, 1)), (8, 8, 0, 0, 0, (4, 1)),
          (8, 20, 14, 2, 0, (3, 1)), (8, 18, 14, 4, 0, (3, 1)), (8, 20, 12, 4, 0, (3, 1)), (8, 24, 12, 8, 0, (3, 1)), (8, 18, 12, 4, 0, (3, 1)),
          (8, 14, 10, 2, 0, (3, 1)), (8, 14, 10, 2, 0, (4, 1)), (8, 10, 6, 2, 0, (4, 1)), (8, 12, 8, 2, 0, (7, 2))]


def run(c, n, steps):
    for _ in range(max(8, steps // 8)):
        c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu()
    t0 = time.perf_counter()
    for _ in range(steps):
        c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    global SHAPES
    if os.environ.get("SWEEP_SHAPES"):                         # e.g. SWEEP_SHAPES="[(0,0,0,0),(8,8,0,0,0,(3,1))]"
        SHAPES = eval(os.environ["SWEEP_SHAPES"])
    sizes = [int(x) for x in sys.argv[1:]] or [65536, 100000, 131072]
    for n in sizes:
        steps = max(20, int(200 * (65536.0 / n) ** 2))
        with mapn.Compute(n, mass=70000.0 / n) as c:
            c.set_timers(0)
            if os.environ.get("SWEEP_XCD") == "1":
                w = c.calibrate_sym_xcds(4)
                c.set_sym_xcd_weights(w)
                print(f"N={n}: XCD weights {w}")
            for _ in range(300 if n <= 131072 else 20):
                c.Simulate(n, c.GetFenceValue())
            c.WaitForGpu()
            best = {}
            for rep in range(3):
                for sh in SHAPES:
                    try:
                        c.set_sym_plan(*sh)
                    except mapn.MapnError as e:
                        if rep == 0:
                            print(f"N={n} shape {sh}: refused ({str(e)[:90]})")
                        continue
                    ms = run(c, n, steps)
                    best.setdefault(sh, []).append(ms)
            pl = None
            for sh, v in sorted(best.items(), key=lambda kv: min(kv[1])):
                print(f"N={n} waves,parts,t1,t2,gpw,bias={sh}: ms/step {' '.join('%.4f' % x for x in v)}  best {min(v):.4f}  = {n * n / min(v) / 1e-3:.3e} interactions/s", flush=True)


if __name__ == "__main__":
    main()
