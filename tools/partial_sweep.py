#!/usr/bin/env python3
"""Partially active steps (num_active < N): ms per step of the three forms -- one-sided (active x N), full symmetric (N x N, the reduce
launch stops at num_active) and split (active x active symmetric + active x frozen one-sided) -- over a sweep of num_active, and the form
the library picks by itself.  The forms are selected through the MAPN_PARTIAL_FORM hook.  Usage: partial_sweep.py [--bodies N ...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapn  # noqa: E402


def ms_per_step(c, na, steps):
    for _ in range(max(3, steps // 4)):
        c.Simulate(na, c.GetFenceValue())
    c.WaitForGpu()
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            c.Simulate(na, c.GetFenceValue())
        c.WaitForGpu()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, nargs="+", default=[65536, 262144])
    ap.add_argument("--fractions", type=float, nargs="+", default=[0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 0.9375, 0.97])
    a = ap.parse_args()
    os.environ["MAPN_TEST_HOOKS"] = "1"
    for n in a.bodies:
        steps = max(2, min(300, int(0.15 / (0.6e-3 * (n / 65536.0) ** 2))))
        with mapn.Compute(n, mass=70000.0 / n) as c:
            c.set_timers(0)
            os.environ.pop("MAPN_PARTIAL_FORM", None)
            full_all = ms_per_step(c, n, steps)
            print(f"# N={n}: all bodies active {full_all:.4f} ms per step ({steps} steps per region, best of 3)", flush=True)
            print(f"# {'active':>9} {'one-sided':>10} {'full sym':>10} {'split':>10} {'split/one':>9} {'picked':>8} {'picked ms':>10}", flush=True)
            for f in a.fractions:
                na = int(n * f) // 64 * 64
                row = {}
                for form in ("one", "full", "split"):
                    os.environ["MAPN_PARTIAL_FORM"] = form
                    row[form] = ms_per_step(c, na, steps)
                    st = c.kernel_stats()
                    assert (st.split_active != 0) == (form == "split"), (form, st.split_active)
                os.environ.pop("MAPN_PARTIAL_FORM")
                picked_ms = ms_per_step(c, na, steps)
                st = c.kernel_stats()
                picked = "split" if st.split_active else ("full" if st.kernel_name.decode() == "force_sym_kernel" else "one")
                print(f"  {na:>9} {row['one']:>10.4f} {row['full']:>10.4f} {row['split']:>10.4f} {row['one'] / row['split']:>9.3f} {picked:>8} {picked_ms:>10.4f}", flush=True)


if __name__ == "__main__":
    main()
