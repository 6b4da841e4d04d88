#!/usr/bin/env python3
"""XCD-weighted parts A/B on one GPU (development tool): ms per step with the default plan and with calibrated weights,
interleaved.  Usage: tools/xcd_ab.py [N ...]   (also the sharded loopback step at 65 536 / 8 with --shard)"""
import os
import sys
import time

if "--shard" in sys.argv:
    os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_P2P_LOOPBACK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapn  # noqa: E402


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
    shard = "--shard" in sys.argv
    sizes = [int(x) for x in sys.argv[1:] if x.isdigit()] or [65536, 262144]
    for n in sizes:
        steps = max(20, int(300 * (65536.0 / n) ** 2)) * (8 if shard else 1)
        kw = dict(rank=0, world_size=8) if shard else {}
        with mapn.Compute(n, mass=70000.0 / n, **kw) as c:
            if shard:
                blob = c.p2p_export(); c.p2p_import([blob] * 8); c.set_gather_algorithm(5)
            c.set_timers(0)
            for _ in range(300 if n <= 131072 else 20):
                c.Simulate(n, c.GetFenceValue())
            c.WaitForGpu()
            w = c.calibrate_sym_xcds(4)
            print(f"N={n}{' / 8 loopback' if shard else ''}: calibrated XCD weights {w}", flush=True)
            if shard and "--from-unsharded" in sys.argv:
                # the dies' speeds measured by an UNSHARDED context of the same size on this GPU (every die holds heavy and light blocks there:
                # no class in the measurement), applied to the sharded launch
                with mapn.Compute(n, mass=70000.0 / n, flags=mapn.FLAG_XCD_CALIBRATE) as t:
                    w = list(t.sym_plan().xcd_weight)
                print(f"  ... weights of an unsharded context's creation-time calibration instead: {w}", flush=True)
            res = {"default": [], "weighted": []}
            for rep in range(4):
                c.set_sym_xcd_weights(None); res["default"].append(run(c, n, steps))
                c.set_sym_xcd_weights(w); res["weighted"].append(run(c, n, steps))
            pl = c.sym_plan()
            for k, v in res.items():
                print(f"  {k:9s} ms/step {' '.join('%.4f' % x for x in v)}  best {min(v):.4f}", flush=True)
            print(f"  weighted / default (best): {min(res['weighted']) / min(res['default']):.4f}   plan {pl.waves}x{pl.parts} ({pl.taper1},{pl.taper2}) sets {pl.sets} xcd_mode {pl.xcd_mode} bias {pl.wave_bias} class dies {pl.class_die if pl.xcd_mode == 2 else None}")
            w2 = c.calibrate_sym_xcds(4)
            print(f"  re-calibrated under the weighted plan: {w2}")


if __name__ == "__main__":
    main()
