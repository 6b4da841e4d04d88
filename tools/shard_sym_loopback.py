#!/usr/bin/env python3
"""One rank's sharded step at the TRUE shard size, on a 1-GPU box (MAPN_P2P_LOOPBACK=1: rank 0 of a P-way job with
every peer mapped to itself -- the real force / exchange launches run, results are not a simulation).  One-sided step
(gather algorithm 2; its separate pull launch is skipped in loopback) against the sharded symmetric step with pushed
positions (algorithm 5; the pushes go to the rank's own buffers).  Usage: tools/shard_sym_loopback.py [bodies] [steps]"""
import os
import sys
import time

os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_P2P_LOOPBACK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapn  # noqa: E402


def run(n, world, algo, steps):
    with mapn.Compute(n, device=0, mass=70000.0 / n, rank=0, world_size=world) as c:
        blob = c.p2p_export()
        c.p2p_import([blob] * world)
        c.set_gather_algorithm(algo)
        c.set_timers(4)
        for _ in range(max(20, steps // 4)):
            c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu()
        c.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu()
        dt = (time.perf_counter() - t0) / steps
        st = c.kernel_stats()
        return dt, st.avg_seconds, st.kernel_name.decode(), (st.grid_x, st.grid_y)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    print(f"# {n} bodies, rank 0's step at the shard size in loopback (symmetric: including the position exchange), {steps} steps")
    for world in (2, 4, 8):
        if (n // world) % 1024:
            continue
        r = {}
        for algo, name in ((2, "one-sided"), (5, "symmetric")):
            dt, ks, kn, grid = run(n, world, algo, steps)
            r[name] = dt
            print(f"world {world}  {name:10s} step {dt*1e6:8.1f} us   force kernel {ks*1e6:8.1f} us  {kn} grid {grid}"
                  f"   = {n * (n / world) / dt:.3e} interactions/s per rank", flush=True)
        print(f"world {world}  symmetric / one-sided step time: {r['symmetric'] / r['one-sided']:.3f}")


if __name__ == "__main__":
    main()
