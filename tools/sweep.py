#!/usr/bin/env python3
"""Kernel-variant sweep on the GPU box (development tool): times the force kernel for a list of
plans at a given N and prints interactions/s from the in-library event timers."""
import argparse
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bodies", type=int, default=65536)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--kernels", default="lds,sgpr")
ap.add_argument("--ks", default="2,4,8")
ap.add_argument("--waves", default="4,8,16")
ap.add_argument("--sbs", default="1,2,4")
ap.add_argument("--timer-interval", type=int, default=1)
ap.add_argument("--overlap", action="store_true")
ap.add_argument("--fused", type=int, default=1, help="0 = rows + reduce launch, 1 = one launch (ticket when sb > 1)")
ap.add_argument("--auto", action="store_true", help="do not force a plan: time the library's own choice once")
ap.add_argument("--world", type=int, default=1, help="emulate one shard of a world_size-way job (external gather, rank 0)")
a = ap.parse_args()
n = a.bodies
KN = {"lds": mapn.KERNEL_LDS, "sgpr": mapn.KERNEL_SCALAR}
with mapn.Compute(n, mass=70000.0 / n, rank=0, world_size=a.world, flags=mapn.FLAG_SHARD_OVERLAP if a.overlap else 0) as c:
    if a.world > 1 and os.environ.get("MAPN_COMM_LOOPBACK") == "1":    # (run as: MAPN_TEST_HOOKS=1 MAPN_COMM_LOOPBACK=1 python tools/sweep.py ...)
        c.comm_init(mapn.Compute.comm_unique_id())      # real sharded step structure, 1-rank RCCL
    elif a.world > 1:
        c.set_external_gather(True)
    c.set_timers(a.timer_interval)
    rows = []
    n_i = n // a.world
    combos = list(itertools.product(a.kernels.split(","), map(int, a.ks.split(",")), map(int, a.waves.split(",")), map(int, a.sbs.split(","))))
    if a.auto:
        combos = [("auto", 0, 0, 0)] * 3                 # the library's own plan, three repeats
    for kn, k, w, sb in combos:
        try:
            if a.auto:
                c.set_force_plan(mapn.KERNEL_AUTO)
            else:
                c.set_force_plan(KN[kn], k, w, sb, a.fused)
        except mapn.MapnError:
            continue
        for _ in range(3):
            c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu(); c.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu()
        wall = (time.perf_counter() - t0) / a.steps
        st = c.kernel_stats()
        if a.auto:
            kn, k, w, sb = st.kernel_name.decode().replace("force_", "").replace("_kernel", ""), st.bodies_per_lane, st.block_x // 64, st.grid_y
        ks = st.avg_seconds if st.launches else float("nan")
        rate = n_i * n / (ks if st.launches else wall)
        rows.append((rate, kn, k, w, sb, ks * 1e3, wall * 1e3))
        print(f"{kn:5s} k={k} waves={w:2d} sb={sb:2d}  kernel {ks*1e3:8.3f} ms  step {wall*1e3:8.3f} ms  {rate:.3e} pairs/s  {20*rate/157.3e12*100:5.1f}% fp32 peak", flush=True)
    rows.sort(reverse=True)
    print("best:", rows[:5])
