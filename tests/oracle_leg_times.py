#!/usr/bin/env python3
"""What a step of each oracle leg costs on THIS host (the GPU box's cores): the budget behind tests/test_parity_1000.py and the
1000-step split-form test (the gpu suite's time is mostly these legs).  python tests/oracle_leg_times.py [bodies] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import Oracle, OracleSim, OracleSim64, Params, SumSpec, SUM_FP64_ACC, SUM_ORDER_MATCHED   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0                     # 0: all hardware threads, -1: Oracle.best_threads()
o = Oracle()
pos, vel = o.initial_state(n, seed=1)
prm = Params(mass=70000.0 / n)
legs = {"ref": lambda: OracleSim(o, pos, vel, params=prm, threads=threads),
        "acc64": lambda: OracleSim(o, pos, vel, params=prm, sum_spec=SumSpec(SUM_FP64_ACC), threads=threads),
        "matched(8x8)": lambda: OracleSim(o, pos, vel, params=prm, sum_spec=SumSpec(SUM_ORDER_MATCHED, 8, 8), threads=threads),
        "f64": lambda: OracleSim64(o, pos, vel, params=prm, threads=threads)}
if threads < 0:
    threads = o.best_threads()
quota = o.cpu_quota_cores()
print(f"{n} bodies, {threads or o.hardware_threads()} of {o.hardware_threads()} threads (affinity {len(os.sched_getaffinity(0))}, cpu quota {quota} cores), {steps} steps per leg (after one warm-up step)")
for name, make in legs.items():
    sim = make()
    sim.simulate(steps=1)
    t0 = time.perf_counter()
    sim.simulate(steps=steps)
    dt = (time.perf_counter() - t0) / steps
    print(f"  {name:14s} {dt * 1e3:8.1f} ms per step = {n * n / dt:.3e} pairs/s -> {dt * 1000:.0f} s per 1000 steps")
