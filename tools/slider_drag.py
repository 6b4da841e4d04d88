import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mapn
for n in (65536, 262144):
    with mapn.Compute(n, mass=70000.0 / n) as c:
        c.set_timers(0)
        a, b = n // 2, n // 2 + 1024
        for na in (a, b):
            for _ in range(20): c.Simulate(na, c.GetFenceValue())
        c.WaitForGpu()
        k = 100 if n == 65536 else 20
        t0 = time.perf_counter()
        for _ in range(k): c.Simulate(a, c.GetFenceValue())
        c.WaitForGpu(); steady = (time.perf_counter() - t0) / k * 1e3
        t0 = time.perf_counter()
        for i in range(k): c.Simulate(a if i % 2 else b, c.GetFenceValue())
        c.WaitForGpu(); drag = (time.perf_counter() - t0) / k * 1e3
        t0 = time.perf_counter()
        for i in range(k): c.Simulate(a + 64 * (i % 50), c.GetFenceValue())
        c.WaitForGpu(); drag2 = (time.perf_counter() - t0) / k * 1e3
        built = c.kernel_stats().split_plans_built
        print(f"N={n}: steady {steady:.4f} ms per step; num_active alternating between two counts every step {drag:.4f}; a new count every step (slider drag: 50 counts "
              f"through a cache of 4 plans) {drag2:.4f}; host plans built in all: {built}")
