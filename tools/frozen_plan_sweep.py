import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MAPN_TEST_HOOKS"] = "1"
import mapn
def ms(c, na, k):
    for _ in range(k // 4): c.Simulate(na, c.GetFenceValue())
    c.WaitForGpu(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(k): c.Simulate(na, c.GetFenceValue())
        c.WaitForGpu(); best = min(best, (time.perf_counter() - t0) / k * 1e3)
    return best
for n, na, k in ((65536, 32768, 300), (65536, 49152, 250), (262144, 131072, 20)):
    for plan in ("", "2,8,8", "2,16,4", "2,16,8", "2,8,32", "4,8,8", "4,8,16", "4,4,16", "2,4,32"):
        if plan: os.environ["MAPN_FROZEN_PLAN"] = plan
        else: os.environ.pop("MAPN_FROZEN_PLAN", None)
        with mapn.Compute(n, mass=70000.0 / n) as c:
            c.set_timers(0)
            t = ms(c, na, k)
            sp, pl = c.split_plan()
            print(f"N={n} active={na} frozen plan {plan or 'default'} -> k={sp.frozen_bodies_per_lane} waves={sp.frozen_waves} sb={sp.frozen_sb}: {t:.4f} ms", flush=True)
