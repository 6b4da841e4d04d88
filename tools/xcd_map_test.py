import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mapn
def run(c, n, steps):
    for _ in range(max(8, steps // 8)): c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu(); t0 = time.perf_counter()
    for _ in range(steps): c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu(); return (time.perf_counter() - t0) / steps * 1e3
for n in (65536, 262144):
    steps = max(20, int(300 * (65536.0 / n) ** 2))
    with mapn.Compute(n, mass=70000.0 / n) as c:
        c.set_timers(0)
        for _ in range(300 if n <= 131072 else 20): c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu()
        w = c.calibrate_sym_xcds(4)
        inv = [int(1024 * 1024 / x) for x in w]; inv = [int(x * 1024 / max(inv)) for x in inv]
        for name, ws in (("default", None), ("spread, ~equal", [1024] * 7 + [1023]), ("weighted", w), ("default", None), ("weighted", w)):
            c.set_sym_xcd_weights(ws)
            print(n, name, ws, "ms/step %.4f %.4f" % (run(c, n, steps), run(c, n, steps)), "plan", c.sym_plan().parts, c.sym_plan().taper1, c.sym_plan().sets, flush=True)
