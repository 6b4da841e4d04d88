"""The one-sided launch of a FROZEN rank in a partially active sharded step (active x its frozen bodies; enqueue_sym_shard_split) over a sweep
of launch shapes (MAPN_FROZEN_PLAN hook: bodies per lane, waves per workgroup, rows), one rank in loopback: ms per step, wall clock.
python tools/shard_frozen_plan_sweep.py [N WORLD]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_P2P_LOOPBACK"] = "1"
import mapn
n, world = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65536, 8)
for na in (n // 2, 5 * n // 8):
    for plan in ("", "2,8,8", "2,8,4", "2,8,2", "2,16,8", "2,16,4", "2,16,2", "4,8,16", "4,8,8", "4,8,4", "4,8,2", "4,16,4", "4,16,2", "8,8,4", "8,8,2", "8,4,4", "2,4,16", "4,4,8"):
        if plan: os.environ["MAPN_FROZEN_PLAN"] = plan
        else: os.environ.pop("MAPN_FROZEN_PLAN", None)
        with mapn.Compute(n, device=0, mass=70000.0 / n, rank=world - 1, world_size=world) as c:
            blob = c.p2p_export(); c.p2p_import([blob] * world); c.set_gather_algorithm(5); c.set_timers(0)
            for _ in range(300): c.Simulate(na, c.GetFenceValue())
            c.WaitForGpu(); best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(1000): c.Simulate(na, c.GetFenceValue())
                c.WaitForGpu(); best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
            sp, _ = c.split_plan()
            print(f"N={n} / {world}, rank {world - 1} (frozen), active={na}: frozen plan {plan or 'default'} -> k={sp.frozen_bodies_per_lane} waves={sp.frozen_waves} sb={sp.frozen_sb}: {best:.2f} us per step", flush=True)
