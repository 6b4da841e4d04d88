import os, sys
os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_P2P_LOOPBACK"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mapn
n, world = 16384, 2
with mapn.Compute(n, device=0, mass=70000.0 / n, rank=0, world_size=world) as c:
    blob = c.p2p_export(); c.p2p_import([blob] * world); c.set_gather_algorithm(5)
    pl = c.sym_plan()
    for _ in range(20): c.Simulate(n, c.GetFenceValue())
    c.WaitForGpu()
    print(os.environ.get("HSA_CU_MASK"), os.environ.get("ROC_GLOBAL_CU_MASK"), "active CUs", pl.active_compute_units, "exchange cap", pl.exchange_workgroups, "status", c.p2p_status())
