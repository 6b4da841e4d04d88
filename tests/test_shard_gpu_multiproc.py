"""Two real processes sharing ONE GPU, each owning half of the bodies (world_size 2): every rank
advances its slice on the device through the C ABI, the slices are exchanged with a gloo
all-gather (RCCL refuses two ranks on one device, so the in-library RCCL exchange itself is
covered by the 1-rank test in test_gpu_parity.py), and the composed trajectory is compared with
the CPU oracle.  Covers the per-rank slicing, the frozen tail and velocity locality of the sharded
device step in a true one-process-per-rank setting."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, steps, num_active, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import mapn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = mapn.Compute(n, device=0, mass=70000.0 / n, rank=rank, world_size=world)
    c.set_external_gather(True)
    first, count = c.shard_range()
    for _ in range(steps):
        c.Simulate(num_active, c.GetFenceValue())
        pos, vel = c.download_state()                       # latest buffer: own slice is fresh
        mine = torch.from_numpy(pos[first:first + count].copy())
        full = torch.empty((n, 4), dtype=torch.float32)
        dist.all_gather_into_tensor(full, mine)
        # the caller's all-gather: upload_state writes both ping-pong buffers, which is exact for a
        # continuation with num_active == n (every body is rewritten by the next step anyway)
        c.upload_state(full.numpy(), vel)
    pos, vel = c.download_state()
    mine_v = torch.from_numpy(vel[first:first + count].copy())
    fullv = torch.empty((n, 3), dtype=torch.float32)
    dist.all_gather_into_tensor(fullv, mine_v)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gpu_sharded.npz"), pos=pos, vel=fullv.numpy())
    c.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_one_gpu_sharded_device_steps(tmp_path, oracle):
    import torch.multiprocessing as mp
    from oracle import OracleSim, Params
    n, steps, world = 4096, 3, 2
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n, steps, n, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "gpu_sharded.npz"))
    pos0, vel0 = oracle.initial_state(n, seed=1)
    sim = OracleSim(oracle, pos0, vel0, params=Params(mass=70000.0 / n))
    sim.simulate(steps=steps)
    dx = np.linalg.norm(got["pos"][:, :3].astype(np.float64) - sim.latest[0][:, :3], axis=1).max() / 400.0
    dv = np.linalg.norm(got["vel"].astype(np.float64) - sim.latest[1], axis=1).max() / 15.0
    assert dx < 3e-6 and dv < 6e-5, (dx, dv)
