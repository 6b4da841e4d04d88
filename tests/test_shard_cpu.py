"""Multi-rank logic on CPU: world_size-2 gloo processes shard the bodies, each advances its own
slice (with the oracle -- tests may use it as the compute), all-gathers the new float4 position
slices, and the composed trajectory must equal the single-rank one bit for bit.  This pins the
host-side decomposition the GPU path uses (ShardPlan == csrc/mapn_context.cpp's slicing)."""
import os
import sys

import numpy as np
import pytest

import mapn
from shard_model import ShardPlan, active_bodies, remote_segments, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_bodies():
    for n, p in [(65536, 8), (1048576, 8), (4096, 2), (96, 3), (7, 1)]:
        cover = []
        for r in range(p):
            first, count = shard_range(n, r, p)
            cover += list(range(first, first + count))
            segs = remote_segments(n, r, p)
            assert sum(c for _, c in segs) == n - count
            assert segs[0] == (0, first) and segs[1][0] == first + count
        assert cover == list(range(n))
    with pytest.raises(ValueError):
        shard_range(100, 0, 3)
    with pytest.raises(ValueError):
        shard_range(100, 2, 2)


def test_active_slice_clamps_per_rank():
    plan = [ShardPlan(1024, r, 4) for r in range(4)]
    assert [p.active_slice(1024) for p in plan] == [(0, 256), (256, 256), (512, 256), (768, 256)]
    assert [p.active_slice(300) for p in plan] == [(0, 256), (256, 64), (512, 0), (768, 0)]     # roundup64(300) = 320
    assert [p.active_slice(0) for p in plan] == [(0, 0), (256, 0), (512, 0), (768, 0)]
    assert active_bodies(300, 1024) == 320
    assert plan[1].gather_bytes_sent() == 256 * 16 and plan[1].gather_bytes_received() == 768 * 16


def _worker(rank, world, port, n, steps, num_active, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import Oracle, Params
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    prm = Params(mass=70000.0 / n)
    pos0, vel0 = o.initial_state(n, seed=1)
    plan = ShardPlan(n, rank, world)
    # full position replica + full-size velocity array of which only the own slice is maintained
    pos = [pos0.copy(), pos0.copy()]
    vel = [vel0.copy(), vel0.copy()]
    idx = 0
    for _ in range(steps):
        w, r = idx, 1 - idx
        first, count = plan.active_slice(num_active)
        if count:
            p, v = o.step_slice(pos[r], vel[r], first, count, params=prm, threads=2)
            pos[w][first:first + count] = p
            vel[w][first:first + count] = v
        # all-gather of the WRITTEN buffer's slices (frozen bodies ride along unchanged)
        mine = torch.from_numpy(pos[w][plan.first:plan.first + plan.count].copy())
        full = torch.empty((n, 4), dtype=torch.float32)
        dist.all_gather_into_tensor(full, mine)
        pos[w][:] = full.numpy()
        idx = 1 - idx
    # velocities: gather once for the check
    mine = torch.from_numpy(vel[1 - idx][plan.first:plan.first + plan.count].copy())
    fullv = torch.empty((n, 3), dtype=torch.float32)
    dist.all_gather_into_tensor(fullv, mine)
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), pos=pos[1 - idx], vel=fullv.numpy(), other=pos[idx])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_active", [1024, 600])
def test_two_rank_gloo_sharded_steps_equal_single_rank(tmp_path, oracle, num_active):
    import torch.multiprocessing as mp
    from oracle import OracleSim, Params
    n, steps, world = 1024, 3, 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n, steps, num_active, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "sharded.npz"))
    pos0, vel0 = oracle.initial_state(n, seed=1)
    sim = OracleSim(oracle, pos0, vel0, params=Params(mass=70000.0 / n))
    sim.simulate(num_active=num_active, steps=steps)
    np.testing.assert_array_equal(got["pos"], sim.latest[0])
    np.testing.assert_array_equal(got["other"], sim.pos[sim.buffer_index])
    na = active_bodies(num_active, n)
    np.testing.assert_array_equal(got["vel"][:na], sim.latest[1][:na])
