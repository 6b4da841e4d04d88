"""Multi-rank logic on CPU: world_size-2 gloo processes shard the bodies, each advances its own
slice (with the oracle -- tests may use it as the compute), all-gathers the new float4 position
slices, and the composed trajectory must equal the single-rank one bit for bit.

Which bodies a rank owns and which of them a step advances comes from the PRODUCT -- `mapn_shard_describe`, the device-less
entry point over the very helpers `mapn_create` and the step use (csrc/mapn_internal.h: shard_slice, shard_active_count;
active_bodies; sym_shard_masks) -- so the run is evidence about the library's host arithmetic (until round 5
it exercised tests/shard_model.py, a Python model, alone).  The model stays as a second opinion: the two must agree on every shape."""
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


def test_product_shard_arithmetic_agrees_with_the_model_and_with_itself_across_ranks():
    """mapn_shard_describe (product, no device) against tests/shard_model.py on many shapes: slices, active slices, and -- where the
    sharded symmetric step applies -- masks that are each other's mirror (q in send(r)  <=>  r in recv(q)), every rank sending to
    itself, and block ranges that tile the job."""
    from mapn.compute import describe_shard
    import shard_model as shard
    for n, world in [(65536, 8), (1048576, 8), (8192, 2), (9216, 3), (16384, 8), (6144, 2), (4096, 2), (96, 3), (7, 1), (262144, 4)]:
        infos = [describe_shard(n, r, world, na) for r in range(world) for na in (n,)]
        for na in (n, n // 2 + 100, 64, 1, 0, -5, 10 * n):
            for r in range(world):
                i = describe_shard(n, r, world, na)
                assert (i.first, i.count) == shard_range(n, r, world)
                assert (i.active_first, i.active_count) == ShardPlan(n, r, world).active_slice(na), (n, world, r, na)
        applies = world >= 2 and (n // world) % 1024 == 0
        assert all(bool(i.sym_applies) == applies for i in infos)
        if applies:
            assert [i.a0 for i in infos] == [r * (n // world // 1024) for r in range(world)] and all(i.nb == n // 1024 and i.nbl == n // world // 1024 for i in infos)
            for r, i in enumerate(infos):
                assert (i.send_mask >> r) & 1 and (i.recv_mask >> r) & 1          # a rank's own blocks meet each other
                for q, j in enumerate(infos):
                    assert ((i.send_mask >> q) & 1) == ((j.recv_mask >> r) & 1), (n, world, r, q)
            # the masks are what the meeting schedule says (the model's restatement of it)
            for r, i in enumerate(infos):
                send, recv = shard.sym_shard_masks(n // 1024, world, r)
                assert (send, recv) == (i.send_mask, i.recv_mask)
    with pytest.raises(mapn.MapnError):
        describe_shard(100, 0, 3)
    with pytest.raises(mapn.MapnError):
        describe_shard(100, 2, 2)


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
    from mapn.compute import describe_shard
    plan = describe_shard(n, rank, world)                  # the PRODUCT's slicing (no device needed)
    assert (plan.first, plan.count) == (ShardPlan(n, rank, world).first, ShardPlan(n, rank, world).count)
    # full position replica + full-size velocity array of which only the own slice is maintained
    pos = [pos0.copy(), pos0.copy()]
    vel = [vel0.copy(), vel0.copy()]
    idx = 0
    for _ in range(steps):
        w, r = idx, 1 - idx
        step_info = describe_shard(n, rank, world, num_active)     # ... and which of the rank's bodies THIS step advances
        first, count = step_info.active_first, step_info.active_count
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


def _split_roles(n, world, num_active):
    from mapn.compute import describe_shard_split
    return [describe_shard_split(n, r, world, num_active) for r in range(world)]


def test_partially_active_sharded_step_roles_tile_the_job_and_the_masks_mirror_each_other():
    """mapn_shard_split_describe (product, no device): the active bodies form a ring of their own whose blocks are dealt to their owners in
    order; every frozen body is owned -- and its force computed -- by exactly one rank; the masks are each other's mirror and hold exactly
    (a) the ring's meetings between the ranks' blocks and (b) frozen owner -> active owner."""
    import shard_model as shard
    for n, world, na in [(65536, 8, 32768), (65536, 8, 40960), (65536, 8, 33000), (65536, 4, 20000), (8192, 2, 4096), (8192, 4, 5120), (8192, 4, 5000),
                         (16384, 8, 2048), (16384, 8, 16320), (9216, 3, 4096), (65536, 2, 32768), (1048576, 8, 524288 + 64)]:
        roles = _split_roles(n, world, na)
        A, count, cblk = active_bodies(na, n), n // world, n // world // 1024
        assert all(r.applies == 1 and r.active == A for r in roles), (n, world, na)
        nba = (A + 1023) // 1024
        assert all(r.ring_blocks == nba for r in roles)
        # the ring's blocks, in order, without gaps; the active bodies likewise
        blocks = [b for r in roles for b in range(r.first_block, r.first_block + r.blocks)]
        assert blocks == list(range(nba)), (n, world, na, blocks)
        assert sum(r.active_count for r in roles) == A and sum(r.frozen_count for r in roles) == n - A
        for q, r in enumerate(roles):
            assert r.first_block == q * cblk and r.blocks == (r.active_count + 1023) // 1024
            assert r.active_count == ShardPlan(n, q, world).active_slice(na)[1]
            assert r.frozen_first == q * count + r.active_count and r.frozen_first + r.frozen_count == (q + 1) * count
        for a, ra in enumerate(roles):
            for b, rb in enumerate(roles):
                assert ((ra.send_mask >> b) & 1) == ((rb.recv_mask >> a) & 1), (n, world, na, a, b)
                ring_send, _ = shard.sym_shard_masks(nba, world, a, blocks_per_rank=cblk) if nba else (0, 0)
                want = bool((ring_send >> b) & 1) or (ra.frozen_count > 0 and rb.active_count > 0)
                assert bool((ra.send_mask >> b) & 1) == want, (n, world, na, a, b)
    # where the form does not apply: all bodies active, a handful, nothing
    for na in (65536, 70000, 1024, 64, 0):
        assert all(r.applies == 0 for r in _split_roles(65536, 8, na))
    with pytest.raises(mapn.MapnError):
        _split_roles(6144, 4, 3000)                              # a rank's slice must be whole blocks


@pytest.mark.parametrize("n,world,num_active", [(8192, 4, 5120), (8192, 4, 4500), (8192, 2, 4096), (12288, 3, 7168), (16384, 8, 9216)])
def test_sharded_split_restatement_follows_the_reference_order_oracle(oracle, n, world, num_active):
    """The order-matched restatement of the PARTIALLY ACTIVE sharded step (oracle: ORDER_MATCHED_SHARDED_SPLIT), driven entirely by what the
    PRODUCT reports without a device -- every rank's roles (mapn_shard_split_describe) and the plan of its blocks in the active ring
    (mapn_sym_plan_describe) -- against the oracle proper: another summation order of the same pair terms, so one step agrees to rounding
    (a row sent to the wrong rank, a frozen body counted twice or not at all, a block missing from the ring would show at 1e-3 and more),
    and the frozen bodies come back bit for bit.  Shapes: whole ranks active + whole ranks frozen, a rank that is both, a count that is
    not whole blocks, one block per rank."""
    from mapn.compute import describe_sym_plan
    from oracle import OracleSim, Params, step_sym_sharded_split
    roles = _split_roles(n, world, num_active)
    A = roles[0].active
    plans, frozen = [], []
    for r in roles:
        plans.append(describe_sym_plan(r.ring_blocks, parts=8, waves=4, launch_blocks=r.blocks, launch_a0=r.first_block) if r.blocks else None)
        frozen.append((4, 2) if r.frozen_count else None)
    pos, vel = oracle.initial_state(n, seed=3)
    prm = Params(mass=70000.0 / n)
    p, v = step_sym_sharded_split(oracle, pos, vel, prm, A, plans, frozen)
    sim = OracleSim(oracle, pos, vel, params=prm); sim.simulate(num_active=num_active)
    rp, rv = sim.latest
    rel = np.linalg.norm(p[:A, :3].astype(np.float64) - rp[:A, :3], axis=1) / 400.0
    assert rel.max() < 1e-6, rel.max()
    assert np.linalg.norm(v[:A].astype(np.float64) - rv[:A], axis=1).max() / 15.0 < 2e-5
    np.testing.assert_array_equal(p[A:], pos[A:]); np.testing.assert_array_equal(v[A:], vel[A:])
    # ... and the force on an active body does not depend on how the frozen bodies' work is cut (another launch shape: same sum to rounding)
    p2, _ = step_sym_sharded_split(oracle, pos, vel, prm, A, plans, [(8, 1) if f else None for f in frozen])
    assert np.linalg.norm(p2[:A, :3].astype(np.float64) - rp[:A, :3], axis=1).max() / 400.0 < 1e-6
