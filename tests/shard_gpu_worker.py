"""One rank of tests/test_shard_gpu_multiproc.py (run as a script, one process per rank).

How the ranks find each other is NOT what these tests are about (the product's exchanges run on the device: hipIpc-mapped buffers, device
flags): the few host-side collectives a test needs -- hand round the hipIpc blobs, start together, compare checksums, collect the
velocities -- go through FILES in the test's directory (FileGroup) since round 6, which spares every one of ~150 rank processes per
suite run the import of torch.  Mode "external" keeps torch.distributed over gloo -- imported BEFORE the library, so that the
process holds a single HIP runtime (torch's), exactly like bench.py under torch.distributed.run; the bench tests cover that combination too."""
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FileGroup:
    """all_gather / barrier between the rank processes of ONE test through files in its directory: every call writes rank-numbered files
    (atomically: write + rename) and polls for the others'.  A rank that dies leaves a marker, so the others fail at once instead of
    waiting out the time limit."""

    def __init__(self, root, rank, world, timeout=600.0):
        self.dir, self.rank, self.world, self.timeout, self.seq = os.path.join(root, "_rendezvous"), rank, world, timeout, 0
        os.makedirs(self.dir, exist_ok=True)

    def all_gather(self, obj):
        self.seq += 1
        mine = os.path.join(self.dir, f"{self.seq}_{self.rank}")
        with open(mine + ".tmp", "wb") as f:
            pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(mine + ".tmp", mine + ".pkl")
        out, t0 = [], time.monotonic()
        for r in range(self.world):
            path = os.path.join(self.dir, f"{self.seq}_{r}.pkl")
            while not os.path.exists(path):
                if any(os.path.exists(os.path.join(self.dir, f"FAILED_{q}")) for q in range(self.world)):
                    raise RuntimeError(f"rank {self.rank}: another rank of this test failed")
                if time.monotonic() - t0 > self.timeout:
                    raise TimeoutError(f"rank {self.rank}: rank {r} never reached collective {self.seq}")
                time.sleep(0.0005)
            with open(path, "rb") as f:
                out.append(pickle.load(f))
        return out

    def barrier(self):
        self.all_gather(None)

    def failed(self):
        open(os.path.join(self.dir, f"FAILED_{self.rank}"), "w").close()

    def close(self):
        pass


class TorchGroup:
    """The same two collectives over torch.distributed (gloo)."""

    def __init__(self, rank, world, port):
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist, self.world = dist, world

    def all_gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        self.dist.barrier()

    def failed(self):
        pass

    def close(self):
        self.dist.destroy_process_group()


def main():
    rank, world, port, n, steps = (int(x) for x in sys.argv[1:6])
    out_dir = sys.argv[6]
    mode = sys.argv[7] if len(sys.argv) > 7 else "external"
    if mode == "external":
        import torch                       # noqa: F401  (first: its HIP runtime is then the one libmapn binds to)
        grp = TorchGroup(rank, world, port)
    else:
        grp = FileGroup(out_dir, rank, world)
    try:
        run(grp, rank, world, n, steps, out_dir, mode)
    except BaseException:
        grp.failed()
        raise


def p2p_setup(c, grp):
    """mapn_p2p_export on every rank, the blobs handed round, mapn_p2p_import (compute.py's p2p_setup_torch does the same over torch.distributed)."""
    c.p2p_import(grp.all_gather(c.p2p_export()))


def run(grp, rank, world, n, steps, out_dir, mode):
    import mapn
    c = mapn.Compute(n, device=0, mass=70000.0 / n, rank=rank, world_size=world)
    first, count = c.shard_range()
    if os.environ.get("MAPN_WORKER_PLAN"):          # "kernel,k,waves,sb,fused": keep every rank's launch small enough to be co-resident
        kn, k, w, sb, fu = os.environ["MAPN_WORKER_PLAN"].split(",")
        c.set_force_plan({"lds": mapn.KERNEL_LDS, "sgpr": mapn.KERNEL_SCALAR}[kn], int(k), int(w), int(sb), int(fu))
    if mode == "p2p_timeout":
        # rank 0 steps once, the other ranks never do: rank 0's wait for a peer's slice must give up
        # after the configured bound and SURFACE it (MAPN_ERR_COMM naming the peer), not pass silently
        p2p_setup(c, grp)
        c.set_gather_algorithm(2)
        c.set_timeouts(p2p_ms=100)
        if rank == 0:
            c.Simulate(n, c.GetFenceValue())
            try:
                c.WaitForGpu()
                raise AssertionError("a timed-out peer-to-peer wait went unreported")
            except mapn.MapnError as e:
                assert e.status == -4 and "rank 1" in str(e), str(e)
            assert c.p2p_status() == 2
            try:
                c.Simulate(n, c.GetFenceValue())
                raise AssertionError("simulate after a timed-out exchange must fail")
            except mapn.MapnError as e:
                assert e.status == -4
            open(os.path.join(out_dir, "timeout_reported"), "w").write("ok")
        grp.barrier()
        c.close()
        grp.close()
        return
    if mode == "sympush_corrupt":
        # gather algorithm 5 between two processes: rank 1's THIRD publication carries one flipped bit in one pushed position (test
        # hook); rank 0 must find it -- MAPN_ERR_COMM naming rank 1 -- when it next touches its replica, rank 1 itself is fine
        if rank == 1:
            os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_TEST_CORRUPT_PUSH"] = "3"
        p2p_setup(c, grp)
        c.set_gather_algorithm(5)
        c.set_timeouts(p2p_ms=5000)
        grp.barrier()
        for _ in range(3):
            c.Simulate(n, c.GetFenceValue())
        if rank == 0:
            try:
                c.WaitForGpu()
                raise AssertionError("a corrupted pushed position went unreported")
            except mapn.MapnError as e:
                assert e.status == -4 and "rank 1" in str(e) and "PUSHED" in str(e), str(e)
            assert c.p2p_status() == 0x200 + 1
            open(os.path.join(out_dir, "corruption_reported"), "w").write("ok")
        else:
            c.WaitForGpu()
            assert c.p2p_status() == 0
        grp.barrier()
        c.close()
        grp.close()
        return
    if mode == "symrow_corrupt":
        # ADVICE r4 (medium) / VERDICT r5 #1: a rank whose reaction rows never validate must not integrate and publish as if they had.
        # Rank 1 flips one bit of ONE row of exchange 3 after its tag was formed (test hook; the row is one it sends to itself, for its
        # body 5): its receiving thread waits out the bound, rank 1 reports and POISONS its position counter at rank 0, whose wait for
        # rank 1's slice gives up as well: BOTH ranks report.  And nothing of that body was published: what the buffers hold afterwards
        # is read past the (refusing) download entry points, straight from the exported device pointers, and compared with what they
        # held before the step.
        import ctypes as C
        algo = int(sys.argv[8])
        if rank == 1:
            os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_TEST_CORRUPT_ROW"] = "3"
        p2p_setup(c, grp)
        c.set_gather_algorithm(algo)
        c.set_timeouts(p2p_ms=400)
        handles = c.GetSharedHandles(consumer_fence=False)
        hip = None
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):     # (already in the process: libmapn is linked against it)
            try:
                hip = C.CDLL(name)
                break
            except OSError:
                continue
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def raw_positions(index):
            got = np.empty((n, 4), np.float32)
            assert hip.hipDeviceSynchronize() == 0
            assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), C.c_void_p(handles.positions[index]), got.nbytes, 2) == 0
            return got
        grp.barrier()
        for _ in range(2):
            c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu()
        w = c.buffer_index                                  # the buffer step 3 writes: it holds step 1's positions
        before = c.download_buffer(w)[0]
        healthy_latest = c.download_state()[0]
        assert np.array_equal(raw_positions(w), before)     # (the raw read sees what the library's download sees)
        grp.barrier()
        failed = None
        try:
            c.Simulate(n, c.GetFenceValue())
            c.WaitForGpu()
        except mapn.MapnError as e:
            failed = e
        assert failed is not None and failed.status == -4, f"rank {rank}: the failed exchange went unreported"
        if rank == 1:
            assert "never arrived whole" in str(failed) and 0x100 <= c.p2p_status() < 0x200, (str(failed), c.p2p_status())
        else:
            assert "rank 1" in str(failed) and c.p2p_status() == 2, (str(failed), c.p2p_status())
        grp.barrier()                                       # both ranks have given up: nothing is in flight any more
        after = raw_positions(w)
        bad = count + 5                                      # rank 1's body 5 in the whole job
        others = np.ones(n, bool); others[bad] = False
        mine = slice(first, first + count)
        assert np.array_equal(after[bad], before[bad]), f"rank {rank}: the body whose rows never validated was published: {after[bad]} (held {before[bad]})"
        moved = (after[:, :3] != before[:, :3]).any(axis=1)
        if rank == 1:
            # the failing rank: every other body of its slice was integrated from validated rows and stored
            assert moved[mine][np.arange(count) != 5].all()
        elif algo == 4:
            # PULLED positions: rank 0 never pulled (its wait for rank 1's poisoned counter gave up) -- NOTHING of rank 1's slice landed
            assert np.array_equal(after[count:], before[count:]), "rank 0 pulled a slice whose counter was poisoned"
            assert moved[mine].all()
        else:
            # PUSHED positions: rank 1's healthy threads pushed their (correct) bodies long before the failing one gave up; the failed body
            # did not land, and the poisoned counter keeps rank 0 from consuming the slice
            assert moved[count:][np.arange(count) != 5].all() and moved[mine].all()
        # what DID land of rank 1's slice is what a healthy step gives: one more step of the oracle-checked trajectory, i.e. within a
        # step's displacement bound of the healthy latest state (|v| dt <= ~15 * 0.1 + kick), never garbage
        disp = np.linalg.norm(after[others, :3].astype(np.float64) - healthy_latest[others, :3], axis=1)
        landed = moved & others
        assert disp[landed[others]].max() < 5.0, disp.max()
        open(os.path.join(out_dir, f"row_failure_reported_by_rank{rank}"), "w").write(str(failed))
        grp.barrier()
        c.close()
        grp.close()
        return
    mixed = mode.endswith("_mixed")                     # every third step freezes part of the bodies: the step then runs one-sided
    if mixed:
        mode = mode[:-len("_mixed")]
    if mode in ("p2p", "flow", "sym", "sympush"):
        # the in-library direct exchange: hipIpc-mapped peer buffers + device flags, no caller help;
        # "flow" = the same exchange overlapped inside the force launch (gather algorithm 3)
        p2p_setup(c, grp)
        # "sym" = the symmetric step sharded over the ranks (gather algorithm 4): reactions stored into the
        # owners' receive regions, positions pulled as in "p2p"
        # "sympush" = the same with the new positions PUSHED into the peers' replicas by the exchange launch (gather algorithm 5)
        c.set_gather_algorithm({"p2p": 2, "flow": 3, "sym": 4, "sympush": 5}[mode])
        if mode in ("sym", "sympush") and count % 1024 == 0 and not os.environ.get("MAPN_SYM_SHARD_PLAN"):
            # the ranks share ONE device here (mapn_p2p_import compares their PCI ids): the plan must be the 4-wave one -- an
            # 8-wave workgroup needs a whole compute unit's registers and cannot be placed beside a peer's waiting exchange launch
            assert c.sym_plan().waves == 4 and c.sym_plan().wave_bias == (1, 1), (c.sym_plan().waves, c.sym_plan().wave_bias)
        if os.environ.get("MAPN_WORKER_XCD_W") and mode in ("sym", "sympush"):
            c.set_sym_xcd_weights([int(x) for x in os.environ["MAPN_WORKER_XCD_W"].split(",")])
            assert (c.sym_plan().xcd_mode != 0) == ((count // 1024) % 8 == 0)
        if os.environ.get("HSA_CU_MASK") and mode in ("sym", "sympush") and count % 1024 == 0:
            plan = c.sym_plan()                 # the probe must have seen the mask, and the launch must be sized for it
            assert plan.active_compute_units <= 64 and plan.exchange_workgroups <= 4 * plan.active_compute_units, (plan.active_compute_units, plan.exchange_workgroups)
        # several processes time-slice ONE GPU here: be generous (the big jobs allocate GBs of scratch inside their
        # first step, one process after the other)
        c.set_timeouts(p2p_ms=5000 if n <= 65536 else 60000)
        num_active = int(sys.argv[8]) if len(sys.argv) > 8 else n
        grp.barrier()                   # start stepping together: process start-up skews by seconds on a cold box
        slider = os.environ.get("MAPN_WORKER_SLIDER")      # "a,b,c,...": num_active of step k = the k-th entry (cyclic) -- the slider dragged on a sharded job
        counts = [int(x) for x in slider.split(",")] if slider else None
        for k in range(steps):
            na = counts[k % len(counts)] if counts else (num_active if not (mixed and k % 3 == 1) else num_active // 2 + 100)
            c.Simulate(na, c.GetFenceValue())
        c.WaitForGpu()
        assert c.p2p_status() == 0, f"p2p wait timed out: status {c.p2p_status()}"
        # a PARTIALLY ACTIVE step of the sharded symmetric forms takes the split form (round 6) where mapn_shard_split_describe says so
        split_form = False
        if mode in ("sym", "sympush") and count % 1024 == 0 and num_active < n and not os.environ.get("MAPN_SHARD_PARTIAL_FORM"):
            from mapn.compute import describe_shard_split
            split_form = bool(describe_shard_split(n, rank, world, num_active).applies)
        if mode in ("sym", "sympush") and not mixed and not counts:
            st = c.kernel_stats()
            want = "force_sym_kernel" if ((num_active == n or split_form) and count % 1024 == 0) else "force_sgpr_kernel"
            assert st.kernel_name.decode() == want, st.kernel_name
            assert (st.split_active != 0) == split_form, (st.split_active, split_form)
        pos, vel = c.download_state()
        other = c.download_buffer(c.buffer_index)[0]
        if split_form and not mixed and not counts:
            # this rank's part of the step, for the order-matched restatement (oracle: ORDER_MATCHED_SHARDED_SPLIT) -- and what the device-less
            # description promised must be what the context ran
            role = describe_shard_split(n, rank, world, num_active)
            split, pl = c.split_plan()
            assert (split.active, split.frozen, split.frozen_first, split.has_plan) == (role.active, role.frozen_count, role.frozen_first, 1 if role.blocks else 0), (split.active, split.frozen)
            if role.blocks:
                assert (pl.nb, pl.a0, pl.nbl) == (role.ring_blocks, role.first_block, role.blocks), (pl.nb, pl.a0, pl.nbl)
                np.savez(os.path.join(out_dir, f"split_rank{rank}.npz"), windows=pl.windows, tables=pl.tables, frozen=np.array([split.frozen_waves, split.frozen_sb], np.uint32),
                         shape=np.array([pl.nb, pl.groups, pl.parts, pl.waves, pl.brows, pl.max_meetings, pl.table_stride, pl.sets, pl.a0, pl.nbl], np.uint32))
            else:
                np.savez(os.path.join(out_dir, f"split_rank{rank}.npz"), frozen=np.array([split.frozen_waves, split.frozen_sb], np.uint32), shape=np.zeros(10, np.uint32))
            # the frozen bodies: untouched in BOTH buffers of this rank's replica (the seeded state put them there)
            pos0, _ = mapn.generate_initial_state(n, seed=1)
            a = role.active
            assert np.array_equal(pos[a:], pos0[a:]) and np.array_equal(other[a:], pos0[a:]), f"rank {rank}: a frozen body moved"
        if mode in ("sym", "sympush") and not mixed and not counts and num_active == n and count % 1024 == 0:
            # this rank's launch plan, for the order-matched restatement of the sharded step (oracle: ORDER_MATCHED_SHARDED)
            pl = c.sym_plan()
            np.savez(os.path.join(out_dir, f"plan_rank{rank}.npz"), windows=pl.windows, tables=pl.tables,
                     shape=np.array([pl.nb, pl.groups, pl.parts, pl.waves, pl.brows, pl.max_meetings, pl.table_stride, pl.sets, pl.a0, pl.nbl], np.uint32))
        # every replica must hold the same positions, bit for bit
        sums = grp.all_gather((int(np.frombuffer(pos.tobytes(), np.uint32).sum(dtype=np.uint64)),
                               int(np.frombuffer(other.tobytes(), np.uint32).sum(dtype=np.uint64))))
        assert all(s == sums[0] for s in sums), f"position replicas differ across ranks: {sums}"
        fullv = np.concatenate(grp.all_gather(vel[first:first + count].copy()))
        if rank == 0:
            np.savez(os.path.join(out_dir, "gpu_sharded.npz"), pos=pos, vel=fullv, other=other)
        c.close()
        grp.barrier()
        grp.close()
        return
    c.set_external_gather(True)
    for _ in range(steps):
        c.Simulate(n, c.GetFenceValue())
        pos, vel = c.download_state()                       # latest buffer: own slice is fresh
        full = np.concatenate(grp.all_gather(pos[first:first + count].copy()))
        # the caller's all-gather: upload_state writes both ping-pong buffers, which is exact for a
        # continuation with num_active == n (every body is rewritten by the next step anyway)
        c.upload_state(full, vel)
    pos, vel = c.download_state()
    fullv = np.concatenate(grp.all_gather(vel[first:first + count].copy()))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gpu_sharded.npz"), pos=pos, vel=fullv)
    c.close()
    grp.barrier()
    grp.close()


if __name__ == "__main__":
    main()
