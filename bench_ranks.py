"""bench_ranks.py -- the N > 1 side of bench.py: the launcher (`python bench.py --gpus N` without torch.distributed.run), the fallback
transport, and the exchange trial that picks -- in untimed steps, verified -- the step structure the timed region then runs.  One process
per GPU; the ranks agree on every decision through a collective (MAX / MIN over ranks)."""
from __future__ import annotations

import os
import sys
import time


def launch_ranks(n_ranks, script):
    """`python bench.py --gpus N` with no launcher environment: this process becomes the launcher.  It starts N children of
    this same command line (`script` = bench.py) -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way
    torch.distributed.run sets them -- passes their output through (rank 0 prints the ONE JSON line) and returns the worst
    exit code.  The parent never loads the library, never touches the GPU and never exec's (a process that has initialised the
    GPU must not be replaced).  A rank that dies takes the others down after a grace period instead of leaving them in a
    collective for ever.  (Particles.cpp:446-448 is the caller sequence every rank then runs.)"""
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks), "GROUP_RANK": "0",
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # the host driver supports dmabuf IPC only (hipIpc handles, RCCL)
        env.setdefault("OMP_NUM_THREADS", "1")
        children.append(subprocess.Popen([sys.executable, script] + sys.argv[1:], env=env, start_new_session=True))
    worst, deadline = 0, None
    try:
        while any(ch.poll() is None for ch in children):
            codes = [ch.poll() for ch in children]
            if deadline is None and any(rc not in (None, 0) for rc in codes):
                deadline = time.monotonic() + 20.0             # a rank failed: the others get 20 s to notice (bounded waits, barriers)
            if deadline is not None and time.monotonic() > deadline:
                for ch in children:
                    if ch.poll() is None:
                        os.killpg(ch.pid, signal.SIGKILL)      # exactly the process groups started above
                deadline = float("inf")
            time.sleep(0.05)
    except KeyboardInterrupt:
        for ch in children:
            if ch.poll() is None:
                os.killpg(ch.pid, signal.SIGKILL)
        worst = 130
    for r, ch in enumerate(children):
        rc = ch.wait()
        if rc != 0:
            print(f"[bench launcher] rank {r} exited with code {rc}", file=sys.stderr, flush=True)
            worst = worst or (rc if rc > 0 else 128 - rc)
    return worst


def make_torch_gather(c, torch, dist, n, rank, world):
    """Fallback transport: torch.distributed (RCCL) all-gather on zero-copy views of the
    library's position buffers."""
    h = c.GetSharedHandles(consumer_fence=False)

    class _View:
        def __init__(self, ptr):
            self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<f4", "data": (int(ptr), False), "version": 2}

    bufs = [torch.as_tensor(_View(h.positions[i]), device="cuda") for i in range(2)]
    count = n // world

    def gather():
        c.WaitForGpu()
        w = 1 - c.buffer_index                     # buffer the step just wrote
        dist.all_gather_into_tensor(bufs[w], bufs[w][rank * count:(rank + 1) * count].clone())
        torch.cuda.synchronize()

    return gather



class RankFailure(RuntimeError):
    """A failure the ranks have agreed on (Job.together): raised on every rank of the job at the same point."""


class Job:
    """One rank's side of a bench run: the context (replaced when a device-side check leaves it failed), the process group, and -- N > 1 --
    every decision the ranks take together: which exchange structure runs (exchange_trial), whether the die weights stay (sharded_xcd_ab),
    what takes over when the kept form fails in the run that counts (fall_back)."""

    P2P_ONLY = ("p2p", "flow", "sym", "sympush", "p2pall")     # --gather values that need no RCCL communicator

    def __init__(self, a, mapn, torch, dist, rank, world, local_rank, mode, flags, kern):
        self.a, self.mapn, self.torch, self.dist = a, mapn, torch, dist
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.n, self.mode, self.flags, self.kern = a.bodies, mode, flags, kern
        self.red_dev = "cuda" if a.dist_backend == "nccl" else "cpu"     # where small reduction tensors live
        self.c = None
        self.gather_fn = None
        self.transport = "none"
        self.gather_algo = "n/a"
        self.trial, self.candidates = {}, []
        self.p2p_ok, self.p2p_failure, self.fallback_after_failure = False, None, None
        self.timer_interval = 0

    # -- the context -----------------------------------------------------------------------------------------------------------
    def create(self):
        a = self.a
        self.c = self.mapn.Compute(self.n, device=self.local_rank, force_mode=self.mode, mass=70000.0 / self.n, seed=a.seed,
                                   rank=self.rank, world_size=self.world, flags=self.flags, kernel=self.kern)
        return self.c

    def apply_plan(self):
        if self.a.plan:
            kname, k, w, sb, fused = self.a.plan.split(",")
            self.c.set_force_plan({"lds": self.mapn.KERNEL_LDS, "sgpr": self.mapn.KERNEL_SCALAR}[kname], int(k), int(w), int(sb), int(fused))

    def step(self):
        fence = self.c.GetFenceValue()             # Particles.cpp:446-448
        self.c.Simulate(self.n, fence)
        if self.gather_fn:
            self.gather_fn()

    def idle(self):
        """This rank's device idle.  NO collective in it."""
        self.c.WaitForGpu()
        if self.torch is not None:
            self.torch.cuda.synchronize()

    def run_idle(self, k):
        for _ in range(k):
            self.step()
        self.idle()

    def sync(self):
        """idle + barrier.  The barrier is taken even when this rank's wait reports a failure (raised behind it): the peers are in it."""
        err = None
        try:
            self.idle()
        except self.mapn.MapnError as e:
            err = e
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()
        if err is not None:
            raise err

    def together(self, fn):
        """fn() on every rank -- no collective inside it -- then ONE collective that is barrier and verdict: the failure's text on EVERY rank
        when any rank's library reported one, else None.  Whatever the ranks decide from the result they decide alike, so a check that fails
        on one rank (a pushed position against its checksum, a row that never validated, a wait that gave up) can never leave the others in
        a collective that rank does not take part in."""
        err = None
        try:
            fn()
        except self.mapn.MapnError as e:
            err = str(e) or "failed"
        if self.dist is None:
            return err
        if not self.all_reduce(1 if err else 0):
            return None
        texts = [None] * self.world
        self.dist.all_gather_object(texts, err)                # (only on the failure path; every rank takes it)
        return next(f"rank {r}: {t}" for r, t in enumerate(texts) if t)

    def rebuild(self, with_p2p):
        """A context whose device-side wait timed out or whose row / position check failed stays failed: replace it (collective: all ranks)."""
        self.c.close()
        self.create()                                  # (the SAME kernel choice and plan as asked for: ADVICE r2)
        if self.a.gather not in self.P2P_ONLY:
            self.c.comm_init_torch()
        if with_p2p:
            self.c.p2p_setup_torch()
            self.c.set_timeouts(p2p_ms=self.a.p2p_timeout_ms)
        self.apply_plan()
        self.c.set_timers(self.timer_interval)

    def setup_transport(self):
        """The in-library RCCL communicator (or, loudly, torch's all-gather on views of the library's buffers)."""
        a = self.a
        self.transport = a.transport
        if self.transport == "rccl" and a.gather not in self.P2P_ONLY:
            ok = 1
            try:
                self.c.comm_init_torch()
            except Exception as e:     # RCCL-in-library unavailable: use torch's RCCL instead, loudly
                print(f"[bench rank {self.rank}] native RCCL transport failed ({e}); falling back to torch.distributed all-gather",
                      file=sys.stderr, flush=True)
                ok = 0
            if not self.all_reduce(ok, "MIN"):                 # all ranks or none
                self.transport = "torch"
        if self.transport == "torch":
            self.c.set_external_gather(True)
            self.gather_fn = make_torch_gather(self.c, self.torch, self.dist, self.n, self.rank, self.world)

    def closing_collective(self):
        """What closes a timed region for N > 1, made BEFORE any timed region (bench.py: run_steps): the verdict word, and -- RCCL backend -- the
        pre-made device word of the stream-ordered barrier, the two events around a rank's own steps, and the library's compute stream
        as a torch stream (one wrapper per context: a rebuilt context has another stream)."""
        torch, dist = self.torch, self.dist
        nccl = dist is not None and self.a.dist_backend == "nccl"
        cache = {}

        def compute_stream_of(ctx):
            ptr = ctx.compute_stream
            if ptr not in cache:
                cache.clear()
                cache[ptr] = torch.cuda.ExternalStream(ptr, device=torch.device("cuda", self.local_rank))
            return cache[ptr]
        return {"verdict": torch.zeros(1, dtype=torch.int32, device=self.red_dev) if dist is not None else None,
                "barrier_word": torch.zeros(1, dtype=torch.int32, device="cuda") if nccl else None,
                "ev_first": torch.cuda.Event(enable_timing=True) if nccl else None, "ev_last": torch.cuda.Event(enable_timing=True) if nccl else None,
                "stream_of": compute_stream_of}

    # -- checks the ranks make together ----------------------------------------------------------------------------------------------
    def all_reduce(self, value, op="MAX", dtype=None):
        torch, dist = self.torch, self.dist
        t = torch.tensor([value], device=self.red_dev, **({"dtype": dtype} if dtype is not None else {}))
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return t.item()

    def replicas_consistent(self):
        """Every rank keeps a full replica of the positions; after any correct exchange they are
        bit-identical.  Compares a checksum of both ping-pong buffers across ranks."""
        try:
            sums = list(self.c.replica_checksum())       # mapn_replica_checksum: one C call (drains, reports failed device-side waits)
        except self.mapn.MapnError as e:                 # (every rank must still take part in the collective below)
            print(f"[bench rank {self.rank}] replica check: {e}", file=sys.stderr, flush=True)
            sums = None
        allsums = [None] * self.world
        self.dist.all_gather_object(allsums, sums)
        return all(x is not None and x == allsums[0] for x in allsums)

    def symmetric_deviation(self, sym_algo):
        """Four steps from the initial state with the sharded symmetric step against the one-sided sharded step over the same
        kind of transport (peer-to-peer pull for 4 / 5, RCCL all-gather for 6)."""
        import numpy as np
        c, got = self.c, []
        for algo in ((0 if sym_algo == 6 else 2), sym_algo):
            def fresh():
                c.set_gather_algorithm(algo)
                pos0, vel0 = self.mapn.generate_initial_state(self.n, seed=self.a.seed)
                c.upload_state(pos0, vel0)
                self.idle()
            # ALL ranks must have finished before anyone re-initialises: a rank that is a step ahead would overwrite the
            # buffers a slower peer is still pulling from / pushing into (seen with 8 ranks time-slicing one device): every stage
            # ends in `together`'s collective.  A failure on any rank: infinity on all of them (the caller's check fails, nobody hangs).
            def four():
                self.run_idle(4)
                got.append(c.download_state()[0][:, :3].astype(np.float64))
            if self.together(self.idle) or self.together(fresh) or self.together(four):
                return float("inf")
        return float(np.linalg.norm(got[0] - got[1], axis=1).max() / 400.0)

    def reinit(self):
        self.c.set_gather_algorithm(0 if self.a.gather not in self.P2P_ONLY else 2)
        self.c.set_shard_overlap(False)
        pos0, vel0 = self.mapn.generate_initial_state(self.n, seed=self.a.seed)
        self.c.upload_state(pos0, vel0)
        self.sync()

    def select(self, chosen):
        self.c.set_gather_algorithm(chosen[1])
        self.c.set_shard_overlap(chosen[2])
        self.transport = "p2p (hipIpc + device flags)" if 2 <= chosen[1] <= 5 else "rccl"

    # -- the exchange trial ------------------------------------------------------------------------------------------------------
    def build_candidates(self):
        """Every way of issuing the exchange that sets up on this node: (name, algorithm, overlap structure).
        The DEFAULT trial (--gather auto) holds the forms that can win, safest first: the RCCL collectives, the symmetric step over
        RCCL alone, then the peer-to-peer forms in the order they build on each other.  The OVERLAP structures (north_star's: own-segment
        launch beside the previous all-gather) measure SLOWER than their plain forms in loopback -- two under-filled launches and a
        cross-stream event pair: 152.6 against 116.2 us per step at 65 536 / 8, profiles/r05_shard_step_timeline.txt -- and the in-kernel
        exchange loses too: they are tried LAST and only when asked for by name (--overlap, --gather allgather / sendrecv / flow / p2pall):
        fewer code paths in the one run that counts."""
        a, mapn, world, n = self.a, self.mapn, self.world, self.n
        cands, overlaps = [], []
        if a.gather in ("auto", "allgather"):
            cands.append(("allgather", 0, False))
            if a.overlap or a.gather == "allgather":
                overlaps.append(("allgather+overlap", 0, True))
        if a.gather in ("auto", "sendrecv"):
            cands.append(("sendrecv", 1, False))
            if a.overlap or a.gather == "sendrecv":
                overlaps.append(("sendrecv+overlap", 1, True))
        if a.overlap:                                   # --overlap: only the overlap structures
            cands = []
        sym_fits = self.mode == mapn.FORCE_ALL_PAIRS and (n // world) % 1024 == 0 and n % world == 0
        if a.gather in ("auto", "symrccl") and world > 1 and sym_fits:
            # the sharded symmetric step over RCCL alone: pack launch, grouped send/recv of the reaction rows, reduce launch, all-gather
            cands.append(("rccl+symmetric", 6, False))
        if a.gather in ("auto",) + self.P2P_ONLY and world > 1:
            try:
                self.c.p2p_setup_torch()
                self.c.set_timeouts(p2p_ms=a.p2p_timeout_ms)
                ok = 1
            except Exception as e:
                print(f"[bench rank {self.rank}] p2p setup failed: {e}", file=sys.stderr, flush=True)
                ok = 0
            self.p2p_ok = bool(self.all_reduce(ok, "MIN"))              # all ranks or none
            # LAST: these have to prove themselves on this node
            if self.p2p_ok and a.gather in ("auto", "p2p", "sym", "sympush", "p2pall"):
                cands.append(("p2p", 2, False))               # (also the yardstick the symmetric forms are verified against)
            if self.p2p_ok and a.gather in ("flow", "p2pall"):
                cands.append(("p2p+inkernel", 3, False))  # the same exchange overlapped inside the force launch (by name only)
            if self.p2p_ok and a.gather in ("auto", "sym", "p2pall") and sym_fits:
                # the SYMMETRIC step sharded over the ranks: every unordered pair of the job once, reactions
                # stored into the owners' receive regions, positions pulled by the same launch
                cands.append(("p2p+symmetric", 4, False))
            if self.p2p_ok and a.gather in ("auto", "sympush", "p2pall") and sym_fits:
                # ... the same with the new positions PUSHED into the peers' replicas (posted writes instead of read round trips)
                cands.append(("p2p+symmetric+push", 5, False))
        self.candidates = cands + overlaps                 # (the overlap structures: behind everything else)
        return self.candidates

    def exchange_trial(self):
        """Time every candidate (same bytes) on untimed steps; every rank must take the same decision -> MAX over ranks.  A peer-to-peer
        form must also PROVE itself here: no timed-out wait (the library reports one as MAPN_ERR_COMM), bit-identical replicas on all
        ranks, and -- the symmetric forms -- four steps within 1e-5 of the one-sided sharded step."""
        a, c, mapn, dist = self.a, self.c, self.mapn, self.dist
        candidates = self.build_candidates()
        if self.world > 1 and len(candidates) > 1:
            p2p_dead = False
            t_trial0 = time.perf_counter()
            for name, algo, overlap in candidates:
                is_p2p = 2 <= algo <= 5
                if is_p2p and p2p_dead:
                    continue
                if self.all_reduce(1 if (self.trial and time.perf_counter() - t_trial0 > a.trial_seconds) else 0):     # (all ranks decide alike)
                    if self.rank == 0:
                        print(f"[bench] exchange trial: budget of {a.trial_seconds:.0f} s spent -> '{name}' not tried", file=sys.stderr, flush=True)
                    continue
                # every stage is `together`: its collective is the barrier (the device-side waits are bounded -- --p2p-timeout-ms -- so the ranks
                # start together) AND carries a failure of any rank to all of them
                def choose():
                    self.c.set_gather_algorithm(algo)
                    self.c.set_shard_overlap(overlap)
                    if is_p2p:                               # a form's FIRST launches load code objects and touch lazily mapped peer memory
                        self.c.set_timeouts(p2p_ms=max(10000, a.p2p_timeout_ms))     # for the first time: a generous bound for those ...

                def first_steps():
                    self.run_idle(5)
                    if is_p2p:
                        self.c.set_timeouts(p2p_ms=a.p2p_timeout_ms)                 # ... the bound asked for from then on
                inject = a.test_inject_trial_failure and algo == 5 and self.rank == 1
                if inject:
                    os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_TEST_CORRUPT_PUSH"] = "once"
                failed = self.together(choose) or self.together(first_steps)
                dt_trial = float("inf")
                if not failed:
                    t0 = time.perf_counter()
                    failed = self.together(lambda: self.run_idle(30))
                    dt_trial = time.perf_counter() - t0
                if inject:
                    os.environ.pop("MAPN_TEST_HOOKS", None); os.environ.pop("MAPN_TEST_CORRUPT_PUSH", None)
                bad = bool(failed)
                if not bad and algo >= 2:
                    same = self.replicas_consistent()           # (a collective: before the status word is looked at, on every rank)
                    bad = self.all_reduce(0 if (self.c.p2p_status() == 0 and same) else 1)
                    if bad:
                        failed = "replicas differ across ranks"
                if not bad and algo in (4, 5, 6):
                    # identical replicas do not show that the reactions ARRIVED: compare four steps from the
                    # initial state with the one-sided sharded step (verified above)
                    dev = self.symmetric_deviation(algo)
                    bad = self.all_reduce(0 if dev < 1e-5 else 1)
                    if bad:
                        failed = f"symmetric sharded step deviates from the one-sided one by {dev:.2e} of the spread after 4 steps"
                if bad:
                    if self.rank == 0:
                        print(f"[bench] exchange '{name}' failed on this node ({failed or 'on another rank'}) -> not used; state re-initialised",
                              file=sys.stderr, flush=True)
                    # a context whose device-side wait gave up or whose row / position check failed STAYS failed, whatever the
                    # algorithm (6 sets the status word too): replace it on every rank, or the next candidate inherits the fault
                    stuck = self.all_reduce(1 if self.c.p2p_status() != 0 else 0)
                    if is_p2p:
                        self.p2p_failure = f"{name}: {failed or 'failed on another rank'}"
                        p2p_dead = algo == 2                    # the plain exchange failed on a healthy context: every peer-to-peer form shares its transport
                    if is_p2p or stuck:
                        self.rebuild(with_p2p=self.p2p_ok and not p2p_dead and ("p2p" in self.trial or not is_p2p))
                    else:
                        self.reinit()
                    continue
                self.trial[name] = float(self.all_reduce(dt_trial, dtype=self.torch.float64)) / 30
            self.gather_algo = min(self.trial, key=self.trial.get) if self.trial else "allgather"
            for asked, nm in (("sym", "p2p+symmetric"), ("sympush", "p2p+symmetric+push"), ("symrccl", "rccl+symmetric")):
                if a.gather == asked and nm in self.trial:
                    self.gather_algo = nm                   # asked for by name: it only had to pass its check
            if self.rank == 0:
                print("[bench] exchange trial: " + ", ".join(f"{k} {v*1e6:.1f} us/step" for k, v in self.trial.items()) + f" -> {self.gather_algo}",
                      file=sys.stderr, flush=True)
        else:
            self.gather_algo = candidates[0][0] if candidates else "allgather"
            if self.gather_algo.startswith("p2p") and not self.p2p_ok:
                sys.exit("bench: --gather p2p requested but the peer-to-peer setup failed")
        if a.gather in ("sym", "sympush", "symrccl") and "symmetric" not in self.gather_algo:
            sys.exit(f"bench: --gather {a.gather}: the sharded symmetric step does not apply (N / ranks must be a multiple of 1024) or failed its check")
        self.select({x[0]: x for x in candidates}.get(self.gather_algo, (self.gather_algo, 0, False)))

    def fall_back(self, fail):
        """A peer-to-peer form that had passed its trial failed in the run itself (a pushed position that did not match its checksum, a
        reaction row that never arrived whole, a wait that gave up).  ONCE, the fastest OTHER form the trial verified takes over: new
        contexts on all ranks (a failed one stays failed), the seeded initial state, warm-up and the K timed steps again."""
        others = sorted((k for k in self.trial if k != self.gather_algo), key=self.trial.get)
        if self.fallback_after_failure is not None or not others:
            sys.exit(f"bench: exchange '{self.gather_algo}' failed during the run ({fail}) and no verified form is left to fall back to")
        self.fallback_after_failure = f"'{self.gather_algo}' failed during the run: {fail[:300]}"
        if self.rank == 0:
            print(f"[bench] {self.fallback_after_failure} -> falling back to '{others[0]}'", file=sys.stderr, flush=True)
        self.gather_algo = others[0]
        chosen = {x[0]: x for x in self.candidates}[self.gather_algo]
        self.rebuild(with_p2p=2 <= chosen[1] <= 5)
        self.select(chosen)

    def sharded_xcd_ab(self, xcd):
        """SHARDED symmetric step: the library has planned every rank's launch with the die weights of ITS GPU (MAPN_FLAG_XCD_CALIBRATE:
        a temporary unsharded context's calibration when the step was prepared -- no collective in it, and every die holds heavy and
        light blocks there, so the measurement is of the dies, not of the blocks' classes); an untimed A/B against the unweighted plan,
        MAX over ranks, decides for all of them.  Loopback at 65 536 / 8: -1.2 ... -1.5 % per step before the heavy blocks were moved to
        the odd dispatch slots by default, less since."""
        a, c, mapn, n, world = self.a, self.c, self.mapn, self.n, self.world
        try:
            def burst_all(k):
                err = self.together(self.idle)
                t0 = time.perf_counter()
                err = err or self.together(lambda: self.run_idle(k))
                if err:
                    raise RankFailure(err)                  # (on EVERY rank: `together`)
                return float(self.all_reduce(time.perf_counter() - t0, dtype=self.torch.float64)) / k

            def all_set(w):
                err = self.together(lambda: c.set_sym_xcd_weights(w))
                if err:
                    raise RankFailure(err)
            kk = max(20, min(400, int(0.05 / (0.1e-3 * (n / 65536.0) ** 2 * 8 / world))))
            pl = c.sym_plan()
            w = list(pl.xcd_weight)
            if self.all_reduce(1 if pl.xcd_mode != 0 else 0, "MIN"):            # all ranks or none
                t_w = min(burst_all(kk), burst_all(kk))
                all_set(None)
                t_def = min(burst_all(kk), burst_all(kk))
                xcd.update({"weights": w, "source": "library (MAPN_FLAG_XCD_CALIBRATE: a temporary unsharded context on every rank's GPU; rank 0's weights shown)",
                            "form": {1: "spread", 2: "class-aware"}.get(pl.xcd_mode), "trial_ms": {"default": t_def * 1e3, "weighted": t_w * 1e3}})
                if a.xcd == "on" or t_w < t_def * 0.998:
                    all_set(w)
                    xcd["used"] = c.sym_plan().xcd_mode != 0
            else:
                xcd["note"] = "the library's calibration did not apply on every rank (a rank's share must be a multiple of 8 blocks)"
                c.set_sym_xcd_weights(None)
        except (mapn.MapnError, RankFailure) as e:            # (a failure here leaves the default plan: the run goes on)
            xcd["error"] = str(e)[:200]
            try:
                c.set_sym_xcd_weights(None)
            except mapn.MapnError:
                pass
