#!/usr/bin/env python3
"""bench.py -- body-pair interactions/s of the all-pairs n-body step on 1..8 MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched as
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (one rank per
GPU) -- or WITHOUT a launcher: ``python bench.py --gpus N`` then starts its N ranks itself (launch_ranks).  W untimed warm-up steps, then EXACTLY K steps timed between barrier + device sync on both
sides, MAX over ranks, rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): 65 536 bodies, fp32, seeded two-shell state (seed 1),
dt 0.1, damping 1, soft^2 25, mass 70000/N (the parity regime; timing does not depend on mass).
A "step" is one Compute::Simulate over all bodies.  Inputs are resident in HBM when the timed
region starts.  metric = N^2 ordered pairs x steps / wall seconds, whole job.

roofline: the force kernel is bound by the fp32 VECTOR ALU, not HBM and not MFMA (arithmetic
intensity ~23 000 flop/B); `peak` is CUs x clock x 256 flop/clk = 157.3 TFLOP/s, which is also the
dense f32 MFMA peak of MI355X_MICROARCH.md.  `achieved` = 20 flop per ordered pair
x the pairs one launch processes / the force kernel's mean launch duration, measured live with
HIP events recorded on the compute stream around sampled force launches of the timed region (one
event pair per sampled step: the step IS one launch).  `held_clock_ghz` is the shader clock the chip
held under this kernel, stamped in-kernel right after the timed region (mapn_measure_clock), and
`frac_at_held_clock` prices the same achieved rate against CUs x held clock x 256.

This file: the arguments, the TIMED REGION (run_steps) and the JSON line.  bench_legs.py: cpu_baseline, the roofline object, the power /
central-well / partial-active legs and SURVEY 8(d)'s statistic; bench_ranks.py: the launcher, the exchange trial and the fallback (N > 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_legs import (FLOP_PER_PAIR, HBM_BYTES_PER_BODY, central_well_leg, cpu_baseline, kernel_source_sha16, partial_active_leg, pmc_traffic,   # noqa: E402,F401
                        power_leg, power_sensor, read_sensor, roofline_all_pairs, roofline_central_well, step_spread, survey_8d)
from bench_ranks import Job, launch_ranks   # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--bodies", type=int, default=65536)
    ap.add_argument("--mode", choices=["all_pairs", "central_well"], default="all_pairs")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--plan", default="", help="kernel,bodies_per_lane,waves,sb,fused e.g. sgpr,2,8,8,1 (fused: 0 = rows + reduce launch, 1 = one launch, 2 = ticket form always)")
    ap.add_argument("--kernel", choices=["auto", "lds", "sgpr", "sym"], default="auto",
                    help="force kernel: scalar-cache (auto), LDS-tiled, or the symmetric (Newton's third law) kernel")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl")
    ap.add_argument("--gather", choices=["auto", "allgather", "sendrecv", "p2p", "flow", "sym", "sympush", "symrccl", "p2pall"], default="auto",
                    help="how the in-library exchange is issued: RCCL ncclAllGather, one group of RCCL send/recv pairs, the "
                         "direct peer-to-peer pull kernel (hipIpc + device flags), that exchange inside the force launch (flow), or the "
                         "SYMMETRIC step sharded over the ranks (sym: positions pulled, sympush: pushed, symrccl: over RCCL alone); auto "
                         "times every way that sets up and verifies on this node during untimed steps and keeps the fastest; p2pall = "
                         "the same trial over the peer-to-peer forms only (no RCCL communicator: ranks sharing one device in tests)")
    ap.add_argument("--timer-interval", type=int, default=-1, help="time every T-th force launch of the timed region with HIP events (0 = off; default 8: each hipEventRecord costs ~4 us of queue time, 1.6 %% of a 0.9 ms step when every step carries three)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for barriers / rendezvous (gloo for single-GPU tests)")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses device 0 (needs --dist-backend gloo --gather p2p)")
    ap.add_argument("--no-power-leg", action="store_true", help="skip the untimed 2.5 s of steps behind the run during which the package power sensor is read (single GPU)")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed clock-ramp phase before the W warm-up steps (the chip needs a few hundred ms of "
                         "load to settle its clock; 0 disables)")
    ap.add_argument("--p2p-timeout-ms", type=int, default=3000,
                    help="bound of the device-side waits of the peer-to-peer forms (raise it when ranks time-slice one device in tests)")
    ap.add_argument("--xcd", choices=["auto", "off", "on"], default="auto",
                    help="XCD-aware parts of the symmetric kernel (1 GPU): calibrate the dies' speeds during the untimed prewarm and size "
                         "every part by the die it runs on; auto keeps them only if an untimed A/B against the default plan wins")
    ap.add_argument("--xcd-weights", default="", help="w0,...,w7: run the symmetric kernel under THESE relative die speeds (1024 = the fastest; mapn_set_sym_xcd_weights) -- no "
                                                      "calibration, no A/B: replays the plan of an earlier run's line (config.xcd_aware_parts.weights) bit for bit")
    ap.add_argument("--no-partial-leg", action="store_true", help="skip the untimed ~0.3 s behind the run that measures partially active steps (num_active = N/2, 3N/4) against the one-sided step (single GPU)")
    ap.add_argument("--no-survey-leg", action="store_true", help="skip the five untimed 100-step regions behind a run with K < 100 (config.survey_8d; ranks time-slicing ONE device in tests make them slow)")
    ap.add_argument("--no-central-well-leg", action="store_true", help="skip the untimed ~0.3 s behind the run that measures the HBM-bound CENTRAL_WELL step at 4 Mi and 16 Mi bodies (single GPU)")
    ap.add_argument("--overlap", action="store_true", help="sharded mode: own-segment launch overlapped with the all-gather")
    ap.add_argument("--trial-seconds", type=float, default=90.0,
                    help="wall-time budget of the exchange trial (N > 1, --gather auto): once it is spent the candidates not yet tried are skipped")
    ap.add_argument("--test-inject-trial-failure", action="store_true",
                    help="testing only: rank 1 corrupts ONE pushed position while the exchange TRIAL steps 'p2p+symmetric+push': every rank must drop that form and go on")
    ap.add_argument("--test-inject-push-failure", action="store_true",
                    help="testing only: rank 1 corrupts ONE pushed position in the timed region (MAPN_TEST_HOOKS): the run must notice and fall back")
    ap.add_argument("--force-comm", action="store_true",
                    help="create the torch.distributed group and the in-library RCCL communicator even for one rank (exercises the sharded code path on a 1-GPU box)")
    return ap.parse_args()


def run_steps(job, k, recoverable, closing):
    """THE TIMED REGION's body: k steps, then the barrier + device sync.  run_steps.closed_at: the moment the region closed (barrier passed,
    device idle); run_steps.idle_s: how long THIS rank's k steps took without the closing collective (RCCL backend: HIP events on the compute
    stream around them; gloo: wall clock until its device was idle).  With a peer-to-peer form on N > 1 ranks a device-side check or wait that
    fails on ANY rank (they are all bounded: every rank gets out of its own) comes back as text on EVERY rank -- the verdict collective
    carries it -- instead of leaving the others in a barrier for ever.
    Closing a region for N > 1 (`closing`, made BEFORE any timed region, never inside one: VERDICT r4 #7):
      * RCCL backend: the barrier is an all-reduce of one pre-made device word enqueued, stream-ordered, BEHIND this rank's k steps on the
        library's compute stream (torch.cuda.ExternalStream) while the device is still working on them -- its host-side launch cost and
        the peers' arrival overlap the steps, and what is left after the last rank's last step is the collective's kernel and one host
        sync.  The verdict (did a device-side check fail on any rank?) travels in a second collective AFTER the region is closed.
      * gloo (tests: ranks sharing one device, host tensors): one host-side all-reduce is barrier and verdict."""
    mapn, torch, dist, c = job.mapn, job.torch, job.dist, job.c
    verdict, barrier_word, ev_first, ev_last = closing["verdict"], closing["barrier_word"], closing["ev_first"], closing["ev_last"]
    fail = None
    t_begin = time.perf_counter()
    ordered = None
    if barrier_word is not None and job.gather_fn is None:
        try:
            ordered = closing["stream_of"](c)
        except Exception as e:                             # (no stream wrapper: the host-side form below)
            print(f"[bench rank {job.rank}] stream-ordered closing barrier unavailable ({e}); closing on the host", file=sys.stderr, flush=True)
    issued = False
    try:
        if ordered is not None:
            ev_first.record(ordered)
        for _ in range(k):
            job.step()
        if ordered is not None:
            ev_last.record(ordered)
            with torch.cuda.stream(ordered):
                dist.all_reduce(barrier_word, op=dist.ReduceOp.MAX)      # the barrier: behind this rank's k steps, on the device
            issued = True
        else:
            c.WaitForGpu()
    except mapn.MapnError as e:
        if not recoverable:
            raise
        fail = str(e)
    if ordered is not None and not issued:                 # (a step failed to enqueue: the peers' barrier still needs this rank)
        dist.all_reduce(barrier_word, op=dist.ReduceOp.MAX)
    if torch is not None:
        torch.cuda.synchronize()
    run_steps.idle_s = time.perf_counter() - t_begin       # THIS rank's device is idle: its k steps are done
    if ordered is not None:
        # stream-ordered form: the device sync above returned behind the barrier collective -- every rank's k steps are done, the region
        # is closed.  The library's own drain (its bookkeeping of timer events, the check of the status words the device-side waits and
        # data checks leave) follows OUTSIDE the region; what it finds goes into the verdict.
        run_steps.closed_at = time.perf_counter()
        try:
            c.WaitForGpu()
        except mapn.MapnError as e:
            if not recoverable:
                raise
            fail = fail or str(e)
    if dist is not None and ordered is None:
        verdict.fill_(1 if fail else 0)
        dist.all_reduce(verdict, op=dist.ReduceOp.MAX)      # (host-side form: this collective IS the barrier, and carries the verdict)
        bad = bool(verdict.item())
        torch.cuda.synchronize()
        run_steps.closed_at = time.perf_counter()
    else:
        if ordered is None:
            run_steps.closed_at = time.perf_counter()      # (one GPU, no process group)
        bad = bool(fail)
        if dist is not None:
            try:
                run_steps.idle_s = ev_first.elapsed_time(ev_last) * 1e-3 if issued else run_steps.idle_s
            except RuntimeError:
                pass
            verdict.fill_(1 if fail else 0)                # the verdict: outside the timed region
            dist.all_reduce(verdict, op=dist.ReduceOp.MAX)
            bad = bool(verdict.item())
    if bad and not fail:
        fail = "a device-side check or wait failed on another rank"
    if fail and not recoverable:
        raise mapn.MapnError(-4, fail)
    return fail


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
            # no launcher around this process: start the ranks here (before anything touches the GPU) and relay rank 0's line
            sys.exit(launch_ranks(a.gpus, os.path.abspath(__file__)))
        a.gpus = world
    dist = None
    torch = None
    under_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ
    saved_stdout_fd = None
    if world > 1 or (a.force_comm and under_launcher):
        # stdout carries ONE JSON line (rank 0's) and nothing else: what the collective libraries print on it while they set up (RCCL's
        # version banner, for one) goes to stderr -- file descriptor 1 points there until the line is printed
        sys.stdout.flush()
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        import torch            # first: its HIP runtime is then the one libmapn binds to
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)

    import mapn
    from mapn.compute import device_info

    n = a.bodies
    mode = mapn.FORCE_ALL_PAIRS if a.mode == "all_pairs" else mapn.FORCE_CENTRAL_WELL
    flags = (mapn.FLAG_USE_GRAPH if a.graph else 0) | (mapn.FLAG_SHARD_OVERLAP if a.overlap else 0)
    kern = {"auto": mapn.KERNEL_AUTO, "lds": mapn.KERNEL_LDS, "sgpr": mapn.KERNEL_SCALAR, "sym": mapn.KERNEL_SYMMETRIC}[a.kernel]
    # XCD-aware parts (1 GPU, symmetric kernel): the LIBRARY calibrates the dies when the context is created (MAPN_FLAG_XCD_CALIBRATE:
    # the plan any C-ABI caller gets with that one config bit); below, an untimed A/B decides whether the weights stay
    given_w = [int(x) for x in a.xcd_weights.split(",")] if a.xcd_weights else None
    if given_w is not None and (len(given_w) != 8 or world > 1):
        sys.exit("bench: --xcd-weights takes eight comma-separated weights and applies to the single-GPU run")
    if given_w is not None:
        a.xcd = "off"                                          # (no calibration and no A/B: the plan is the one asked for)
    xcd_by_library = dist is None and a.mode == "all_pairs" and a.xcd != "off" and kern in (mapn.KERNEL_AUTO, mapn.KERNEL_SYMMETRIC) and not a.plan and not a.graph
    # (N > 1: the same flag makes every rank's library measure ITS GPU with a temporary unsharded context when the sharded symmetric step
    #  is prepared -- no collective in it -- and plan the rank's launch with those weights; the A/B below decides here too)
    if a.mode == "all_pairs" and a.xcd != "off" and kern in (mapn.KERNEL_AUTO, mapn.KERNEL_SYMMETRIC) and not a.plan and not a.graph:
        flags |= mapn.FLAG_XCD_CALIBRATE
    job = Job(a, mapn, torch, dist, rank, world, local_rank, mode, flags, kern)
    job.create()
    created_note = mapn.load_library().mapn_last_error().decode(errors="replace")    # (what mapn_create left behind: why a calibration did not stay)
    info = device_info(local_rank)
    if dist is not None:
        job.setup_transport()
    job.apply_plan()
    # each hipEventRecord costs ~4 us of queue time: at most one event PAIR per 4 steps, also for short runs
    timer_interval = a.timer_interval if a.timer_interval >= 0 else (8 if a.steps >= 32 else 4)
    if a.timer_interval < 0 and world > 1:
        # a sharded step is 0.09 ms at 65 536 / 8, an event pair 8 us of it: two sampled steps in a short region, every 16th in a long one
        timer_interval = 16 if a.steps >= 32 else max(4, a.steps // 2)
    job.timer_interval = timer_interval
    job.c.set_timers(timer_interval)
    if dist is not None and job.transport == "rccl":
        job.exchange_trial()                               # N > 1: which step structure runs (untimed, verified; bench_ranks.py)
    prewarm_steps = 0
    if a.prewarm_ms > 0:
        # same work as a timed step, just not timed: lets the clock settle so that a short K does
        # not measure the ramp (1.18 ms/step over the first 5 steps vs 0.88 ms steady state)
        if dist is None:
            t_end = time.perf_counter() + a.prewarm_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(16):
                    job.step()
                job.c.WaitForGpu()
                prewarm_steps += 16
        else:
            # sharded: every step holds a collective, so all ranks must run the SAME number of
            # steps -- a fixed count (about prewarm_ms at the single-GPU step time / world)
            prewarm_steps = max(16, int(a.prewarm_ms * 1e-3 / (0.9e-3 / world * (n / 65536.0) ** 2)) // 16 * 16)
            prewarm_steps = min(prewarm_steps, 4096)
            fail = job.together(lambda: job.run_idle(prewarm_steps))
            if fail:                                           # (a verified form failing here: the next verified one takes over, as in the timed run)
                job.fall_back(fail)
    # XCD-aware parts (symmetric kernel): the eight dies do not run at one speed and a launch gives each the same work.
    # Untimed: the library has calibrated; an A/B of the weighted plan against the default one decides -- the weights stay only if they win.
    xcd = {"mode": a.xcd, "weights": None, "used": False, "source": None}
    if given_w is not None:
        job.c.set_sym_xcd_weights(given_w)
        plw = job.c.sym_plan()
        xcd.update({"mode": "given", "weights": list(plw.xcd_weight) if plw.xcd_mode else given_w, "used": plw.xcd_mode != 0, "source": "--xcd-weights (no calibration, no A/B)",
                    "form": {0: None, 1: "spread", 2: "class-aware"}.get(plw.xcd_mode)})
    if dist is not None and world > 1 and "symmetric" in job.gather_algo and a.xcd != "off" and not a.plan:
        job.sharded_xcd_ab(xcd)
    if xcd_by_library:
        c = job.c
        try:
            def burst(k):
                c.WaitForGpu(); t = time.perf_counter()
                for _ in range(k):
                    job.step()
                c.WaitForGpu()
                return (time.perf_counter() - t) / k
            kk = max(8, min(120, int(0.08 / (0.65e-3 * (n / 65536.0) ** 2))))
            pl = c.sym_plan()
            if pl.xcd_mode != 0:                           # the library's creation-time calibration applied its weights
                w = list(pl.xcd_weight)
                xcd["weights"], xcd["source"] = w, "library (MAPN_FLAG_XCD_CALIBRATE at mapn_create)"
                xcd["form"] = "class-aware" if pl.xcd_mode == 2 else "spread"
                t_w = min(burst(kk), burst(kk))
                c.set_sym_xcd_weights(None)
                t_def = min(burst(kk), burst(kk))
                xcd["trial_ms"] = {"default": t_def * 1e3, "weighted": t_w * 1e3}
                if a.xcd == "on" or t_w < t_def * 0.998:
                    c.set_sym_xcd_weights(w)
                    xcd["used"] = c.sym_plan().xcd_mode != 0
            else:
                xcd["note"] = ("the library runs the default plan: its calibration did not apply (block count not a multiple of 8, or the one-sided kernel "
                               "runs) or the calibrated plan did not win the A/B mapn_create runs behind it: " + created_note[:200])
        except mapn.MapnError as e:
            xcd["error"] = str(e)[:200]
            try:
                c.set_sym_xcd_weights(None)
            except mapn.MapnError:
                pass
    if a.prewarm_ms > 0 and "trial_ms" in xcd:
        # the A/B ends with a re-plan (scratch freed and allocated: the device idles for milliseconds) and the chip needs ~15 steps to be back at
        # the clock it holds under load -- with W = 5 the driver's K = 20 region sat on that ramp (first region 0.616 ms against 0.593 for the
        # four behind it, same run).  A short untimed ramp, as in front of the A/B: the W warm-up steps and the K timed ones follow it.
        per_step_s = min(xcd["trial_ms"].values()) * 1e-3     # (150 ms of steps, whatever the size: a 4 Mi-body step is 2.5 s)
        ramp = max(4, min(256, int(0.15 / per_step_s))) if dist is None else max(16, prewarm_steps // 4)
        fail = job.together(lambda: job.run_idle(ramp))
        if fail:
            job.fall_back(fail); xcd["used"] = False           # (N = 1: nothing to fall back to -- it exits with the library's text)
        prewarm_steps += ramp
    closing = job.closing_collective()                     # its tensors and events: made here, before any timed region (run_steps)

    def reduce_max(*vals):
        """MAX over ranks of a few wall-clock figures (N = 1: themselves)."""
        if dist is None:
            return vals
        t = torch.tensor(list(vals), dtype=torch.float64, device=job.red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(x) for x in t.tolist())

    # ---- W untimed warm-up steps, then EXACTLY K steps between barrier + device sync on both sides ----
    while True:
        recoverable = dist is not None and world > 1 and job.transport.startswith("p2p")
        fail = run_steps(job, a.warmup, recoverable, closing)
        if not fail:
            if a.test_inject_push_failure and rank == 1 and job.fallback_after_failure is None:
                os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_TEST_CORRUPT_PUSH"] = "once"
            job.c.set_timers(timer_interval)      # restart the sampling phase: the first timed step carries events
            job.c.kernel_stats(reset=True)
            t0 = time.perf_counter()
            fail = run_steps(job, a.steps, recoverable, closing)
            elapsed = run_steps.closed_at - t0                 # barrier + device sync on both sides of exactly K steps
            elapsed_idle = min(run_steps.idle_s, elapsed)      # (N > 1: without the closing collective -- reported beside the contract's figure, never instead of it)
        if not fail:
            break
        job.fall_back(fail)                                    # ONCE: the fastest other verified form, new contexts, warm-up and K steps again
        xcd["used"] = False
    elapsed, elapsed_idle = reduce_max(elapsed, elapsed_idle)
    c = job.c
    gather_algo, trial = job.gather_algo, job.trial

    st = c.kernel_stats()
    quarters, spread = step_spread(mapn, c, a.steps)       # how the step time is spread over the timed region (the steps that carried HIP events), by quarter
    first, count = c.shard_range()
    # the clock the chip held under this kernel: stamped diagnostic steps right behind the timed region
    # (same state of the chip; untimed).  Single GPU, scalar-cache and symmetric kernels.
    clock = None
    if world == 1 and a.mode == "all_pairs" and dist is None:
        try:
            clock = c.measure_clock(8)
        except mapn.MapnError as e:
            print(f"[bench] clock measurement unavailable: {e}", file=sys.stderr, flush=True)
    consistent = None
    sym_dev_after = None
    if dist is not None and world > 1 and gather_algo != "n/a":
        consistent = job.replicas_consistent() and c.p2p_status() == 0      # (the status word serves every device-side wait)
        if consistent and "symmetric" in gather_algo:
            # once more, after the timed run: the sharded symmetric step against the one-sided one
            dev_after = job.symmetric_deviation({"p2p+symmetric": 4, "p2p+symmetric+push": 5, "rccl+symmetric": 6}[gather_algo])
            sym_dev_after = reduce_max(dev_after)[0]
            consistent = sym_dev_after < 1e-5 and c.p2p_status() == 0
            sym_dev_after = sym_dev_after if sym_dev_after != float("inf") else None      # (a rank's library failed in the check: not a JSON number)
        if not consistent and rank == 0:
            print("[bench] WARNING: position replicas differ across ranks after the run (or the sharded symmetric step "
                  "failed its check) -- the exchange misbehaved; this result is INVALID", file=sys.stderr, flush=True)
    # SURVEY 8(d) asks for >= 100 steps and a median of five repeats.  (1) `repeats`: four more regions of the same K steps follow the timed
    # one (untimed by the contract: `value` stays the first region's) -- how far one region is from the typical one on this box.  (2)
    # `survey_8d`: the statistic itself whatever K is -- for K >= 100 those five regions ARE it; for a shorter K (the driver's
    # own command has K = 20) five more regions of 100 steps each.  All of it AFTER the validity checks of the timed run; a failure here
    # (N > 1, peer-to-peer form) only ends the lists: it cannot taint the result.
    repeats = s8d = None
    pairs_per_step = float(n) * float(n) if a.mode == "all_pairs" else float(n)
    if a.steps >= 1 and consistent is not False:
        recoverable = dist is not None and world > 1 and job.transport.startswith("p2p")

        def regions(k, how_many):
            out = []
            for _ in range(how_many):
                if job.together(job.idle):
                    break
                t0 = time.perf_counter()
                if run_steps(job, k, recoverable, closing):
                    break
                out.append(reduce_max(run_steps.closed_at - t0)[0] / k * 1e3)
            return out
        reps = [elapsed / a.steps * 1e3] + regions(a.steps, 4)
        repeats = {"ms_per_step": [round(x, 5) for x in reps], "median_ms_per_step": round(sorted(reps)[len(reps) // 2], 5), "meets_survey_8d": a.steps >= 100,
                   "note": "the timed region (first entry: `ms_per_step`, `value`) and four more regions of the same K steps behind it (after the run's validity checks); "
                           "SURVEY 8(d)'s statistic proper: config.survey_8d"}
        # (the five 100-step regions only where they are cheap -- under 6 s: 65 536 bodies 0.3 s, 262 144 4.8 s; a 1 Mi-body step is 0.15 s, a 4 Mi-body one 2.5 s)
        cheap = elapsed / a.steps * 500.0 < 6.0
        r8 = reps if a.steps >= 100 else (regions(100, 5) if len(reps) == 5 and cheap and not a.no_survey_leg else [])
        if len(r8) == 5:
            s8d = survey_8d(r8, max(a.steps, 100), pairs_per_step)
    # what the package draws under this kernel (single GPU; untimed, behind everything that is timed or checked): the symmetric kernel
    # runs AT or within a few per cent of the board's power cap (`at_cap` is this box's reading) -- the clock it holds (`held_clock_ghz`) is
    # what the cap leaves, which is why `frac` (priced at the nominal 2.4 GHz) stops near 0.92 while `frac_at_held_clock` is 1.0
    power = None
    if world == 1 and dist is None and not a.no_power_leg:
        try:
            hw = power_sensor(local_rank)
            if hw:
                def step_batch(k):
                    for _ in range(k):
                        c.Simulate(n, c.GetFenceValue())
                    job.sync()
                power = power_leg(step_batch, hw, ms_per_step=elapsed / a.steps * 1e3)
        except (mapn.MapnError, OSError) as e:
            print(f"[bench] power leg unavailable: {e}", file=sys.stderr, flush=True)
    sym_plan_desc = None
    if st.kernel_name.decode() == "force_sym_kernel":
        try:
            pl = c.sym_plan()
            sym_plan_desc = {"waves_per_workgroup": pl.waves, "parts_per_block": pl.parts, "taper": [pl.taper1, pl.taper2],
                             "wave_bias_older_to_younger": list(pl.wave_bias), "windows": len(pl.windows), "table_sets": pl.sets,
                             "xcd_mode": {0: "none", 1: "spread", 2: "class-aware (heavy blocks on the faster dies)"}.get(pl.xcd_mode), "class_dies": pl.class_die if pl.xcd_mode == 2 else None}
        except mapn.MapnError:
            pass
    # The plan of this run, replayable: from the seeded initial state exactly 10 steps under the plan that was timed, then the
    # checksums of both position buffers.  Two runs with the same --xcd-weights (or both with the default plan) print the same pair; runs
    # whose calibrations differ do not -- the weights are part of the summation order.  Untimed, single GPU.
    replay = None
    if world == 1 and dist is None and a.mode == "all_pairs":
        try:
            pos0, vel0 = mapn.generate_initial_state(n, seed=a.seed)
            c.upload_state(pos0, vel0)
            for _ in range(10):
                c.Simulate(n, c.GetFenceValue())
            replay = {"steps": 10, "checksums": [int(x) for x in c.replica_checksum()],
                      "note": "sum of the 32-bit words of each position buffer after 10 steps from the seeded state under the timed plan: equal for equal plans (pass config.xcd_aware_parts.weights as --xcd-weights to replay a calibrated run)"}
        except mapn.MapnError as e:
            print(f"[bench] replay checksum unavailable: {e}", file=sys.stderr, flush=True)
    # PARTIALLY ACTIVE steps in the driver's line: num_active = N/2 and 3N/4 -- the form the library picks by itself (the split
    # form at this size) against the one-sided step round 4 ran there, selected through the MAPN_PARTIAL_FORM hook for the comparison only.
    # Untimed, ~0.3 s, single GPU, after everything that is timed or checked.
    partial = None
    if world == 1 and dist is None and a.mode == "all_pairs" and not a.no_partial_leg and n >= 16384 and st.kernel_name.decode() == "force_sym_kernel":
        try:
            partial = partial_active_leg(mapn, c, n)
        except mapn.MapnError as e:
            print(f"[bench] partial-active leg unavailable: {e}", file=sys.stderr, flush=True)
        finally:
            os.environ.pop("MAPN_PARTIAL_FORM", None)
            if partial_active_leg.set_hooks:
                os.environ.pop("MAPN_TEST_HOOKS", None)
    # The HBM-bound mode, in the driver's line ("CENTRAL_WELL mode is HBM-bound at 56 B/body -- report GB/s for it"):
    # CSMain as shipped at the reference's maximum (defines.h:45: 4 194 304 bodies -- 235 MB per step, inside the 256 MiB Infinity Cache: a CACHE
    # rate) and at 16 777 216 bodies (940 MB per step: past every cache, the kernel's non-temporal form: the HBM figure).  Untimed, ~0.1 s of steps each.
    central_well = None
    if world == 1 and dist is None and a.mode == "all_pairs" and not a.no_central_well_leg and rank == 0:
        central_well = []
        c.close()                                              # (the leg's contexts take up to 0.9 GB: give the main context's scratch back first)
        for nb_cw in (4 * 1024 * 1024, 16 * 1024 * 1024):
            try:
                central_well.append(central_well_leg(mapn, local_rank, nb_cw))
            except (mapn.MapnError, MemoryError) as e:
                central_well.append({"bodies": nb_cw, "error": str(e)[:200]})
    if rank == 0:
        value = pairs_per_step * a.steps / elapsed
        out = {
            "metric": "body-pair interactions/s" if a.mode == "all_pairs" else "bodies/s",
            "value": value,
            "unit": "interactions/s" if a.mode == "all_pairs" else "bodies/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "ms_per_step_before_closing_barrier": (elapsed_idle / a.steps * 1e3) if dist is not None else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{n} bodies, all-pairs softened gravity + kick-drift step, fp32 (BASELINE configs[1])"
                       if a.mode == "all_pairs" else f"{n} bodies, central-well step as shipped (nBodyGravityCS.hlsl:86-109)",
                       "bodies": n, "mode": a.mode, "parallelism": f"bodies sharded x{world}" if world > 1 else "1 GPU",
                       "transport": job.transport, "exchange": gather_algo, "exchange_trial_us_per_step": {k: v * 1e6 for k, v in trial.items()},
                       "replicas_bit_identical_after_run": consistent, "valid": consistent is not False, "prewarm_steps_untimed": prewarm_steps, "seed": a.seed, "mass": "70000/N", "device": (info.name.decode() or "MI355X") + " / " + info.arch.decode(),
                       "kernel": st.kernel_name.decode(), "bodies_per_lane": st.bodies_per_lane, "j_splits": st.j_splits,
                       "grid": [st.grid_x, st.grid_y], "block": st.block_x, "fused_integrator": bool(st.fused),
                       "epilogue": {0: "partial rows + reduce_integrate launch", 1: "fused in the workgroup", 2: "last-arriver ticket (one launch per step)", 3: "symmetric kernel: force rows + sym_reduce_integrate launch"}.get(st.epilogue, "?"),
                       "launches_per_step": int(st.force_launches_per_step) * (1 if st.fused else 2), "timer_interval": timer_interval,
                       "step_ms_by_quarter_of_the_timed_region": quarters, "step_ms_spread": spread,
                       "repeats": repeats, "survey_8d": s8d,
                       "step_ms_note": "device time (HIP events on the compute stream) of the steps of the timed region that carried events; ms_per_step is the wall clock over all of them",
                       "p2p_failure": job.p2p_failure, "fallback_after_failure": job.fallback_after_failure, "sharded_symmetric_deviation_after_run": sym_dev_after,
                       "xcd_aware_parts": xcd, "symmetric_plan": sym_plan_desc, "replay": replay, "central_well": central_well, "partial_active": partial,
                       "closing_collective_us": ((elapsed - elapsed_idle) * 1e6) if dist is not None else None},
        }
        if a.mode == "all_pairs":
            out["roofline"] = roofline_all_pairs(st, info, clock, power, n, count, world, pairs_per_step)
        else:
            out["roofline"] = roofline_central_well(n, a.steps, elapsed)
        if world == 1 and not a.no_cpu_baseline and a.mode == "all_pairs":
            out["cpu_baseline"] = cpu_baseline(n, a.seed, a.cpu_seconds)
        if saved_stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout_fd, 1)
        print(json.dumps(out), flush=True)
    c.close()                                                  # (idempotent: the central-well leg has already closed it on a single GPU)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
