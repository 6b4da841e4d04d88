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
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

FLOP_PER_PAIR = 20          # op count of nBodyGravityCS.hlsl:44-57, rsqrt = 1 flop
HBM_BYTES_PER_BODY = 56     # 16+12 read, 16+12 written


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
    ap.add_argument("--p2p-timeout-ms", type=int, default=1000,
                    help="bound of the device-side waits of the peer-to-peer forms (raise it when ranks time-slice one device in tests)")
    ap.add_argument("--xcd", choices=["auto", "off", "on"], default="auto",
                    help="XCD-aware parts of the symmetric kernel (1 GPU): calibrate the dies' speeds during the untimed prewarm and size "
                         "every part by the die it runs on; auto keeps them only if an untimed A/B against the default plan wins")
    ap.add_argument("--xcd-weights", default="", help="w0,...,w7: run the symmetric kernel under THESE relative die speeds (1024 = the fastest; mapn_set_sym_xcd_weights) -- no "
                                                      "calibration, no A/B: replays the plan of an earlier run's line (config.xcd_aware_parts.weights) bit for bit")
    ap.add_argument("--no-partial-leg", action="store_true", help="skip the untimed ~0.3 s behind the run that measures partially active steps (num_active = N/2, 3N/4) against the one-sided step (single GPU)")
    ap.add_argument("--no-central-well-leg", action="store_true", help="skip the untimed ~0.3 s behind the run that measures the HBM-bound CENTRAL_WELL step at 4 Mi and 16 Mi bodies (single GPU)")
    ap.add_argument("--overlap", action="store_true", help="sharded mode: own-segment launch overlapped with the all-gather")
    ap.add_argument("--trial-seconds", type=float, default=90.0,
                    help="wall-time budget of the exchange trial (N > 1, --gather auto): once it is spent the candidates not yet tried are skipped")
    ap.add_argument("--test-inject-push-failure", action="store_true",
                    help="testing only: rank 1 corrupts ONE pushed position in the timed region (MAPN_TEST_HOOKS): the run must notice and fall back")
    ap.add_argument("--force-comm", action="store_true",
                    help="create the torch.distributed group and the in-library RCCL communicator even for one rank (exercises the sharded code path on a 1-GPU box)")
    return ap.parse_args()


def cpu_baseline(n, seed, target_seconds):
    """The CPU restatement (oracle, kind 'port': the reference has no CPU path) timed
    on the host cores of this box on a bounded sample of the same workload: whole steps of the
    N-body state (or, on a small host, a slice of one step), all host threads."""
    from oracle import Oracle, OracleSim, Params
    o = Oracle()
    cores = o.hardware_threads()
    pos, vel = o.initial_state(n, seed=seed)
    prm = Params(mass=70000.0 / n)
    k = min(n, 256 * cores)
    t0 = time.perf_counter(); o.step_slice(pos, vel, 0, k, params=prm); t = time.perf_counter() - t0
    if (k * n / t) * target_seconds >= 2.0 * n * n:    # a whole step fits the budget comfortably
        sim = OracleSim(o, pos, vel, params=prm)
        sim.simulate(steps=1)                          # warm up threads / caches
        # time-bounded, not count-bounded: a two-step calibration underestimated the steady step time 3x
        # on the 256-thread box (965 steps, 39.8 s for a 12 s target)
        steps, t0 = 0, time.perf_counter()
        while True:
            sim.simulate(steps=4)
            steps += 4
            t = time.perf_counter() - t0
            if t >= target_seconds:
                break
        pairs = float(steps) * n * n
        what = f"{steps} whole steps of {n} bodies"
    else:
        k = max(16, int(n * (k * n / t) * target_seconds / (float(n) * n)) // 16 * 16)
        k = min(k, n)
        t0 = time.perf_counter(); o.step_slice(pos, vel, 0, k, params=prm); t = time.perf_counter() - t0
        pairs = float(k) * n
        what = f"bodies [0,{k}) of {n} against all {n}, 1 step"
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {"value": pairs / t, "unit": "body-pair interactions/s", "cores": cores, "kind": "port",
            "sample": f"oracle (C, fp32, exact HLSL op order, vectorised over i): {what}, {cores} threads, {t:.1f} s",
            "cpu": model}


def kernel_source_sha16():
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256()
    try:
        for f in ("mapn_kernels.hip", "mapn_sym.hip"):
            h.update(open(os.path.join(here, "multi-adapter-particles_amd", "csrc", f), "rb").read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def pmc_traffic(kernel_name, n, world):
    """HBM bytes per force launch from the committed PMC passes (profiles/*_pmc_summary.json,
    collected with separate rocprofv3 --pmc runs of this same command and corrected as the
    MI355X guide prescribes: 2 x FETCH_SIZE + WRITE_SIZE).  PMC counters cannot be read from
    inside an un-profiled run, so this is the profiled value for the default 65 536-body
    single-GPU workload, or None for any other configuration -- and None when the summary was taken
    from a different kernel source than the one running (its sha is stored in the summary)."""
    if n != 65536 or world != 1:
        return None, None
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    sha = kernel_source_sha16()
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("_kernel_source_sha16") != sha:
            continue                                   # stale: measured on another version of the kernels
        for k, v in d.items():
            if isinstance(v, dict) and kernel_name in k and "hbm_bytes_per_launch" in v:
                pmc_traffic.valu_busy = v.get("valu_busy_fraction")
                return v["hbm_bytes_per_launch"], os.path.relpath(f, here)
    return None, None


pmc_traffic.valu_busy = None    # VALU busy fraction of the same (sha-matched) PMC summary, if it holds one


def power_sensor(device_index, root="/sys/class/drm"):
    """The hwmon directory of the GPU the HIP runtime calls `device_index` (amdgpu: power1_input / power1_average in microwatts, power1_cap,
    freq1_input = shader clock in Hz), or None.  Matched by PCI address (hipDeviceGetPCIBusId against the sysfs device directory: the
    driver's 1-GPU boxes show the sensors of all of the host's GPUs); a box with ONE sensor is taken as that GPU."""
    import glob
    cands = []
    for hw in sorted(glob.glob(os.path.join(root, "card*", "device", "hwmon", "hwmon*"))):
        if any(os.access(os.path.join(hw, f), os.R_OK) for f in ("power1_input", "power1_average")):
            cands.append(hw)
    if not cands:
        return None
    try:
        import ctypes
        hip = None
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):     # (already in the process: libmapn is linked against it)
            try:
                hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        buf = ctypes.create_string_buffer(64)
        if hip is not None and hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0:
            want = buf.value.decode().lower()               # "0000:5a:00.0"
            for hw in cands:
                if os.path.basename(os.path.realpath(os.path.join(hw, "..", ".."))).lower() == want:
                    return hw
    except (OSError, AttributeError, ValueError):
        pass
    return cands[0] if len(cands) == 1 else None


def read_sensor(hw, names):
    for f in names:
        try:
            return float(open(os.path.join(hw, f)).read().strip())
        except (OSError, ValueError):
            continue
    return None


def power_leg(step_batch, hw, seconds=2.5, batch_ms=100.0, ms_per_step=1.0):
    """What the package draws under the timed kernel: ~`seconds` more of the same steps (untimed, behind everything that is), the
    sensor read between batches of ~`batch_ms`.  The figures are those of the second half (the sensor's own averaging has settled)."""
    per = max(1, int(batch_ms / max(ms_per_step, 1e-3)))
    watts, mhz = [], []
    t0 = time.perf_counter()
    t_end, steps = t0 + seconds, 0
    while time.perf_counter() < t_end:
        step_batch(per)
        steps += per
        w = read_sensor(hw, ("power1_input", "power1_average"))
        f = read_sensor(hw, ("freq1_input",))
        if w is not None:
            watts.append(w / 1e6)
        if f is not None:
            mhz.append(f / 1e6)
    leg_ms = (time.perf_counter() - t0) * 1e3 / max(steps, 1)       # (includes the sensor reads: a few per cent above the timed figure)
    if len(watts) < 4:
        return None
    tail = sorted(watts[len(watts) // 2:])
    cap = read_sensor(hw, ("power1_cap",))
    out = {"package_w": round(tail[len(tail) // 2], 1), "package_w_max": round(max(watts), 1), "cap_w": round(cap / 1e6, 1) if cap else None,
           "samples": len(watts), "seconds": seconds, "ms_per_step_during": round(leg_ms, 5)}
    if mhz:
        tm = sorted(mhz[len(mhz) // 2:])
        out["sensor_sclk_mhz"] = round(tm[len(tm) // 2], 0)
    if out["cap_w"]:
        out["at_cap"] = bool(out["package_w"] >= 0.98 * out["cap_w"])
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
            # no launcher around this process: start the ranks here (before anything touches the GPU) and relay rank 0's line
            sys.exit(launch_ranks(a.gpus))
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
    red_dev = "cuda" if a.dist_backend == "nccl" else "cpu"     # where small reduction tensors live

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
    c = mapn.Compute(n, device=local_rank, force_mode=mode, mass=70000.0 / n, seed=a.seed,
                     rank=rank, world_size=world, flags=flags, kernel=kern)
    created_note = mapn.load_library().mapn_last_error().decode(errors="replace")    # (what mapn_create left behind: why a calibration did not stay)
    info = device_info(local_rank)
    transport = "none"
    gather_fn = None
    if dist is not None:
        transport = a.transport
        if transport == "rccl" and a.gather not in ("p2p", "flow", "sym", "sympush", "p2pall"):
            try:
                c.comm_init_torch()
            except Exception as e:     # RCCL-in-library unavailable: use torch's RCCL instead, loudly
                print(f"[bench rank {rank}] native RCCL transport failed ({e}); falling back to torch.distributed all-gather",
                      file=sys.stderr, flush=True)
                transport = "torch"
        if transport == "torch":
            c.set_external_gather(True)
            gather_fn = make_torch_gather(c, torch, dist, n, rank, world)
    def apply_plan():
        if a.plan:
            kname, k, w, sb, fused = a.plan.split(",")
            c.set_force_plan({"lds": mapn.KERNEL_LDS, "sgpr": mapn.KERNEL_SCALAR}[kname], int(k), int(w), int(sb), int(fused))

    apply_plan()
    # each hipEventRecord costs ~4 us of queue time: at most one event PAIR per 4 steps, also for short runs
    timer_interval = a.timer_interval if a.timer_interval >= 0 else (8 if a.steps >= 32 else 4)
    if a.timer_interval < 0 and world > 1:
        # a sharded step is 0.09 ms at 65 536 / 8, an event pair 8 us of it: two sampled steps in a short region, every 16th in a long one
        timer_interval = 16 if a.steps >= 32 else max(4, a.steps // 2)
    c.set_timers(timer_interval)

    def step():
        fence = c.GetFenceValue()             # Particles.cpp:446-448
        c.Simulate(n, fence)
        if gather_fn:
            gather_fn()

    def sync():
        c.WaitForGpu()
        if torch is not None:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def rebuild(with_p2p):
        """A context whose device-side wait timed out or whose row / position check failed stays failed: replace it (collective: all ranks)."""
        nonlocal c
        c.close()
        c = mapn.Compute(n, device=local_rank, force_mode=mode, mass=70000.0 / n, seed=a.seed, rank=rank, world_size=world, flags=flags,
                         kernel=kern)               # (the SAME kernel choice and plan as asked for: ADVICE r2)
        if a.gather not in ("p2p", "flow", "sym", "sympush", "p2pall"):
            c.comm_init_torch()
        if with_p2p:
            c.p2p_setup_torch()
            c.set_timeouts(p2p_ms=a.p2p_timeout_ms)
        apply_plan()
        c.set_timers(timer_interval)

    gather_algo = "n/a"
    trial = {}

    def replicas_consistent():
        """Every rank keeps a full replica of the positions; after any correct exchange they are
        bit-identical.  Compares a checksum of both ping-pong buffers across ranks."""
        try:
            sums = list(c.replica_checksum())            # mapn_replica_checksum: one C call (drains, reports failed device-side waits)
        except mapn.MapnError as e:                      # (every rank must still take part in the collective below)
            print(f"[bench rank {rank}] replica check: {e}", file=sys.stderr, flush=True)
            sums = None
        allsums = [None] * world
        dist.all_gather_object(allsums, sums)
        return all(x is not None and x == allsums[0] for x in allsums)

    p2p_failure = None
    if dist is not None and transport == "rccl":
        # every way of issuing the exchange that sets up on this node: (name, algorithm, overlap structure)
        # The DEFAULT trial (--gather auto) holds the forms that can win, safest first: the RCCL collectives, the symmetric step over
        # RCCL alone, then the peer-to-peer forms in the order they build on each other.  The overlap structures (two under-filled
        # launches: 150 against 120 us in loopback) and the in-kernel exchange lose to these by the builder's own numbers and are
        # run only when asked for by name (--overlap, --gather flow / p2pall): fewer code paths in the one run that counts.
        candidates = []
        if a.gather in ("auto", "allgather"):
            candidates += [("allgather", 0, False)] + ([("allgather+overlap", 0, True)] if a.overlap or a.gather == "allgather" else [])
        if a.gather in ("auto", "sendrecv"):
            candidates += [("sendrecv", 1, False)] + ([("sendrecv+overlap", 1, True)] if a.overlap or a.gather == "sendrecv" else [])
        if a.overlap:                                   # --overlap: only the overlap structures
            candidates = [x for x in candidates if x[2]]
        p2p_ok = False
        sym_fits = mode == mapn.FORCE_ALL_PAIRS and (n // world) % 1024 == 0 and n % world == 0
        if a.gather in ("auto", "symrccl") and world > 1 and sym_fits:
            # the sharded symmetric step over RCCL alone: pack launch, grouped send/recv of the reaction rows, reduce launch, all-gather
            candidates.append(("rccl+symmetric", 6, False))
        if a.gather in ("auto", "p2p", "flow", "sym", "sympush", "p2pall") and world > 1:
            try:
                c.p2p_setup_torch()
                c.set_timeouts(p2p_ms=a.p2p_timeout_ms)
                ok = torch.tensor([1], device=red_dev)
            except Exception as e:
                print(f"[bench rank {rank}] p2p setup failed: {e}", file=sys.stderr, flush=True)
                ok = torch.tensor([0], device=red_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # all ranks or none
            p2p_ok = bool(ok.item())
            # LAST: these have to prove themselves on this node
            if p2p_ok and a.gather in ("auto", "p2p", "sym", "sympush", "p2pall"):
                candidates.append(("p2p", 2, False))               # (also the yardstick the symmetric forms are verified against)
            if p2p_ok and a.gather in ("flow", "p2pall"):
                candidates.append(("p2p+inkernel", 3, False))  # the same exchange overlapped inside the force launch (by name only)
            if p2p_ok and a.gather in ("auto", "sym", "p2pall") and sym_fits:
                # the SYMMETRIC step sharded over the ranks: every unordered pair of the job once, reactions
                # stored into the owners' receive regions, positions pulled by the same launch
                candidates.append(("p2p+symmetric", 4, False))
            if p2p_ok and a.gather in ("auto", "sympush", "p2pall") and sym_fits:
                # ... the same with the new positions PUSHED into the peers' replicas (posted writes instead of read round trips)
                candidates.append(("p2p+symmetric+push", 5, False))

        def select(name, algo, overlap):
            c.set_gather_algorithm(algo)
            c.set_shard_overlap(overlap)

        def symmetric_deviation(sym_algo):
            """Four steps from the initial state with the sharded symmetric step against the one-sided sharded step over the same
            kind of transport (peer-to-peer pull for 4 / 5, RCCL all-gather for 6)."""
            import numpy as np
            got = []
            for algo in ((0 if sym_algo == 6 else 2), sym_algo):
                sync()
                c.set_gather_algorithm(algo)
                pos0, vel0 = mapn.generate_initial_state(n, seed=a.seed)
                c.upload_state(pos0, vel0)
                sync()
                for _ in range(4):
                    step()
                # ALL ranks must have finished before anyone re-initialises: a rank that is a step ahead would overwrite the
                # buffers a slower peer is still pulling from / pushing into (seen with 8 ranks time-slicing one device)
                sync()
                got.append(c.download_state()[0][:, :3].astype(np.float64))
            return float(np.linalg.norm(got[0] - got[1], axis=1).max() / 400.0)

        def reinit():
            c.set_gather_algorithm(0 if a.gather not in ("p2p", "flow", "sym", "sympush", "p2pall") else 2)
            c.set_shard_overlap(False)
            pos0, vel0 = mapn.generate_initial_state(n, seed=a.seed)
            c.upload_state(pos0, vel0)
            sync()

        if world > 1 and len(candidates) > 1:
            # time every way (same bytes) on untimed steps; every rank must take the same decision -> MAX
            # over ranks.  The peer-to-peer kernel must also PROVE itself here: no timed-out wait (the
            # library now reports one as MAPN_ERR_COMM) and bit-identical replicas on all ranks.
            p2p_dead = False
            t_trial0 = time.perf_counter()
            for name, algo, overlap in candidates:
                is_p2p = 2 <= algo <= 5
                if is_p2p and p2p_dead:
                    continue
                over = torch.tensor([1 if (trial and time.perf_counter() - t_trial0 > a.trial_seconds) else 0], device=red_dev)
                dist.all_reduce(over, op=dist.ReduceOp.MAX)     # (all ranks decide alike)
                if over.item():
                    if rank == 0:
                        print(f"[bench] exchange trial: budget of {a.trial_seconds:.0f} s spent -> '{name}' not tried", file=sys.stderr, flush=True)
                    continue
                failed = None
                try:
                    select(name, algo, overlap)
                    dist.barrier()                          # the device-side waits are bounded (--p2p-timeout-ms): start together
                    for _ in range(5):
                        step()
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(30):
                        step()
                    sync()
                    dt_trial = time.perf_counter() - t0
                except mapn.MapnError as e:
                    failed, dt_trial = str(e), float("inf")
                bad = torch.tensor([1 if failed else 0], device=red_dev)
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                if not bad.item() and algo >= 2:
                    bad = torch.tensor([0 if (c.p2p_status() == 0 and replicas_consistent()) else 1], device=red_dev)
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                    if bad.item() and not failed:
                        failed = "replicas differ across ranks"
                if not bad.item() and algo in (4, 5, 6):
                    # identical replicas do not show that the reactions ARRIVED: compare four steps from the
                    # initial state with the one-sided sharded step (verified above)
                    dev = symmetric_deviation(algo)
                    bad = torch.tensor([0 if dev < 1e-5 else 1], device=red_dev)
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                    if bad.item():
                        failed = f"symmetric sharded step deviates from the one-sided one by {dev:.2e} of the spread after 4 steps"
                if bad.item():
                    if rank == 0:
                        print(f"[bench] exchange '{name}' failed on this node ({failed or 'on another rank'}) -> not used; state re-initialised",
                              file=sys.stderr, flush=True)
                    # a context whose device-side wait gave up or whose row / position check failed STAYS failed, whatever the
                    # algorithm (6 sets the status word too): replace it on every rank, or the next candidate inherits the fault
                    stuck = torch.tensor([1 if c.p2p_status() != 0 else 0], device=red_dev)
                    dist.all_reduce(stuck, op=dist.ReduceOp.MAX)
                    if is_p2p:
                        p2p_failure = f"{name}: {failed or 'failed on another rank'}"
                        p2p_dead = algo == 2                    # the plain exchange failed on a healthy context: every peer-to-peer form shares its transport
                    if is_p2p or stuck.item():
                        rebuild(with_p2p=p2p_ok and not p2p_dead and ("p2p" in trial or not is_p2p))
                    else:
                        reinit()
                    continue
                t = torch.tensor([dt_trial], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                trial[name] = float(t.item()) / 30
            gather_algo = min(trial, key=trial.get) if trial else "allgather"
            for asked, nm in (("sym", "p2p+symmetric"), ("sympush", "p2p+symmetric+push"), ("symrccl", "rccl+symmetric")):
                if a.gather == asked and nm in trial:
                    gather_algo = nm                        # asked for by name: it only had to pass its check
            if rank == 0:
                print("[bench] exchange trial: " + ", ".join(f"{k} {v*1e6:.1f} us/step" for k, v in trial.items()) + f" -> {gather_algo}",
                      file=sys.stderr, flush=True)
        else:
            gather_algo = candidates[0][0] if candidates else "allgather"
            if gather_algo.startswith("p2p") and not p2p_ok:
                sys.exit("bench: --gather p2p requested but the peer-to-peer setup failed")

        if a.gather in ("sym", "sympush", "symrccl") and "symmetric" not in gather_algo:
            sys.exit(f"bench: --gather {a.gather}: the sharded symmetric step does not apply (N / ranks must be a multiple of 1024) or failed its check")
        chosen = {x[0]: x for x in candidates}.get(gather_algo, (gather_algo, 0, False))
        c.set_gather_algorithm(chosen[1])
        c.set_shard_overlap(chosen[2])
        transport = "p2p (hipIpc + device flags)" if 2 <= chosen[1] <= 5 else "rccl"
    prewarm_steps = 0
    if a.prewarm_ms > 0:
        # same work as a timed step, just not timed: lets the clock settle so that a short K does
        # not measure the ramp (1.18 ms/step over the first 5 steps vs 0.88 ms steady state)
        if dist is None:
            t_end = time.perf_counter() + a.prewarm_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(16):
                    step()
                c.WaitForGpu()
                prewarm_steps += 16
        else:
            # sharded: every step holds a collective, so all ranks must run the SAME number of
            # steps -- a fixed count (about prewarm_ms at the single-GPU step time / world)
            prewarm_steps = max(16, int(a.prewarm_ms * 1e-3 / (0.9e-3 / world * (n / 65536.0) ** 2)) // 16 * 16)
            prewarm_steps = min(prewarm_steps, 4096)
            for _ in range(prewarm_steps):
                step()
            c.WaitForGpu()
    # XCD-aware parts (1 GPU, symmetric kernel): the eight dies do not run at one speed and a launch gives each the same work.
    # Untimed: calibrate, then an A/B of the weighted plan against the default one; the weights stay only if they win.
    xcd = {"mode": a.xcd, "weights": None, "used": False, "source": None}
    if given_w is not None:
        c.set_sym_xcd_weights(given_w)
        plw = c.sym_plan()
        xcd.update({"mode": "given", "weights": list(plw.xcd_weight) if plw.xcd_mode else given_w, "used": plw.xcd_mode != 0, "source": "--xcd-weights (no calibration, no A/B)",
                    "form": {0: None, 1: "spread", 2: "class-aware"}.get(plw.xcd_mode)})
    if dist is not None and world > 1 and "symmetric" in gather_algo and a.xcd != "off" and not a.plan:
        # SHARDED symmetric step: the library has planned every rank's launch with the die weights of ITS GPU (MAPN_FLAG_XCD_CALIBRATE:
        # a temporary unsharded context's calibration when the step was prepared -- no collective in it, and every die holds heavy and
        # light blocks there, so the measurement is of the dies, not of the blocks' classes); an untimed A/B against the unweighted plan,
        # MAX over ranks, decides for all of them.  Loopback at 65 536 / 8: -1.2 ... -1.5 % per step before the heavy blocks were moved to
        # the odd dispatch slots by default, less since.
        try:
            def burst_all(k):
                sync(); t0 = time.perf_counter()
                for _ in range(k):
                    step()
                sync()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item()) / k
            kk = max(20, min(400, int(0.05 / (0.1e-3 * (n / 65536.0) ** 2 * 8 / world))))
            pl = c.sym_plan()
            w = list(pl.xcd_weight)
            flag = torch.tensor([1 if pl.xcd_mode != 0 else 0], device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)            # all ranks or none
            if flag.item():
                t_w = min(burst_all(kk), burst_all(kk))
                c.set_sym_xcd_weights(None)
                t_def = min(burst_all(kk), burst_all(kk))
                xcd.update({"weights": w, "source": "library (MAPN_FLAG_XCD_CALIBRATE: a temporary unsharded context on every rank's GPU; rank 0's weights shown)",
                            "form": {1: "spread", 2: "class-aware"}.get(pl.xcd_mode), "trial_ms": {"default": t_def * 1e3, "weighted": t_w * 1e3}})
                if a.xcd == "on" or t_w < t_def * 0.998:
                    c.set_sym_xcd_weights(w)
                    xcd["used"] = c.sym_plan().xcd_mode != 0
            else:
                xcd["note"] = "the library's calibration did not apply on every rank (a rank's share must be a multiple of 8 blocks)"
                c.set_sym_xcd_weights(None)
        except mapn.MapnError as e:                            # (a failure here leaves the default plan: the run goes on)
            xcd["error"] = str(e)[:200]
            try:
                c.set_sym_xcd_weights(None)
            except mapn.MapnError:
                pass
    if xcd_by_library:
        try:
            def burst(k):
                c.WaitForGpu(); t = time.perf_counter()
                for _ in range(k):
                    step()
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
    # The closing collective of a timed region (N > 1): made BEFORE any timed region, never inside one.
    #   * RCCL backend: the barrier is an all-reduce of one pre-made device word enqueued, stream-ordered, BEHIND this rank's K steps on the
    #     library's compute stream (torch.cuda.ExternalStream) while the device is still working on them -- its host-side launch cost and
    #     the peers' arrival overlap the steps, and what is left after the last rank's last step is the collective's kernel and one host
    #     sync.  The verdict (did a device-side check fail on any rank?) travels in a second collective AFTER the region is closed.
    #   * gloo (tests: ranks sharing one device, host tensors): one host-side all-reduce is barrier and verdict.
    verdict = torch.zeros(1, dtype=torch.int32, device=red_dev) if dist is not None else None
    barrier_word = torch.zeros(1, dtype=torch.int32, device="cuda") if (dist is not None and a.dist_backend == "nccl") else None
    ev_first = torch.cuda.Event(enable_timing=True) if barrier_word is not None else None
    ev_last = torch.cuda.Event(enable_timing=True) if barrier_word is not None else None
    stream_cache = {}

    def compute_stream_of(ctx):
        """The library's compute stream as a torch stream (one wrapper per context: a rebuilt context has another stream)."""
        ptr = ctx.compute_stream
        if ptr not in stream_cache:
            stream_cache.clear()
            stream_cache[ptr] = torch.cuda.ExternalStream(ptr, device=torch.device("cuda", local_rank))
        return stream_cache[ptr]

    def run_steps(k, recoverable):
        """k steps, then the barrier + device sync.  run_steps.closed_at: the moment the region closed (barrier passed, device idle);
        run_steps.idle_s: how long THIS rank's k steps took without the closing collective (RCCL backend: HIP events on the compute stream
        around them; gloo: wall clock until its device was idle).  With a peer-to-peer form on N > 1 ranks a device-side check or wait that
        fails on ANY rank (they are all bounded: every rank gets out of its own) comes back as text on EVERY rank -- the verdict collective
        carries it -- instead of leaving the others in a barrier for ever."""
        fail = None
        t_begin = time.perf_counter()
        ordered = None
        if barrier_word is not None and gather_fn is None:
            try:
                ordered = compute_stream_of(c)
            except Exception as e:                             # (no stream wrapper: the host-side form below)
                print(f"[bench rank {rank}] stream-ordered closing barrier unavailable ({e}); closing on the host", file=sys.stderr, flush=True)
        issued = False
        try:
            if ordered is not None:
                ev_first.record(ordered)
            for _ in range(k):
                step()
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

    fallback_after_failure = None
    while True:
        recoverable = dist is not None and world > 1 and transport.startswith("p2p")
        fail = run_steps(a.warmup, recoverable)
        if not fail:
            if a.test_inject_push_failure and rank == 1 and fallback_after_failure is None:
                os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_TEST_CORRUPT_PUSH"] = "once"
            c.set_timers(timer_interval)          # restart the sampling phase: the first timed step carries events
            c.kernel_stats(reset=True)
            t0 = time.perf_counter()
            fail = run_steps(a.steps, recoverable)
            elapsed = run_steps.closed_at - t0                 # barrier + device sync on both sides of exactly K steps
            elapsed_idle = min(run_steps.idle_s, elapsed)      # (N > 1: without the closing collective -- reported beside the contract's figure, never instead of it)
        if not fail:
            break
        # A peer-to-peer form that had passed its trial failed in the run itself (a pushed position that did not match its checksum, a
        # reaction row that never arrived whole, a wait that gave up).  ONCE, the fastest OTHER form the trial verified takes over: new
        # contexts on all ranks (a failed one stays failed), the seeded initial state, warm-up and the K timed steps again.
        others = sorted((k for k in trial if k != gather_algo), key=trial.get)
        if fallback_after_failure is not None or not others:
            sys.exit(f"bench: exchange '{gather_algo}' failed during the run ({fail}) and no verified form is left to fall back to")
        fallback_after_failure = f"'{gather_algo}' failed during the run: {fail[:300]}"
        if rank == 0:
            print(f"[bench] {fallback_after_failure} -> falling back to '{others[0]}'", file=sys.stderr, flush=True)
        gather_algo = others[0]
        chosen = {x[0]: x for x in candidates}[gather_algo]
        rebuild(with_p2p=2 <= chosen[1] <= 5)
        c.set_gather_algorithm(chosen[1])
        c.set_shard_overlap(chosen[2])
        transport = "p2p (hipIpc + device flags)" if 2 <= chosen[1] <= 5 else "rccl"
        xcd["used"] = False
    if dist is not None:
        t = torch.tensor([elapsed, elapsed_idle], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_idle = float(t[0].item()), float(t[1].item())

    st = c.kernel_stats()
    # how the step time is spread over the timed region: the steps that carried HIP events (every timer_interval-th), by quarter
    quarters = None
    try:
        idx, step_ms, _ = c.step_samples()
        if len(idx):
            quarters = []
            for qn in range(4):
                m = (idx >= a.steps * qn // 4) & (idx < a.steps * (qn + 1) // 4)
                quarters.append(round(float(step_ms[m].mean()), 5) if m.any() else None)
            spread = {"steps_timed": int(len(idx)), "min_ms": round(float(step_ms.min()), 5), "median_ms": round(float(sorted(step_ms)[len(idx) // 2]), 5),
                      "max_ms": round(float(step_ms.max()), 5)}
    except mapn.MapnError:
        pass
    first, count = c.shard_range()
    # the clock the chip held under this kernel: stamped diagnostic steps right behind the timed region
    # (same state of the chip; untimed).  Single GPU, scalar-cache kernel only.
    clock = None
    if world == 1 and a.mode == "all_pairs" and dist is None:
        try:
            clock = c.measure_clock(8)
        except mapn.MapnError as e:
            print(f"[bench] clock measurement unavailable: {e}", file=sys.stderr, flush=True)
    consistent = None
    sym_dev_after = None
    if dist is not None and world > 1 and gather_algo != "n/a":
        consistent = replicas_consistent() and c.p2p_status() == 0      # (the status word serves every device-side wait)
        if consistent and "symmetric" in gather_algo:
            # once more, after the timed run: the sharded symmetric step against the one-sided one
            dev_after = symmetric_deviation({"p2p+symmetric": 4, "p2p+symmetric+push": 5, "rccl+symmetric": 6}[gather_algo])
            worst = torch.tensor([dev_after], dtype=torch.float64, device=red_dev)
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
            sym_dev_after = float(worst.item())
            consistent = sym_dev_after < 1e-5 and c.p2p_status() == 0
        if not consistent and rank == 0:
            print("[bench] WARNING: position replicas differ across ranks after the run (or the sharded symmetric step "
                  "failed its check) -- the exchange misbehaved; this result is INVALID", file=sys.stderr, flush=True)
    # SURVEY 8(d) asks for >= 100 steps and a median of five repeats: four more regions of the same K steps follow the timed one (untimed by
    # the contract: `value` stays the first region's) and the five are listed with their median -- how far one region is from the typical one
    # on this box.  Done for every K (the driver's own command has K = 20; `meets_survey_8d` says whether K reaches the 100 steps).  They run
    # AFTER the validity checks of the timed run, and a failure in them (N > 1, peer-to-peer form) only ends the list: it cannot taint the result.
    repeats = None
    if a.steps >= 1 and consistent is not False:
        reps = [elapsed / a.steps * 1e3]
        recoverable = dist is not None and world > 1 and transport.startswith("p2p")
        for _ in range(4):
            sync()
            t0 = time.perf_counter()
            if run_steps(a.steps, recoverable):
                break
            dt_rep = run_steps.closed_at - t0
            if dist is not None:
                t = torch.tensor([dt_rep], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_rep = float(t.item())
            reps.append(dt_rep / a.steps * 1e3)
        repeats = {"ms_per_step": [round(x, 5) for x in reps], "median_ms_per_step": round(sorted(reps)[len(reps) // 2], 5), "meets_survey_8d": a.steps >= 100,
                   "note": "the timed region (first entry: `ms_per_step`, `value`) and four more regions of the same K steps behind it (after the run's validity checks)"}
    # what the package draws under this kernel (single GPU; untimed, behind everything that is timed or checked): the symmetric kernel
    # runs AT the board's power cap -- the clock it holds (`held_clock_ghz`) is what the cap leaves, which is why `frac` (priced at the
    # nominal 2.4 GHz) stops near 0.92 while `frac_at_held_clock` is 1.0
    power = None
    if world == 1 and dist is None and not a.no_power_leg:
        try:
            hw = power_sensor(local_rank)
            if hw:
                def step_batch(k):
                    for _ in range(k):
                        c.Simulate(n, c.GetFenceValue())
                    sync()
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
    # CSMain as shipped at the reference's maximum (defines.h:45: 4 194 304 bodies -- 235 MB per step, inside the 256 MiB Infinity Cache) and at
    # 16 777 216 bodies (940 MB per step: past every cache, the kernel's non-temporal form).  Untimed, behind everything that is; ~0.1 s of steps each.
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
        pairs_per_step = float(n) * float(n) if a.mode == "all_pairs" else float(n)
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
                       "transport": transport, "exchange": gather_algo, "exchange_trial_us_per_step": {k: v * 1e6 for k, v in trial.items()},
                       "replicas_bit_identical_after_run": consistent, "valid": consistent is not False, "prewarm_steps_untimed": prewarm_steps, "seed": a.seed, "mass": "70000/N", "device": (info.name.decode() or "MI355X") + " / " + info.arch.decode(),
                       "kernel": st.kernel_name.decode(), "bodies_per_lane": st.bodies_per_lane, "j_splits": st.j_splits,
                       "grid": [st.grid_x, st.grid_y], "block": st.block_x, "fused_integrator": bool(st.fused),
                       "epilogue": {0: "partial rows + reduce_integrate launch", 1: "fused in the workgroup", 2: "last-arriver ticket (one launch per step)", 3: "symmetric kernel: force rows + sym_reduce_integrate launch"}.get(st.epilogue, "?"),
                       "launches_per_step": int(st.force_launches_per_step) * (1 if st.fused else 2), "timer_interval": timer_interval,
                       "step_ms_by_quarter_of_the_timed_region": quarters, "step_ms_spread": spread if quarters else None,
                       "repeats": repeats,
                       "step_ms_note": "device time (HIP events on the compute stream) of the steps of the timed region that carried events; ms_per_step is the wall clock over all of them",
                       "p2p_failure": p2p_failure, "fallback_after_failure": fallback_after_failure, "sharded_symmetric_deviation_after_run": sym_dev_after,
                       "xcd_aware_parts": xcd, "symmetric_plan": sym_plan_desc, "replay": replay, "central_well": central_well, "partial_active": partial,
                       "closing_collective_us": ((elapsed - elapsed_idle) * 1e6) if dist is not None else None},
        }
        if a.mode == "all_pairs":
            peak = info.peak_fp32_flops / 1e12
            if st.launches and st.avg_seconds > 0:
                pairs_per_launch = float(count) * float(n)
                ach = FLOP_PER_PAIR * pairs_per_launch / st.avg_seconds / 1e12
                traffic, traffic_src = pmc_traffic(st.kernel_name.decode(), n, world)
                held = clock.shader_clock_ghz if clock else None
                sym = st.kernel_name.decode() == "force_sym_kernel"
                # flop the kernel EXECUTES per ordered interaction: the one-sided kernels evaluate every ordered
                # pair (20); the symmetric kernel evaluates every unordered pair once and feeds both bodies:
                # 3 sub + 3 fma + rsq + 2 mul + 3 fma + 3 fma = 24 flop per TWO interactions
                executed_per_pair = 12.0 if sym else float(FLOP_PER_PAIR)
                out["roofline"] = {"bound": "mfma",
                                   "bound_actual": "valu-fp32",
                                   "bound_detail": "compute-bound on the fp32 VECTOR ALU (packed v_pk_*_f32 + v_rsq_f32): the kernel issues NO MFMA. "
                                                   "`bound` holds the schema's compute value because of its two bounds (hbm | mfma) the compute one "
                                                   "applies and the dense f32 MFMA peak is the same number as the fp32 vector peak (157.3 TF); "
                                                   "`bound_actual` names the unit that is really the limit",
                                   "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                   "held_clock_ghz": held,
                                   "held_clock_ghz_p10_p90": [clock.shader_clock_ghz_p10, clock.shader_clock_ghz_p90] if clock else None,
                                   "frac_at_held_clock": (ach / (info.compute_units * held * 1e9 * 256 / 1e12)) if held else None,
                                   "flop_executed_per_pair": executed_per_pair,
                                   "frac_executed": ach / peak * executed_per_pair / FLOP_PER_PAIR,
                                   "power": power,
                                   "power_note": ("package power (hwmon power1_input) and sensor clock over ~2.5 s of the same steps run behind the timed region, second half of the "
                                                  "samples; `at_cap`: the kernel runs at the board's power limit, so the clock it holds -- and with it `frac`, priced at the "
                                                  "nominal clock -- is set by the energy a pair costs, not by issue cycles (null: no readable sensor / --no-power-leg)"),
                                   "picojoule_per_interaction": (power["package_w"] * power["ms_per_step_during"] * 1e-3 / pairs_per_step * 1e12) if power else None,
                                   "valu_busy": pmc_traffic.valu_busy,
                                   "valu_busy_note": "4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / cycles of the committed PMC pass of these kernel sources (null: no pass on record for them)",
                                   "instruction_mix_ceiling": ("symmetric kernel: 14 packed ops x 4 cycles + 2 v_rsq_f32 x 8 cycles per 256 ordered interactions per SIMD "
                                                               "(+ 9 ds_bpermute_b32 per 16 per lane, no VALU cycles) = 111 % of the ALGORITHMIC peak at any clock: every "
                                                               "unordered pair is evaluated once (Newton's third law), so `frac` counts 20 flop per ordered pair of "
                                                               "which 12 are executed (`frac_executed`)") if sym else
                                                              "11 packed ops x 4 cycles + 2 v_rsq_f32 x 8 cycles per 128 pairs per SIMD = 66.7 % of peak at any clock",
                                   "traffic": traffic, "traffic_unit": "HBM bytes per force launch (2*FETCH_SIZE+WRITE_SIZE, PMC)",
                                   "traffic_note": ("positions (16 N) read once per XCD L2 + the force rows written once, 12 bytes per body: one 12 KiB row per workgroup for its 1024 bodies i and one "
                                                    "768 B row per meeting for the 64 travelling bodies j (no float atomics: a fixed-order reduce launch adds them) -- "
                                                    "3 N^2/512 + 12 N x parts bytes, ~0.1 TB/s, 1.3 % of the HBM roofline: the kernel is compute-bound") if sym else
                                                   "positions read once per XCD L2 + the partial rows of the j-split written and read back once by the last-arriver epilogue",
                                   "traffic_source": traffic_src, "kernel_source_sha16": kernel_source_sha16(),
                                   "kernel": st.kernel_name.decode(), "launches_timed": int(st.launches),
                                   "avg_launch_ms": st.avg_seconds * 1e3, "flop_per_pair": FLOP_PER_PAIR,
                                   "pairs_per_launch": pairs_per_launch,
                                   "algorithmic_hbm_bytes_per_launch": HBM_BYTES_PER_BODY * count,
                                   "algorithmic_hbm_GBps": HBM_BYTES_PER_BODY * count / st.avg_seconds / 1e9,
                                   "note": "peak = CUs x clock x 256 flop/clk (fp32 vector = dense f32 MFMA peak, 157.3 TF)"}
            else:
                out["roofline"] = None
        else:
            if st.launches == 0:
                pass
            out["roofline"] = {"bound": "hbm", "achieved": HBM_BYTES_PER_BODY * n * a.steps / elapsed / 1e9, "peak": 8000.0,
                               "unit": "GB/s", "frac": HBM_BYTES_PER_BODY * n * a.steps / elapsed / 8e12, "traffic": None,
                               "reachable_GBps": 6290.0, "frac_of_reachable": HBM_BYTES_PER_BODY * n * a.steps / elapsed / 6.29e12,
                               "note": ("central-well step, 56 B per body: `achieved` is wall-clock over the K steps; `peak` the 8.0 TB/s specification, `reachable_GBps` what a "
                                        "float4 copy measures on this part (MI355X guide); up to ~4.5 Mi bodies a step's state is resident in the 256 MiB Infinity Cache (a cache "
                                        "rate, not an HBM one); from 6 Mi bodies on the launch is the non-temporal form of the kernel")}
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


def partial_active_leg(mapn, c, n, seconds=0.04):
    """ms per step with num_active = N/2 and 3N/4 (Particles.cpp:391-394's slider; Compute.cpp:1041): the form the library picks, and the
    one-sided step over active x N it ran there until round 4 (MAPN_PARTIAL_FORM=one, a test hook, for this comparison only)."""
    partial_active_leg.set_hooks = os.environ.get("MAPN_TEST_HOOKS") != "1"
    out = []

    def ms(na, k):
        for _ in range(max(4, k // 4)):
            c.Simulate(na, c.GetFenceValue())
        c.WaitForGpu()
        best = float("inf")
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(k):
                c.Simulate(na, c.GetFenceValue())
            c.WaitForGpu()
            best = min(best, (time.perf_counter() - t0) / k * 1e3)
        return best
    c.set_timers(0)
    for num, den in ((1, 2), (3, 4)):
        na = n * num // den // 64 * 64
        k = max(4, min(200, int(seconds / (0.6e-3 * (n / 65536.0) ** 2 * num / den))))
        os.environ.pop("MAPN_PARTIAL_FORM", None)
        t_pick = ms(na, k)
        st = c.kernel_stats()
        form = "split" if st.split_active else ("full symmetric" if st.kernel_name.decode() == "force_sym_kernel" else "one-sided")
        os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_PARTIAL_FORM"] = "one"
        t_one = ms(na, k)
        os.environ.pop("MAPN_PARTIAL_FORM")
        if partial_active_leg.set_hooks:
            os.environ.pop("MAPN_TEST_HOOKS", None)
        out.append({"num_active": na, "form_picked": form, "ms_per_step": round(t_pick, 5), "ms_per_step_one_sided": round(t_one, 5),
                    "speedup_over_one_sided": round(t_one / t_pick, 4), "interactions_per_s": float(na) * n / (t_pick * 1e-3),
                    "note": "ordered pairs the step must account for: num_active x N (the frozen bodies still exert force); bound of the split form over the "
                            "one-sided step at the two kernels' rates (7.1e12 / 4.9e12): 1 / (x / 1.45 + 1 - x), x = num_active / N"})
    return out


partial_active_leg.set_hooks = False


def central_well_leg(mapn, device, bodies, seconds=0.1):
    """The HBM-bound step (MAPN_FORCE_CENTRAL_WELL: nBodyGravityCS.hlsl:86-109 exactly as shipped, 16 + 12 bytes read and 16 + 12 written per
    body) at `bodies` bodies: wall clock over ~`seconds` of back-to-back steps after a warm-up, no timers; GB/s = 56 x bodies / step time,
    priced against the 8 TB/s specification and the 6.29 TB/s a float4 copy reaches on this part (MI355X guide)."""
    import numpy as np
    rng = np.random.default_rng(1)
    pos = np.zeros((bodies, 4), np.float32)
    pos[:, :3] = rng.uniform(-700.0, 700.0, size=(bodies, 3)).astype(np.float32)      # (the two-shell state's value range; the kernel's time does not depend on the data)
    vel = rng.uniform(-15.0, 15.0, size=(bodies, 3)).astype(np.float32)
    with mapn.Compute(bodies, device=device, force_mode=mapn.FORCE_CENTRAL_WELL, flags=mapn.FLAG_NO_INIT) as w:
        w.upload_state(pos, vel)
        del pos, vel
        w.set_timers(0)
        est = HBM_BYTES_PER_BODY * bodies / 6.0e12
        k = max(20, min(4000, int(seconds / est)))
        for _ in range(max(10, k // 4)):
            w.Simulate(bodies, w.GetFenceValue())
        w.WaitForGpu()
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(k):
                w.Simulate(bodies, w.GetFenceValue())
            w.WaitForGpu()
            best = min(best, (time.perf_counter() - t0) / k)
    gbps = HBM_BYTES_PER_BODY * bodies / best / 1e9
    return {"bodies": bodies, "steps_per_region": k, "ms_per_step": round(best * 1e3, 5), "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4),
            "frac_of_6.29": round(gbps / 6290.0, 4), "bytes_per_step": HBM_BYTES_PER_BODY * bodies,
            "form": "non-temporal loads and stores" if HBM_BYTES_PER_BODY * bodies > (320 << 20) else "plain (the step's state fits the 256 MiB Infinity Cache: a cache rate)",
            "timing": "wall clock over the region (best of 3), untimed by the contract"}


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` with no launcher environment: this process becomes the launcher.  It starts N children of
    this same command line -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way
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
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True))
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


if __name__ == "__main__":
    main()
