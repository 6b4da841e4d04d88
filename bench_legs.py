"""bench_legs.py -- what bench.py reports BESIDE the timed region: the CPU baseline, the roofline object of the dominant kernel
(HIP-event launch time, committed PMC traffic), the package-power leg, SURVEY 8(d)'s statistic, and the untimed legs behind the run
(partially active steps, the HBM-bound central-well mode).  Nothing here runs inside the contract's timed region (bench.py: run_steps)."""
from __future__ import annotations

import json
import os
import time

FLOP_PER_PAIR = 20          # op count of nBodyGravityCS.hlsl:44-57, rsqrt = 1 flop
HBM_BYTES_PER_BODY = 56     # 16+12 read, 16+12 written
HERE = os.path.dirname(os.path.abspath(__file__))


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
        # the thread count the port runs FASTEST at on this host (round 6; Oracle.best_threads): it creates and joins its workers every step and
        # the GPU box's container does not have the 256 cores it shows (65 536 bodies: 30 ms per step on 64 threads, 47 on 128, 42 on 256) --
        # the baseline is the best of all / half / a quarter / an eighth of the hardware threads, and `cores` says which
        cores = o.best_threads()
        sim = OracleSim(o, pos, vel, params=prm, threads=cores)
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
    quota = o.cpu_quota_cores()
    return {"value": pairs / t, "unit": "body-pair interactions/s", "cores": cores, "kind": "port",
            "sample": f"oracle (C, fp32, exact HLSL op order, vectorised over i): {what}, {cores} threads, {t:.1f} s",
            "cpu": model, "hardware_threads_shown": o.hardware_threads(), "cpu_quota_cores": quota,
            "note": "`cores` = the threads the port ran on (its fastest count on this host); `cpu_quota_cores` = the CPU time the container is granted, in cores "
                    "(cgroup cpu.max; null: unlimited) -- on the GPU boxes 16 of the 256 hardware threads shown, which is what bounds this figure"}


def kernel_source_sha16():
    import hashlib
    here = HERE
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
    here = HERE
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



def roofline_all_pairs(st, info, clock, power, n, count, world, pairs_per_step):
    """The `roofline` object of the all-pairs force launch: `achieved` = 20 flop x the ordered pairs one launch processes / the kernel's mean
    launch duration from HIP events on the compute stream around sampled launches of the timed region (mapn_kernel_stats)."""
    if not (st.launches and st.avg_seconds > 0):
        return None
    name = st.kernel_name.decode()
    peak = info.peak_fp32_flops / 1e12
    pairs_per_launch = float(count) * float(n)
    ach = FLOP_PER_PAIR * pairs_per_launch / st.avg_seconds / 1e12
    traffic, traffic_src = pmc_traffic(name, n, world)
    held = clock.shader_clock_ghz if clock else None
    sym = name == "force_sym_kernel"
    # flop the kernel EXECUTES per ordered interaction: the one-sided kernels evaluate every ordered
    # pair (20); the symmetric kernel evaluates every unordered pair once and feeds both bodies:
    # 3 sub + 3 fma + rsq + 2 mul + 3 fma + 3 fma = 24 flop per TWO interactions
    executed_per_pair = 12.0 if sym else float(FLOP_PER_PAIR)
    return {"bound": "mfma",
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
                           "samples; `at_cap` (THIS box, this run): the kernel runs at the board's power limit, so the clock it holds -- and with it `frac`, priced at the "
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
            "kernel": name, "launches_timed": int(st.launches),
            "avg_launch_ms": st.avg_seconds * 1e3, "flop_per_pair": FLOP_PER_PAIR,
            "pairs_per_launch": pairs_per_launch,
            "algorithmic_hbm_bytes_per_launch": HBM_BYTES_PER_BODY * count,
            "algorithmic_hbm_GBps": HBM_BYTES_PER_BODY * count / st.avg_seconds / 1e9,
            "note": "peak = CUs x clock x 256 flop/clk (fp32 vector = dense f32 MFMA peak, 157.3 TF)"}


def roofline_central_well(n, steps, elapsed):
    rate = HBM_BYTES_PER_BODY * n * steps / elapsed
    return {"bound": "hbm", "achieved": rate / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": rate / 8e12, "traffic": None,
            "reachable_GBps": 6290.0, "frac_of_reachable": rate / 6.29e12,
            "note": ("central-well step, 56 B per body: `achieved` is wall-clock over the K steps; `peak` the 8.0 TB/s specification, `reachable_GBps` what a "
                     "float4 copy measures on this part (MI355X guide); up to ~4.5 Mi bodies a step's state is resident in the 256 MiB Infinity Cache (a cache "
                     "rate, not an HBM one); from 6 Mi bodies on the launch is the non-temporal form of the kernel")}


def survey_8d(region_ms, steps, pairs_per_step):
    """SURVEY 8(d)'s statistic -- >= 100 timed steps per region, median of five regions -- from the per-step times of five regions."""
    r = sorted(region_ms)
    med = r[len(r) // 2]
    return {"steps": steps, "repeats": len(r), "median_ms_per_step": round(med, 5), "min": round(r[0], 5), "max": round(r[-1], 5),
            "interactions_per_s": pairs_per_step / (med * 1e-3), "meets_survey_8d": steps >= 100 and len(r) >= 5,
            "note": "five regions of `steps` steps each, every one closed like the contract's (barrier + device sync on both sides, MAX over ranks); untimed by "
                    "the contract -- `value` stays the K-step region the driver asked for"}


def step_spread(mapn, c, steps):
    """How the step time is spread over the timed region: the steps that carried HIP events (every timer_interval-th), by quarter of the
    region, and their minimum / median / maximum.  (None, None) when there are none."""
    try:
        idx, step_ms, _ = c.step_samples()
    except mapn.MapnError:
        return None, None
    if not len(idx):
        return None, None
    quarters = []
    for qn in range(4):
        m = (idx >= steps * qn // 4) & (idx < steps * (qn + 1) // 4)
        quarters.append(round(float(step_ms[m].mean()), 5) if m.any() else None)
    return quarters, {"steps_timed": int(len(idx)), "min_ms": round(float(step_ms.min()), 5), "median_ms": round(float(sorted(step_ms)[len(idx) // 2]), 5),
                      "max_ms": round(float(step_ms.max()), 5)}
