"""bench.py's output contract (one JSON line with the agreed keys) and its refusal to run
without a GPU (no CPU fallback for the measured path)."""
import json
import os
import subprocess
import sys

import pytest

import mapn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.timing     # subprocess benches and time windows: ordered behind every parity test (tests/conftest.py)


def test_bench_fails_loudly_without_a_gpu():
    if mapn.compute.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--bodies", "1024"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())


def test_bench_launches_its_own_ranks_and_relays_their_failure_without_a_gpu():
    """`python bench.py --gpus 2` with NO launcher environment must start its two ranks itself (the driver's
    8-GPU command may have exactly this shape).  Without a GPU both ranks fail loudly; the launcher relays the worst exit
    code and no JSON line appears."""
    if mapn.compute.device_count() > 0:
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--bodies", "2048",
                        "--same-device", "--no-survey-leg", "--dist-backend", "gloo", "--gather", "p2p"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "[bench launcher] rank 0 exited with code" in r.stderr and "[bench launcher] rank 1 exited with code" in r.stderr
    assert not any(line.startswith("{") for line in r.stdout.splitlines())


def test_power_leg_reads_the_sensor_between_batches_and_reports_the_settled_half(tmp_path):
    """bench.py's `roofline.power`: the hwmon sensor is found (one candidate: taken; several and no PCI match: none), read between
    batches of steps, and the figures are those of the second half of the samples."""
    sys.path.insert(0, ROOT)
    import bench
    hw = tmp_path / "card3" / "device" / "hwmon" / "hwmon7"
    hw.mkdir(parents=True)
    (hw / "power1_cap").write_text("1400000000\n")
    (hw / "freq1_input").write_text("2200000000\n")
    (hw / "power1_input").write_text("300000000\n")
    (tmp_path / "card4" / "device").mkdir(parents=True)                      # a card without a sensor (the boxes list dozens)
    assert bench.power_sensor(0, root=str(tmp_path)) == str(hw)
    calls = []

    def step_batch(k):                                                      # the "kernel": the sensor climbs to the limit while it runs
        calls.append(k)
        (hw / "power1_input").write_text(str(min(1400, 300 + 200 * len(calls)) * 1000000))
    out = bench.power_leg(step_batch, str(hw), seconds=0.05, batch_ms=1.0, ms_per_step=0.5)
    assert out is not None and calls and all(k == 2 for k in calls) and out["samples"] == len(calls) >= 4
    assert out["cap_w"] == 1400.0 and out["package_w_max"] == 1400.0 and out["package_w"] == 1400.0 and out["at_cap"] is True and out["sensor_sclk_mhz"] == 2200.0
    hw2 = tmp_path / "card5" / "device" / "hwmon" / "hwmon9"               # a second GPU's sensor and no way to tell them apart here: no guess
    hw2.mkdir(parents=True)
    (hw2 / "power1_input").write_text("100000000\n")
    assert bench.power_sensor(0, root=str(tmp_path)) is None
    assert bench.power_leg(lambda k: None, str(tmp_path / "nowhere"), seconds=0.01, batch_ms=1.0, ms_per_step=0.5) is None


def test_committed_pmc_summaries_belong_to_the_current_kernel_sources():
    """bench.py attaches `roofline.traffic` only while the committed PMC summary was measured on the kernel
    sources being run (their sha is stored in it): a kernel edit without a fresh `--pmc` pass must show up
    here, on the CPU, not as a silent `traffic: null` at round end."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_sha", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    sha = bench.kernel_source_sha16()
    for tag, kernel in (("r06_sym", "force_sym_kernel"), ("r06_onesided", "force_sgpr_kernel")):
        d = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json")))
        assert d["_kernel_source_sha16"] == sha, f"profiles/{tag}_pmc_summary.json is stale: re-run tools/evidence.sh pmc"
    traffic, src = bench.pmc_traffic("force_sym_kernel", 65536, 1)
    assert traffic and 3e7 < traffic < 2e8 and src.endswith("r06_sym_pmc_summary.json")     # rows: N^2 / 128 + 16 N x parts bytes, + the positions


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "body-pair interactions/s" and d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "65536 bodies" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 65536.0 ** 2 * 40 / (d["ms_per_step"] * 40e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "TFLOP/s" and 0.3 < rf["frac"] < 1.2
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and abs(rf["peak"] - 157.2864) < 0.5
    assert rf["launches_timed"] >= 4 and "traffic" in rf
    assert rf["bound_actual"] == "valu-fp32"                       # no MFMA is issued; `bound` holds the schema's compute value
    assert 1.5 < rf["held_clock_ghz"] <= 2.45 and rf["frac"] < rf["frac_at_held_clock"] < 1.2
    assert d["config"]["kernel"] == "force_sym_kernel" and d["config"]["launches_per_step"] == 2
    assert 0.45 < rf["frac_executed"] < rf["frac"]                  # 12 of the 20 algorithmic flop per ordered pair are executed
    q = d["config"]["step_ms_by_quarter_of_the_timed_region"]       # the spread of the step time over the timed region, from HIP events
    assert len(q) == 4 and all(x is not None and 0.3 < x < 2.0 for x in q) and d["config"]["step_ms_spread"]["steps_timed"] >= 4
    assert "valu_busy" in rf
    pw = rf["power"]                        # what the package draws under the kernel (hwmon sensor of THIS GPU, read over 2.5 s of steps behind the run); null without a readable sensor
    assert pw is None or (300.0 < pw["package_w"] <= 1.05 * pw["cap_w"] and pw["samples"] >= 4 and 50.0 < rf["picojoule_per_interaction"] < 1000.0)
    x = d["config"]["xcd_aware_parts"]                              # calibrated BY THE LIBRARY at mapn_create (MAPN_FLAG_XCD_CALIBRATE), kept only if an untimed A/B wins
    # (the library keeps its calibrated plan only if it wins an A/B at mapn_create: where it did not, the note says so and the default plan runs)
    assert x["mode"] == "auto" and (x.get("error") or x.get("note") or (len(x["weights"]) == 8 and max(x["weights"]) == 1024 and isinstance(x["used"], bool)))
    assert x.get("error") or x.get("note") or x["source"].startswith("library")      # (the bench's plan is the library's plan)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 1e8 and "sample" in cb
    rep = d["config"]["repeats"]            # five regions of K steps, listed with their median, for every K; SURVEY 8(d)'s count needs K >= 100
    assert len(rep["ms_per_step"]) == 5 and rep["meets_survey_8d"] is False and abs(rep["ms_per_step"][0] - d["ms_per_step"]) < 1e-4
    # ... and SURVEY 8(d)'s statistic itself whatever K is: K = 40 < 100, so five more regions of 100 steps ran behind the timed one
    s8 = d["config"]["survey_8d"]
    assert s8["steps"] == 100 and s8["repeats"] == 5 and s8["meets_survey_8d"] is True and s8["min"] <= s8["median_ms_per_step"] <= s8["max"]
    assert 0.4 < s8["median_ms_per_step"] < 1.0 and s8["max"] / s8["min"] < 1.15 and abs(s8["interactions_per_s"] - 65536.0 ** 2 / (s8["median_ms_per_step"] * 1e-3)) < 1e-3 * s8["interactions_per_s"]
    assert abs(s8["median_ms_per_step"] / d["ms_per_step"] - 1.0) < 0.15      # (the K-step region the driver asked for is no outlier against it)
    assert d["value"] > 3.0e12            # the 40 %-of-peak target is 3.15e12 at kernel level
    # the HBM-bound mode in the same line: CSMain as shipped at 4 Mi bodies (defines.h:45) and at 16 Mi, GB/s = 56 N / t
    cw = d["config"]["central_well"]
    assert [e["bodies"] for e in cw] == [4 * 1024 * 1024, 16 * 1024 * 1024], cw
    for e in cw:
        assert "error" not in e, e
        assert abs(e["GBps"] - 56.0 * e["bodies"] / (e["ms_per_step"] * 1e-3) / 1e9) < 0.01 * e["GBps"]
        assert 3000.0 < e["GBps"] < 9000.0 and abs(e["frac_of_8TBps"] - e["GBps"] / 8000.0) < 1e-3 and abs(e["frac_of_6.29"] - e["GBps"] / 6290.0) < 1e-3, e
    assert "non-temporal" in cw[1]["form"] and "Infinity Cache" in cw[0]["form"]
    pa = d["config"]["partial_active"]      # partially active steps in the same line: the split form against the one-sided step round 4 ran
    assert [e["num_active"] for e in pa] == [32768, 49152] and all(e["form_picked"] == "split" for e in pa), pa
    assert pa[0]["speedup_over_one_sided"] > 1.06 and pa[1]["speedup_over_one_sided"] > 1.15, pa     # (measured 1.15 - 1.16 and 1.25 - 1.27; the bound at N/2 is 1.18)
    rp = d["config"]["replay"]              # 10 steps from the seeded state under the timed plan: the pair of checksums a re-run with the same weights reproduces
    assert rp["steps"] == 10 and len(rp["checksums"]) == 2 and all(isinstance(x, int) and x > 0 for x in rp["checksums"])
    assert d["config"]["closing_collective_us"] is None     # (N = 1: no collective closes the region)


@pytest.mark.gpu
def test_bench_replays_a_plan_from_given_xcd_weights():
    """the headline plan's die weights differ from box to box and run to run, i.e. so does its summation order.  With
    `--xcd-weights w0,...,w7` (-> mapn_set_sym_xcd_weights, no calibration, no A/B) a run is replayable: two runs with the same weights
    print identical `config.replay.checksums`, another weighting prints others."""
    def run(w):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-power-leg",
                            "--no-central-well-leg", "--no-partial-leg", "--prewarm-ms", "50"] + (["--xcd-weights", w] if w else ["--xcd", "off"]), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    w = "1024,900,1000,950,1024,880,990,1010"
    # (another MULTISET of speeds: a permutation of the same eight values gives the same class-aware plan -- and the same bits --
    #  because the plan sizes a class's parts by its dies' speeds in descending order, whichever dispatch slots hold them)
    a, b, other, plain = run(w), run(w), run("1024,1000,960,1000,990,1000,940,1000"), run("")
    for d in (a, b, other):
        x = d["config"]["xcd_aware_parts"]
        assert x["mode"] == "given" and x["used"] is True and x["form"] == "class-aware" and d["config"]["symmetric_plan"]["xcd_mode"].startswith("class-aware")
    assert a["config"]["xcd_aware_parts"]["weights"] == [int(v) for v in w.split(",")]
    assert a["config"]["replay"]["checksums"] == b["config"]["replay"]["checksums"]
    assert a["config"]["replay"]["checksums"] != other["config"]["replay"]["checksums"]
    assert plain["config"]["xcd_aware_parts"]["used"] is False and plain["config"]["replay"]["checksums"] != a["config"]["replay"]["checksums"]
    assert "central_well" in plain["config"] and plain["config"]["central_well"] is None      # (--no-central-well-leg)


@pytest.mark.gpu
def test_bench_reports_a_median_of_five_repeats_from_100_steps_on():
    """>= 100 timed steps and a median of five repeats.  `value` / `ms_per_step` stay the contract's ONE timed region
    of exactly K steps; four more regions of K steps follow it and the five are listed with their median."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    rep = d["config"]["repeats"]
    assert len(rep["ms_per_step"]) == 5 and rep["meets_survey_8d"] is True and abs(rep["ms_per_step"][0] - d["ms_per_step"]) < 1e-4
    assert sorted(rep["ms_per_step"])[2] == rep["median_ms_per_step"] and 0.4 < rep["median_ms_per_step"] < 1.0
    assert max(rep["ms_per_step"]) / min(rep["ms_per_step"]) < 1.15, rep      # one box, one clock state: the regions agree (2 - 3 % typically)
    s8 = d["config"]["survey_8d"]           # K >= 100: those five regions ARE the statistic
    assert s8["steps"] == 100 and s8["repeats"] == 5 and s8["meets_survey_8d"] is True and s8["median_ms_per_step"] == rep["median_ms_per_step"]


@pytest.mark.gpu
def test_bench_two_ranks_on_two_real_gpus_when_the_box_has_them():
    """every multi-process test shares ONE device, so the cross-GPU forms (hipIpc over xGMI, RCCL with N > 1 ranks) never run
    in this suite -- unless the box has a second GPU.  Then: the default `--gather auto` trial with one rank per GPU, launched by
    bench.py itself, replicas bit-identical, the symmetric forms verified against the one-sided step."""
    if mapn.compute.device_count() < 2:
        pytest.skip("one GPU on this box: the cross-GPU exchange cannot run here (the driver's 8-GPU run is its first execution)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "5"],
                       capture_output=True, text=True, timeout=1200, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # a broken hop is a RED test with bench's own words in the message (the file is ordered behind every parity test, so
    # `pytest -x` hides nothing by stopping here)
    assert r.returncode == 0 and len(lines) == 1, "bench.py --gpus 2 on two real GPUs did not produce its line: " + r.stdout[-1000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    print(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["valid"] is True and d["config"]["replicas_bit_identical_after_run"] is True
    assert len(d["config"]["exchange_trial_us_per_step"]) >= 2, d["config"]


@pytest.mark.gpu
def test_bench_one_sided_kernel_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--kernel", "sgpr"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["kernel"] == "force_sgpr_kernel" and d["config"]["launches_per_step"] == 1 and d["config"]["fused_integrator"] is True
    assert 0.5 < d["roofline"]["frac"] < 0.7 and d["value"] > 4.0e12


@pytest.mark.gpu
def test_bench_one_rank_over_the_rccl_backend_and_the_in_library_communicator():
    """What a 1-GPU box can run of the RCCL path of bench.py: one rank under torch.distributed.run with --force-comm -- the "nccl"
    process group bound to the device, stdout handed to stderr while the collective libraries set up (their banners must not reach
    the ONE JSON line), the unique id broadcast and mapn_comm_init, the all-gather exchange behind every step, the barrier-bracketed
    timing through RCCL collectives.  (Every N > 1 test here shares one device and has to use gloo: RCCL refuses that.)"""
    port = 29900 + (os.getpid() % 90)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "30", "--warmup", "3", "--force-comm", "--no-cpu-baseline", "--prewarm-ms", "50"],
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["transport"] == "rccl" and d["config"]["exchange"] == "allgather" and d["config"]["valid"] is True
    assert 0.0 < d["ms_per_step_before_closing_barrier"] <= d["ms_per_step"] and d["value"] > 3.0e12


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_with_the_direct_exchange():
    """The N > 1 flow of bench.py (rendezvous, sharded contexts, exchange set-up, barrier-bracketed
    timing, MAX over ranks, replica consistency check) with two ranks sharing device 0: gloo for
    the rendezvous (RCCL refuses two ranks on one device) and the in-library peer-to-peer
    exchange for the data path."""
    port = 29700 + (os.getpid() % 200)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "30", "--warmup", "3", "--gather", "p2p", "--dist-backend", "gloo",
                        "--same-device", "--prewarm-ms", "20"], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["exchange"] == "p2p"
    s8 = d["config"]["survey_8d"]           # (this one N > 1 test keeps the five 100-step regions: they run through the same closing collective, MAX over ranks)
    assert s8["steps"] == 100 and s8["repeats"] == 5 and s8["meets_survey_8d"] is True and s8["min"] <= s8["median_ms_per_step"] <= s8["max"]
    assert d["config"]["replicas_bit_identical_after_run"] is True
    assert d["config"]["parallelism"] == "bodies sharded x2" and "cpu_baseline" not in d
    # the contract's figure includes the closing collective; the time until every rank's own device was idle (MAX over ranks) is listed beside it
    assert 0.0 < d["ms_per_step_before_closing_barrier"] <= d["ms_per_step"]
    cc = d["config"]["closing_collective_us"]          # what the ONE closing collective (barrier + verdict) adds to the K steps, MAX over ranks
    assert cc is not None and abs(cc - (d["ms_per_step"] - d["ms_per_step_before_closing_barrier"]) * 30 * 1e3) < 1.0 and 0.0 <= cc < 20000.0


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO torch.distributed.run around it -- the process becomes the
    launcher (before any GPU call), starts one child per rank with RANK / WORLD_SIZE / MASTER_* set, relays rank 0's single
    JSON line and the worst exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "3", "--gather", "p2p",
                        "--dist-backend", "gloo", "--same-device", "--no-survey-leg", "--prewarm-ms", "20"], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["config"]["exchange"] == "p2p"
    assert d["config"]["replicas_bit_identical_after_run"] is True and d["config"]["parallelism"] == "bodies sharded x2"


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_with_the_in_kernel_exchange():
    """The same N > 1 flow with gather algorithm 3 ("flow": the exchange overlapped inside the force
    launch).  Two ranks time-slice ONE device here, so the job is kept small enough for both ranks'
    launches to be resident together (a rank's waiting workgroups must never starve the peer it
    waits for -- on a node every rank has its own GPU)."""
    port = 29900 + (os.getpid() % 90)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "30", "--warmup", "3", "--gather", "flow", "--dist-backend", "gloo",
                        "--same-device", "--no-survey-leg", "--prewarm-ms", "20", "--bodies", "8192"], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["exchange"] == "p2p+inkernel"
    assert d["config"]["replicas_bit_identical_after_run"] is True and d["config"]["p2p_failure"] is None


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_with_the_sharded_symmetric_step():
    """The N > 1 flow with gather algorithm 4: the symmetric step sharded over the ranks, verified inside
    bench.py against the one-sided sharded step before it is used."""
    port = 29990 + (os.getpid() % 90)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "30", "--warmup", "3", "--gather", "sym", "--dist-backend", "gloo",
                        "--same-device", "--no-survey-leg", "--prewarm-ms", "20", "--bodies", "16384"], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["exchange"] == "p2p+symmetric" and d["config"]["kernel"] == "force_sym_kernel"
    assert d["config"]["replicas_bit_identical_after_run"] is True and d["config"]["p2p_failure"] is None
    assert d["config"]["launches_per_step"] == 2 and set(d["config"]["exchange_trial_us_per_step"]) == {"p2p", "p2p+symmetric"}
    assert d["config"]["sharded_symmetric_deviation_after_run"] < 1e-5 and d["config"]["valid"] is True


@pytest.mark.gpu
def test_bench_falls_back_once_when_a_verified_form_fails_in_the_run_itself():
    """The 8-GPU run is the first real execution of the cross-GPU forms: a form can pass its trial and still fail later.  Here rank 1
    corrupts ONE pushed position in the timed region (test hook); rank 0's library reports it (MAPN_ERR_COMM), bench.py -- whose
    closing collective carries the verdict to every rank instead of leaving them in a barrier -- rebuilds the contexts, falls back
    to the next verified form, times the K steps again and says so in its line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "3", "--gather", "sympush",
                        "--dist-backend", "gloo", "--same-device", "--no-survey-leg", "--prewarm-ms", "20", "--bodies", "16384", "--test-inject-push-failure",
                        "--p2p-timeout-ms", "3000"], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert "p2p+symmetric+push" in cfg["fallback_after_failure"] and "PUSHED" in cfg["fallback_after_failure"], cfg["fallback_after_failure"]
    assert cfg["exchange"] == "p2p" and d["n_gpus"] == 2 and d["steps"] == 30          # (the other verified form of this trial)
    assert cfg["replicas_bit_identical_after_run"] is True and cfg["valid"] is True


@pytest.mark.gpu
def test_a_form_that_fails_on_one_rank_during_the_trial_is_dropped_by_all_of_them():
    """The exchange TRIAL is where a cross-GPU form meets real links for the first time.  Here rank 1 corrupts ONE pushed position while
    'p2p+symmetric+push' is being tried (test hook): rank 0's library reports it, rank 1's own steps may or may not -- every stage of the
    trial ends in one collective that carries the failure to all ranks (Job.together), so both drop that form, replace their contexts, and
    the run goes on with what else was verified: nobody is left in a barrier the other rank skipped."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "3", "--gather", "p2pall",
                        "--dist-backend", "gloo", "--same-device", "--no-survey-leg", "--prewarm-ms", "20", "--bodies", "16384", "--test-inject-trial-failure",
                        "--p2p-timeout-ms", "3000", "--xcd", "off"], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    cfg = json.loads(lines[0])["config"]
    assert "p2p+symmetric+push" not in cfg["exchange_trial_us_per_step"] and "p2p+symmetric" in cfg["exchange_trial_us_per_step"], cfg
    assert "p2p+symmetric+push" in cfg["p2p_failure"], cfg["p2p_failure"]
    assert cfg["exchange"] in cfg["exchange_trial_us_per_step"] and cfg["fallback_after_failure"] is None
    assert cfg["replicas_bit_identical_after_run"] is True and cfg["valid"] is True


@pytest.mark.gpu
@pytest.mark.slow
def test_bench_eight_ranks_one_gpu_trial_over_every_peer_to_peer_form():
    """The exchange TRIAL of bench.py as the driver's 8-GPU run goes through it, with eight ranks sharing device 0 (gloo
    rendezvous; RCCL refuses several ranks on one device, so the trial runs over the peer-to-peer forms: pull, in-kernel
    "flow", sharded symmetric with pulled and with pushed positions): every form is set up, stepped, checked for bit-identical
    replicas and -- the symmetric ones -- against the one-sided sharded step; all ranks take the same decision."""
    port = 29400 + (os.getpid() % 150)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "8", "--steps", "24", "--warmup", "3", "--gather", "p2pall", "--dist-backend", "gloo",
                        "--same-device", "--no-survey-leg", "--prewarm-ms", "10", "--bodies", "16384", "--p2p-timeout-ms", "10000", "--xcd", "off"], capture_output=True, text=True, timeout=1500)   # (--xcd off: eight creation-time calibrations on one shared device are 10 s of this test and not its subject)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 8 and cfg["parallelism"] == "bodies sharded x8"
    assert set(cfg["exchange_trial_us_per_step"]) == {"p2p", "p2p+inkernel", "p2p+symmetric", "p2p+symmetric+push"}, (cfg, r.stderr[-2000:])
    assert cfg["exchange"] in cfg["exchange_trial_us_per_step"] and cfg["p2p_failure"] is None
    assert cfg["replicas_bit_identical_after_run"] is True and cfg["valid"] is True


def test_the_exchange_trial_tries_the_overlap_structures_last_and_only_by_name():
    """north_star's overlap structure (own-segment launch beside the previous all-gather) measures SLOWER than its plain form
    in loopback (152 - 158 against 115 - 121 us per step: profiles/r06_shard_step_timeline.txt), so the trial of `bench.py --gpus N` holds it only
    when asked for by name, and then BEHIND every other candidate.  The candidate list is host logic: checked here without a GPU, with the
    peer-to-peer set-up succeeding on every rank."""
    import types
    sys.path.insert(0, ROOT)
    from bench_ranks import Job

    def names(gather, overlap, world=8, n=65536):
        a = types.SimpleNamespace(bodies=n, dist_backend="gloo", gather=gather, overlap=overlap, p2p_timeout_ms=1000, seed=1, plan="")
        fake = types.SimpleNamespace(FORCE_ALL_PAIRS=0, FORCE_CENTRAL_WELL=1)
        job = Job(a, fake, None, None, 0, world, 0, 0, 0, 0)
        job.c = types.SimpleNamespace(p2p_setup_torch=lambda: None, set_timeouts=lambda **k: None)
        job.all_reduce = lambda value, op="MAX", dtype=None: value
        return [c[0] for c in job.build_candidates()]
    assert names("auto", False) == ["allgather", "sendrecv", "rccl+symmetric", "p2p", "p2p+symmetric", "p2p+symmetric+push"]
    assert names("allgather", False) == ["allgather", "allgather+overlap"]
    assert names("sendrecv", False) == ["sendrecv", "sendrecv+overlap"]
    auto_overlap = names("auto", True)                     # --overlap: the overlap structures instead of their plain forms -- still behind everything else
    assert auto_overlap[-2:] == ["allgather+overlap", "sendrecv+overlap"] and "allgather" not in auto_overlap and "p2p+symmetric+push" in auto_overlap
    assert names("p2pall", False) == ["p2p", "p2p+inkernel", "p2p+symmetric", "p2p+symmetric+push"]
    assert names("auto", False, world=8, n=65536 + 8 * 64) == ["allgather", "sendrecv", "p2p"]      # a slice that is not whole blocks: no symmetric forms


def test_survey_8d_statistic_is_the_median_of_five_regions_of_at_least_100_steps():
    """>= 100 timed steps per region, median of five repeats (bench_legs.survey_8d: what `config.survey_8d` carries whatever K is)."""
    sys.path.insert(0, ROOT)
    from bench_legs import survey_8d
    s = survey_8d([0.60, 0.59, 0.61, 0.595, 0.592], 100, 65536.0 ** 2)
    assert s["median_ms_per_step"] == 0.595 and s["min"] == 0.59 and s["max"] == 0.61 and s["repeats"] == 5 and s["meets_survey_8d"] is True
    assert abs(s["interactions_per_s"] - 65536.0 ** 2 / 0.595e-3) < 1.0
    assert survey_8d([0.6] * 5, 40, 1.0)["meets_survey_8d"] is False and survey_8d([0.6] * 3, 100, 1.0)["meets_survey_8d"] is False


class _FakeMapnError(RuntimeError):
    pass


def _trial_worker(rank, world, port, out_dir, fail_where):
    """One rank of bench_ranks.Job.exchange_trial over gloo with a FAKE library: the algorithm under test fails on rank 1 only."""
    import types
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from bench_ranks import Job
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 16384
    made = []

    class FakeCompute:
        def __init__(self):
            self.algo, self.steps, self.failed, self.weights = 0, 0, False, [1.0] * 8
            made.append(self)
        def set_gather_algorithm(self, algo): self.algo = algo
        def set_shard_overlap(self, on): pass
        def GetFenceValue(self): return self.steps
        def Simulate(self, n_active, fence):
            self.steps += 1
            if fail_where == "ab" and rank == 1 and self.weights is None and self.steps > 400:
                raise _FakeMapnError("mapn status 7: a reaction row never validated")
            if self.algo == 5 and rank == 1 and fail_where == "simulate":
                self.failed = True
                raise _FakeMapnError("mapn status 7: a PUSHED position did not match its checksum")
        def WaitForGpu(self):
            if self.algo == 5 and rank == 1 and fail_where == "wait":
                self.failed = True
                raise _FakeMapnError("mapn status 7: a device-side wait gave up")
        def p2p_status(self): return 1 if self.failed else 0
        def replica_checksum(self): return (1, 2)
        def upload_state(self, pos, vel): pass
        def download_state(self): return np.zeros((n, 4), np.float32), None
        def close(self): pass
        def p2p_setup_torch(self): dist.barrier()            # (the real one is a collective too)
        def comm_init_torch(self): dist.barrier()
        def set_timeouts(self, **k): pass
        def set_timers(self, k): pass
        def sym_plan(self): return types.SimpleNamespace(xcd_mode=2 if self.weights else 0, xcd_weight=self.weights or [1.0] * 8)
        def set_sym_xcd_weights(self, w): self.weights = w

    fake = types.SimpleNamespace(FORCE_ALL_PAIRS=0, MapnError=_FakeMapnError, Compute=lambda *a, **k: FakeCompute(),
                                 generate_initial_state=lambda n, seed: (None, None))
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(synchronize=lambda: None), tensor=torch.tensor, float64=torch.float64)
    a = types.SimpleNamespace(bodies=n, dist_backend="gloo", gather="p2pall", overlap=False, p2p_timeout_ms=1000, seed=1, plan="",
                              trial_seconds=60.0, test_inject_trial_failure=False, xcd="auto")
    job = Job(a, fake, fake_torch, dist, rank, world, 0, 0, 0, 0)
    job.create()
    job.exchange_trial()
    xcd = {}
    if fail_where == "ab":
        job.sharded_xcd_ab(xcd)
    json.dump({"trial": sorted(job.trial), "chosen": job.gather_algo, "p2p_failure": job.p2p_failure, "contexts": len(made), "xcd": xcd},
              open(os.path.join(out_dir, f"trial_rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_where", ["simulate", "wait", "ab"])
def test_the_exchange_trial_stays_in_lockstep_when_one_rank_alone_fails(tmp_path, fail_where):
    """bench_ranks.Job.exchange_trial over a real two-rank gloo group and a fake library (no GPU): 'p2p+symmetric+push' fails on rank 1 ONLY -- in
    `Simulate` or in the wait behind the steps -- while rank 0's library reports nothing.  Every stage of the trial ends in one collective that
    carries the failure (Job.together), so both ranks drop the form, both replace their context, both end with the same table and choice; before
    round 6's last change rank 0 sat in a barrier rank 1 had skipped."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + {"simulate": 0, "wait": 7, "ab": 14}[fail_where]
    mp.spawn(_trial_worker, args=(2, port, str(tmp_path), fail_where), nprocs=2, join=True)
    got = [json.load(open(os.path.join(str(tmp_path), f"trial_rank{r}.json"))) for r in range(2)]
    assert got[0] == got[1], got
    if fail_where == "ab":                 # the untimed A/B of the die weights behind the trial: rank 1 fails in the unweighted bursts
        assert len(got[0]["trial"]) == 4 and "rank 1: mapn status 7" in got[0]["xcd"]["error"], got[0]
        return
    assert got[0]["trial"] == ["p2p", "p2p+inkernel", "p2p+symmetric"] and got[0]["chosen"] in got[0]["trial"]
    assert "rank 1: mapn status 7" in got[0]["p2p_failure"] and "p2p+symmetric+push" in got[0]["p2p_failure"] and got[0]["contexts"] == 2


def _setup_worker(rank, world, port, out_dir):
    """mapn.compute.Compute.p2p_setup_torch / comm_init_torch over gloo with the library calls stubbed: rank 1 cannot export, rank 0 cannot make the id."""
    import types
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "multi-adapter-particles_amd"))
    sys.path.insert(0, ROOT)
    from mapn.compute import Compute, MapnError
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = {}

    def export():
        if rank == 1:
            raise MapnError(-2, "hipIpcGetMemHandle: invalid argument")
        return b"x" * 8

    def make_id():
        raise MapnError(-4, "librccl.so not found")
    stub = types.SimpleNamespace(p2p_export=export, p2p_import=lambda blobs: got.setdefault("imported", True),
                                 comm_unique_id=make_id, comm_init=lambda uid: got.setdefault("joined", True))
    for name, fn in (("p2p", Compute.p2p_setup_torch), ("comm", Compute.comm_init_torch)):
        try:
            fn(stub)
            got[name] = "ok"
        except MapnError as e:
            got[name] = str(e)
    json.dump(got, open(os.path.join(out_dir, f"setup_rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_that_cannot_set_up_its_transport_still_takes_part_in_the_rendezvous(tmp_path):
    """The two set-up rendezvous of the N > 1 run (hipIpc blobs: all_gather_object; RCCL unique id: broadcast) with ONE rank's library call
    failing in front of the collective: that rank takes part in it anyway and every rank gets a MapnError -- nobody maps half a job, nobody
    waits for a blob that never comes (a two-rank gloo group, the library calls stubbed: no GPU)."""
    import torch.multiprocessing as mp
    mp.spawn(_setup_worker, args=(2, 29500 + (os.getpid() % 2000) + 21, str(tmp_path)), nprocs=2, join=True)
    got = [json.load(open(os.path.join(str(tmp_path), f"setup_rank{r}.json"))) for r in range(2)]
    assert "hipIpcGetMemHandle" in got[1]["p2p"] and "rank(s) [1] could not export" in got[0]["p2p"], got
    assert "librccl.so not found" in got[0]["comm"] and "rank 0 could not create the RCCL unique id" in got[1]["comm"], got
    assert not any("imported" in g or "joined" in g for g in got)
