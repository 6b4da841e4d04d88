"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/mapn.h declares, and its host-only pieces (config defaults, initial-state generator,
error path without a device) behave.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mapn
from mapn import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("mapn.h", "mapn_tuning.h")     # the drop-in boundary (the Compute surface + the sharded mode) / tuning and introspection, versioned apart


def _declared(headers=HEADERS):
    names = set()
    for h in headers:
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names |= set(re.findall(r"\b(mapn_[a-z_0-9]+)\s*\(", hdr))
    return sorted(names)


def test_library_exports_every_declared_symbol(lib):
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"libmapn.so does not export {n}"
    assert sorted(_lib.SIGNATURES) == names, "python binding and header disagree"


def test_library_exports_nothing_but_the_declared_c_abi():
    """The drop-in boundary is a C ABI: libmapn.so is built with -fvisibility=hidden and every FUNCTION it exports is one that
    include/mapn.h declares (the device-kernel handle objects hipcc emits are data symbols, not entry points)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", mapn.library_path()], capture_output=True, text=True, check=True).stdout
    funcs = sorted(ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in ("T", "t"))
    assert funcs == _declared(), sorted(set(funcs) ^ set(_declared()))


def test_the_two_headers_split_the_boundary_from_the_tuning_surface():
    """include/mapn.h is what replaces a member of the reference's `class Compute` (+ the state hand-off and the sharded
    mode); plans, calibration, statistics and diagnostics live in include/mapn_tuning.h with their own version.  No name is declared
    twice; the shim a Particles.cpp-style caller links (compat/Compute.hpp) needs mapn.h alone."""
    core, tuning = set(_declared(("mapn.h",))), set(_declared(("mapn_tuning.h",)))
    assert not (core & tuning)
    for n in ("mapn_create", "mapn_create_from", "mapn_destroy", "mapn_simulate", "mapn_fence_value", "mapn_wait_idle", "mapn_get_shared_handles",
              "mapn_adopt_position_buffers", "mapn_reset_from_async", "mapn_last_step_seconds", "mapn_comm_init", "mapn_set_gather_algorithm"):
        assert n in core, n
    for n in ("mapn_get_sym_plan", "mapn_set_sym_plan", "mapn_sym_plan_describe", "mapn_get_split_plan", "mapn_calibrate_sym_xcds", "mapn_set_sym_xcd_weights",
              "mapn_get_kernel_stats", "mapn_set_force_plan", "mapn_measure_clock", "mapn_set_timers", "mapn_shard_describe", "mapn_shard_split_describe", "mapn_tuning_abi_version"):
        assert n in tuning, n
    assert len(tuning) == 16 and len(core) == 52, (len(tuning), len(core))
    shim = open(os.path.join(ROOT, "multi-adapter-particles_amd", "compat", "Compute.hpp")).read()
    assert '#include "mapn.h"' in shim and "mapn_tuning.h" not in shim
    used = set(re.findall(r"\b(mapn_[a-z_0-9]+)\s*\(", shim))
    assert used and used <= core, used - core


def test_abi_version_and_timer_name(lib):
    assert lib.mapn_abi_version() == 4 and lib.mapn_tuning_abi_version() == 2
    hdr = open(os.path.join(ROOT, "include", "mapn.h")).read() + open(os.path.join(ROOT, "include", "mapn_tuning.h")).read()
    assert "#define MAPN_ABI_VERSION 4" in hdr and "#define MAPN_TUNING_ABI_VERSION 2" in hdr
    assert lib.mapn_timer_name() == b"simulate ms"          # Compute.cpp:446


def test_config_defaults_are_the_reference_constants(lib):
    cfg = mapn.Config()
    assert lib.mapn_config_default(C.byref(cfg)) == 0
    assert cfg.struct_size == C.sizeof(mapn.Config) == 80
    assert cfg.num_particles == 4 * 1024 * 1024             # defines.h:45
    assert (cfg.mass, cfg.softening_squared) == (70000.0, 25.0)   # hlsl:37-38
    assert np.float32(cfg.dt).view(np.uint32) == 0x3DCCCCCD and cfg.damping == 1.0   # Compute.cpp:545-546
    assert (cfg.spread, cfg.initial_speed) == (400.0, 15.0) # defines.h:42,39
    assert (cfg.force_mode, cfg.world_size, cfg.rank, cfg.seed) == (mapn.FORCE_ALL_PAIRS, 1, 0, 1)


@pytest.mark.parametrize("n,seed", [(256, 1), (4096, 1), (1000, 3), (70001, 11)])
def test_product_generator_matches_oracle_bit_exact(oracle, n, seed):
    pos, vel = mapn.generate_initial_state(n, seed=seed)
    op, ov = oracle.initial_state(n, seed=seed)
    np.testing.assert_array_equal(pos, op)
    np.testing.assert_array_equal(vel, ov)


@pytest.mark.parametrize("variant", [mapn.INIT_LCG, mapn.INIT_SSE, mapn.INIT_MT])
def test_all_three_generator_variants_match_oracle_bit_exact(oracle, variant):
    """The three LoadParticles #if variants (Compute.cpp:581-583) as selectable generators."""
    for n, seed in [(512, 1), (3001, 9)]:
        pos, vel = mapn.generate_initial_state(n, seed=seed, variant=variant)
        op, ov = oracle.initial_state(n, seed=seed, variant=variant)
        np.testing.assert_array_equal(pos, op)
        np.testing.assert_array_equal(vel, ov)
        half = n // 2
        cx = np.where(np.arange(n) < half, 300.0, -300.0)[: 2 * half]
        rel = pos[: 2 * half, :3].astype(np.float64) - np.stack([cx, 0 * cx, 0 * cx], 1)
        np.testing.assert_allclose(np.linalg.norm(rel, axis=1), 400.0, atol=2e-4)
    a, _ = mapn.generate_initial_state(512, seed=1, variant=variant)
    b, _ = mapn.generate_initial_state(512, seed=1, variant=(variant + 1) % 3)
    assert not np.array_equal(a, b)
    with pytest.raises(mapn.MapnError):
        mapn.generate_initial_state(64, variant=7)


def test_mt19937_restatement_matches_libstdcxx(oracle, tmp_path):
    """USE_ORIG's random source: the oracle's MT19937 + uniform mapping against the real
    std::mt19937 / std::uniform_real_distribution<float>(-1,1) of this toolchain."""
    import subprocess
    exe = str(tmp_path / "mt_check")
    subprocess.run(["g++", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "mt_check.cpp")], check=True)
    for seed in (1, 5489, 0xDEADBEEF):
        out = subprocess.run([exe, str(seed), "2000"], capture_output=True, text=True, check=True).stdout.split()
        ref = np.array([int(x, 16) for x in out], np.uint32).view(np.float32)
        np.testing.assert_array_equal(oracle.mt_uniform(seed, 2000), ref)


def test_generator_checksums(golden_dir):
    g = np.load(os.path.join(golden_dir, "init_checksums.npz"))
    for n in (1000, 65536):
        pos, vel = mapn.generate_initial_state(n, seed=1)
        sums = [np.frombuffer(pos.tobytes(), np.uint32).sum(dtype=np.uint64), np.frombuffer(vel.tobytes(), np.uint32).sum(dtype=np.uint64)]
        assert sums == g[str(n)].tolist()


def test_bad_arguments_return_status_not_crash(lib):
    assert lib.mapn_config_default(None) == -1
    cfg = mapn.Config(); lib.mapn_config_default(C.byref(cfg))
    ctx = C.c_void_p()
    cfg.struct_size = 12
    assert lib.mapn_create(C.byref(cfg), C.byref(ctx)) == -1 and b"struct_size" in lib.mapn_last_error()
    lib.mapn_config_default(C.byref(cfg)); cfg.num_particles = 100; cfg.world_size = 3
    assert lib.mapn_create(C.byref(cfg), C.byref(ctx)) == -1 and b"divide" in lib.mapn_last_error()
    lib.mapn_config_default(C.byref(cfg)); cfg.num_particles = 64; cfg.kernel = 9
    assert lib.mapn_create(C.byref(cfg), C.byref(ctx)) == -1 and b"kernel 9" in lib.mapn_last_error()
    assert lib.mapn_create(C.byref(cfg), None) == -1 and b"out_ctx" in lib.mapn_last_error()          # no null dereference
    assert lib.mapn_create_from(C.byref(cfg), None, None) == -1
    assert lib.mapn_simulate(None, 1, 0) == -1
    assert lib.mapn_destroy(None) == 0


def test_no_device_fails_loudly_no_cpu_fallback():
    if mapn.compute.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mapn.MapnError) as e:
        mapn.Compute(4096)
    assert e.value.status == -3 and "no CPU fallback" in str(e.value)


def test_product_does_not_link_or_reference_the_oracle():
    """The shipped library and package must not route through oracle/ (or any CPU step)."""
    so = open(mapn.library_path(), "rb").read()
    assert b"mapn_oracle" not in so
    for sub in ("multi-adapter-particles_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".sh")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in text and "from oracle" not in text and "mapn_oracle" not in text, f
    assert "oracle" not in open(os.path.join(ROOT, "mapn.py")).read()
    # the bench may touch the oracle only inside cpu_baseline() (bench_legs.py since round 6; bench.py and bench_ranks.py not at all)
    for f in ("bench.py", "bench_ranks.py"):
        text = open(os.path.join(ROOT, f)).read()
        assert "from oracle" not in text and "import oracle" not in text, f
    legs = open(os.path.join(ROOT, "bench_legs.py")).read()
    head, rest = legs.split("def cpu_baseline", 1)
    body, tail = rest.split("\ndef ", 1)
    assert "from oracle" in body
    assert "from oracle" not in head and "import oracle" not in head
    assert "from oracle" not in tail and "import oracle" not in tail


def test_bench_py_is_the_timed_region_and_the_line_and_stays_readable():
    """bench.py (the file the driver runs and hashes) holds the arguments, the timed region and the JSON line -- at most 500
    lines; the legs beside the region live in bench_legs.py, the launcher / exchange trial / fallback in bench_ranks.py."""
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert len(bench.splitlines()) <= 500, len(bench.splitlines())
    assert "def run_steps(" in bench and "def parse(" in bench and "json.dumps(out)" in bench
    for name, home in (("def launch_ranks(", "bench_ranks.py"), ("def exchange_trial(", "bench_ranks.py"), ("def fall_back(", "bench_ranks.py"),
                       ("def cpu_baseline(", "bench_legs.py"), ("def power_leg(", "bench_legs.py"), ("def central_well_leg(", "bench_legs.py"),
                       ("def partial_active_leg(", "bench_legs.py"), ("def roofline_all_pairs(", "bench_legs.py"), ("def survey_8d(", "bench_legs.py")):
        assert name in open(os.path.join(ROOT, home)).read() and name not in bench, name


def test_force_kernel_keeps_its_scalar_loads(tmp_path):
    """Guard against a silent 27 % regression (round 2): hipcc turns the wave-uniform j-loads of
    force_sgpr_kernel into s_load only while it can prove nothing in the kernel clobbers that memory
    before the load.  A builtin s_memtime, an asm "memory" clobber or an atomic on a possibly-aliasing
    pointer in front of the loop makes every one of them a uniform global_load_dwordx3 -- same
    results, 1.124 instead of 0.888 ms per 65 536-body step.  Disassemble and count."""
    import subprocess
    src = os.path.join(ROOT, "multi-adapter-particles_amd", "csrc", "mapn_kernels.hip")
    out = str(tmp_path / "k.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
    text = open(out).read()
    bodies = re.findall(r"^(_ZN4mapn17force_sgpr_kernelILi(\d)ELi(\d+)ELi(\d)EEEv\w*):(.*?)s_endpgm", text, re.S | re.M)
    assert len(bodies) >= 40
    for name, k2, waves, epi, body in bodies:
        wide = sum(4 * int(w_) for w_ in re.findall(r"s_load_dwordx(4|8|16)\b", body))     # bytes of wide scalar loads
        vec = len(re.findall(r"global_load_dwordx3\b", body))
        assert wide >= 128, f"{name}: only {wide} bytes of scalar j-loads (8 bodies x 16 B per unrolled iteration expected)"
        # load_bodies: 2 * K2 position loads; the integrator: position + velocity of its bodies
        assert vec <= 2 * int(k2) + 2, f"{name}: {vec} vector dwordx3 loads -- the j-loads were de-scalarised"
    default = [b for n_, k2, w, e, b in bodies if (k2, w, e) == ("1", "8", "2")][0]
    assert len(re.findall(r"v_pk_fma_f32", default)) == 56 and "scratch_" not in default


def test_graft_entry_build_succeeds_on_the_cpu():
    """The driver's "does it build" check (`__graft_entry__.build()`: hipcc cross-compiles without a GPU) -- including the version
    assertions it makes about the two headers (an ABI bump that forgot them would fail the round's build check, not a test)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); print('built')"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "built" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
