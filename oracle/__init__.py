"""CPU oracle for the n-body step.  TEST INFRASTRUCTURE ONLY -- the checker, never the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Parity is UNPINNED by the reference (it has no tests or golden vectors, and cannot
be built here); see ``mapn_oracle.c`` for the line-by-line citations this restatement follows.

``Oracle`` wraps ``_build/libmapn_oracle.so`` (fp32, exact op order of the HLSL);
``model_fp64`` is an independent numpy float64 model used as a second opinion on small N.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmapn_oracle.so")
_SRC = os.path.join(_HERE, "mapn_oracle.c")

MODE_ALL_PAIRS = 0
MODE_CENTRAL_WELL = 1


class Params(C.Structure):
    """nBodyGravityCS.hlsl:37-38 and Compute.cpp:545-546 defaults."""

    _fields_ = [("mass", C.c_float), ("soft2", C.c_float), ("dt", C.c_float), ("damping", C.c_float)]

    def __init__(self, mass=70000.0, soft2=25.0, dt=0.1, damping=1.0):
        super().__init__(mass, soft2, dt, damping)


SUM_REFERENCE, SUM_FP64_ACC, SUM_ORDER_MATCHED = 0, 1, 3


class SumSpec(C.Structure):
    """Diagnostic summation variant of the all-pairs step (mapn_oracle.c header): the reference
    order, fp64 accumulation of the fp32 pair terms, or the device kernel's chunked order + fusion
    (waves, sb = the device plan: mapn_kernel_stats.block_x // 64, .grid_y)."""

    _fields_ = [("mode", C.c_int), ("waves", C.c_uint32), ("sb", C.c_uint32)]

    def __init__(self, mode=SUM_REFERENCE, waves=1, sb=1):
        super().__init__(mode, waves, sb)


class SymShape(C.Structure):
    """Shape of the device's SYMMETRIC launch plan (include/mapn.h, mapn_sym_plan_info) for the order-matched
    restatement of that kernel (mapn_oracle.c, ORDER_MATCHED_SYM)."""

    _fields_ = [(k, C.c_uint32) for k in ("nb", "groups", "windows", "parts", "waves", "brows", "max_meetings", "table_stride", "sets")]


def sym_plan_args(plan):
    """(SymShape, windows, tables) from a plan object as the product's binding returns it (duck-typed: attributes
    nb, groups, parts, waves, brows, max_meetings, table_stride, sets and the uint32 arrays windows [k, 4], tables)."""
    win = np.ascontiguousarray(plan.windows, np.uint32)
    tab = np.ascontiguousarray(plan.tables, np.uint32)[:win.shape[0] * plan.table_stride]   # (a class-aware plan appends its workgroup map -- which workgroup runs a (block, part) does not change any sum)
    shape = SymShape(plan.nb, plan.groups, win.shape[0], plan.parts, plan.waves, plan.brows, plan.max_meetings, plan.table_stride, getattr(plan, "sets", 2))
    assert tab.size == win.shape[0] * plan.table_stride
    return shape, win, tab


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc is in the image on both the CPU and the GPU box)."""
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(_SRC)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _SO


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class Oracle:
    def __init__(self):
        self.lib = lib = C.CDLL(build())
        lib.mapn_oracle_pair_term.argtypes = [_f32p, _f32p, _f32p, C.c_float, C.c_int, C.c_float]
        lib.mapn_oracle_pair_term.restype = None
        lib.mapn_oracle_step_central_well.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.POINTER(Params)]
        lib.mapn_oracle_step_central_well.restype = None
        lib.mapn_oracle_step_all_pairs.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Params), C.c_int]
        lib.mapn_oracle_step_all_pairs.restype = C.c_int
        lib.mapn_oracle_step_all_pairs_ex.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Params), C.c_int, C.POINTER(SumSpec)]
        lib.mapn_oracle_step_all_pairs_ex.restype = C.c_int
        _u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
        lib.mapn_oracle_step_all_pairs_sym.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.POINTER(Params), C.c_int, C.POINTER(SymShape), _u32p, _u32p]
        lib.mapn_oracle_step_all_pairs_sym.restype = C.c_int
        lib.mapn_oracle_step_all_pairs_sym_split.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.POINTER(Params), C.c_int,
                                                             C.POINTER(SymShape), _u32p, _u32p, C.c_uint32, C.c_uint32]
        lib.mapn_oracle_step_all_pairs_sym_split.restype = C.c_int
        _u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
        lib.mapn_oracle_step_all_pairs_sym_sharded.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.POINTER(Params), C.c_int, C.c_uint32,
                                                               C.POINTER(SymShape), _u32p, _u32p, _u64p, C.c_uint32, C.c_int32]
        lib.mapn_oracle_step_all_pairs_sym_sharded.restype = C.c_int
        lib.mapn_oracle_step_all_pairs_sym_sharded_split.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.POINTER(Params), C.c_int, C.c_uint32,
                                                                     C.POINTER(SymShape), _u32p, _u32p, _u64p, C.c_uint32, _u32p, _u32p]
        lib.mapn_oracle_step_all_pairs_sym_sharded_split.restype = C.c_int
        lib.mapn_oracle_step_all_pairs_f64.argtypes = [_f64p, _f64p, _f64p, _f64p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Params), C.c_int]
        lib.mapn_oracle_step_all_pairs_f64.restype = C.c_int
        lib.mapn_oracle_accel_all_pairs.argtypes = [_f32p, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
        lib.mapn_oracle_accel_all_pairs.restype = C.c_int
        lib.mapn_oracle_simulate.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(Params), C.c_int]
        lib.mapn_oracle_simulate.restype = C.c_uint32
        lib.mapn_oracle_active_bodies.argtypes = [C.c_int, C.c_uint32]
        lib.mapn_oracle_active_bodies.restype = C.c_uint32
        lib.mapn_oracle_fast_rand.argtypes = [C.POINTER(C.c_uint32)]
        lib.mapn_oracle_fast_rand.restype = C.c_int
        lib.mapn_oracle_srand_sse.argtypes = [C.POINTER(C.c_uint32 * 4), C.c_uint32]
        lib.mapn_oracle_srand_sse.restype = None
        lib.mapn_oracle_rand_sse.argtypes = [C.POINTER(C.c_uint32 * 4), C.POINTER(C.c_int * 4)]
        lib.mapn_oracle_rand_sse.restype = None
        lib.mapn_oracle_initial_state.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float, _f32p, _f32p]
        lib.mapn_oracle_initial_state.restype = None
        lib.mapn_oracle_initial_state_ex.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_float, C.c_float, _f32p, _f32p]
        lib.mapn_oracle_initial_state_ex.restype = None
        lib.mapn_oracle_mt_uniform.argtypes = [C.c_uint32, C.c_uint32, _f32p]
        lib.mapn_oracle_mt_uniform.restype = None
        lib.mapn_oracle_cbuffer.argtypes = [C.c_uint32, C.POINTER(C.c_uint32 * 4), C.POINTER(C.c_float * 4)]
        lib.mapn_oracle_cbuffer.restype = None
        lib.mapn_oracle_hardware_threads.restype = C.c_int

    # -- scalar pieces -------------------------------------------------------------------
    def pair_term(self, ai, bj, bi, mass=70000.0, particles=1, soft2=25.0):
        ai = np.array(ai, dtype=np.float32)
        self.lib.mapn_oracle_pair_term(ai, np.asarray(bj, np.float32), np.asarray(bi, np.float32), mass, particles, soft2)
        return ai

    def active_bodies(self, num_active, n):
        return int(self.lib.mapn_oracle_active_bodies(int(num_active), int(n)))

    def fast_rand(self, seed, count):
        st = C.c_uint32(seed)
        return [self.lib.mapn_oracle_fast_rand(C.byref(st)) for _ in range(count)]

    def rand_sse(self, seed, calls):
        st = (C.c_uint32 * 4)()
        self.lib.mapn_oracle_srand_sse(C.byref(st), seed)
        out = []
        for _ in range(calls):
            o = (C.c_int * 4)()
            self.lib.mapn_oracle_rand_sse(C.byref(st), C.byref(o))
            out.append(list(o))
        return out

    def cbuffer(self, n):
        p, f = (C.c_uint32 * 4)(), (C.c_float * 4)()
        self.lib.mapn_oracle_cbuffer(n, C.byref(p), C.byref(f))
        return list(p), np.array(list(f), np.float32)

    def hardware_threads(self):
        return int(self.lib.mapn_oracle_hardware_threads())

    _best_threads = None

    @staticmethod
    def cpu_quota_cores():
        """The CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown.  The GPU box
        SHOWS 256 hardware threads and grants 16 cores' worth of time ('1600000 100000'): what bounds the oracle there."""
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            return None if quota == "max" else max(1, int(int(quota) / int(period) + 0.5))
        except (OSError, ValueError):
            pass
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return None if quota <= 0 else max(1, int(quota / period + 0.5))
        except (OSError, ValueError):
            return None

    def best_threads(self):
        """The thread count a step runs FASTEST at on this host: all, half, a quarter or an eighth of the hardware threads, timed once per
        process on two 32 768-body steps each (~0.1 s).  A step creates and joins its workers, and a container may be given fewer cores
        than it shows (cpu_quota_cores: the GPU box shows 256 hardware threads and grants 16 cores' worth of time; there 64 threads run a
        65 536-body step in 30 - 40 ms, 256 in 42; the quota and twice the quota are candidates too)."""
        if Oracle._best_threads is None:
            import time
            hw, n = self.hardware_threads(), 32768
            pos, vel = self.initial_state(n, seed=1)
            prm = Params(mass=70000.0 / n)
            best = None
            cands = {hw, max(1, hw // 2), max(1, hw // 4), max(1, hw // 8)}
            q = self.cpu_quota_cores()
            if q:
                cands |= {min(hw, q), min(hw, 2 * q)}
            for th in sorted(cands, reverse=True):
                self.step_slice(pos, vel, 0, n, params=prm, threads=th)
                t0 = time.perf_counter()
                for _ in range(2):
                    self.step_slice(pos, vel, 0, n, params=prm, threads=th)
                dt = time.perf_counter() - t0
                if best is None or dt < 0.97 * best[0]:        # (ties go to the larger count)
                    best = (dt, th)
            Oracle._best_threads = best[1]
        return Oracle._best_threads

    # -- state ---------------------------------------------------------------------------
    def initial_state(self, n, seed=1, spread=400.0, speed=15.0, variant=0):
        pos = np.zeros((n, 4), np.float32)
        vel = np.zeros((n, 3), np.float32)
        self.lib.mapn_oracle_initial_state_ex(variant, seed, n, spread, speed, pos, vel)
        return pos, vel

    def mt_uniform(self, seed, count):
        out = np.zeros(count, np.float32)
        self.lib.mapn_oracle_mt_uniform(seed, count, out)
        return out

    def accel_all_pairs(self, pos, first=0, count=None, mass=70000.0, soft2=25.0):
        n = pos.shape[0]
        count = n - first if count is None else count
        out = np.zeros((count, 3), np.float32)
        self.lib.mapn_oracle_accel_all_pairs(np.ascontiguousarray(pos, np.float32), out, n, first, count, mass, soft2)
        return out

    def step_slice(self, pos, vel, first, count, mode=MODE_ALL_PAIRS, params=None, threads=0, sum_spec=None):
        """Advance bodies [first, first+count) only; returns (new_pos_slice, new_vel_slice)."""
        params = params or Params()
        if sum_spec is not None and mode == MODE_ALL_PAIRS:
            pos = np.ascontiguousarray(pos, np.float32)
            vel = np.ascontiguousarray(vel, np.float32)
            npos, nvel = pos.copy(), vel.copy()
            if count:
                rc = self.lib.mapn_oracle_step_all_pairs_ex(pos, vel, npos, nvel, pos.shape[0], first, count, C.byref(params), threads, C.byref(sum_spec))
                assert rc == 0, rc
            return npos[first:first + count].copy(), nvel[first:first + count].copy()
        pos = np.ascontiguousarray(pos, np.float32)
        vel = np.ascontiguousarray(vel, np.float32)
        npos, nvel = pos.copy(), vel.copy()
        if count:
            if mode == MODE_CENTRAL_WELL:
                self.lib.mapn_oracle_step_central_well(pos, vel, npos, nvel, first, count, C.byref(params))
            else:
                self.lib.mapn_oracle_step_all_pairs(pos, vel, npos, nvel, pos.shape[0], first, count, C.byref(params), threads)
        return npos[first:first + count].copy(), nvel[first:first + count].copy()


def step_sym_sharded(oracle, pos, vel, params, rank_plans, threads=0, only_rank=-1):
    """One all-active step of the SYMMETRIC kernel SHARDED over len(rank_plans) ranks in the device's summation order
    (ORDER_MATCHED_SHARDED): rank_plans[r] = rank r's launch plan (duck-typed like sym_plan_args' argument: the plan
    mapn_get_sym_plan returned on that rank).  only_rank >= 0: that rank alone, receiving nothing from the others (the device's loopback
    hook MAPN_P2P_LOOPBACK=2; the other entries of rank_plans are then not looked at beyond their shape).  Returns (new_pos, new_vel)."""
    world = len(rank_plans)
    n = pos.shape[0]
    args = [sym_plan_args(pl) for pl in rank_plans]
    shapes = (SymShape * world)(*[a[0] for a in args])
    for a in args:
        assert a[1].shape[0] == 1, "the sharded step is made in one window"
    wins = np.ascontiguousarray(np.concatenate([a[1].reshape(-1) for a in args]), np.uint32)
    offs = np.zeros(world, np.uint64)
    tabs = []
    at = 0
    for r, a in enumerate(args):
        offs[r] = at
        tabs.append(a[2]); at += a[2].size
    tab = np.ascontiguousarray(np.concatenate(tabs), np.uint32)
    count = n // world
    G = 8 if count <= 16384 else 4 if count <= 65536 else 1          # exchange_threads_per_body (csrc/mapn_sym.hip)
    pos = np.ascontiguousarray(pos, np.float32); vel = np.ascontiguousarray(vel, np.float32)
    npos, nvel = pos.copy(), vel.copy()
    rc = oracle.lib.mapn_oracle_step_all_pairs_sym_sharded(pos, vel, npos, nvel, n, C.byref(params), threads, world, shapes, wins, tab, offs, G, int(only_rank))
    assert rc == 0, rc
    return npos, nvel


def step_sym_sharded_split(oracle, pos, vel, params, n_active, rank_plans, rank_frozen, threads=0):
    """One PARTIALLY ACTIVE step of a job SHARDED over len(rank_plans) ranks in the device's summation order (ORDER_MATCHED_SHARDED_SPLIT):
    the bodies [0, n_active) of the whole job advance.  rank_plans[r] = the plan of rank r's blocks in the ACTIVE ring (as
    mapn_get_split_plan returned it on that rank) or None where the rank owns no active body; rank_frozen[r] = (waves, sb) of the
    one-sided launch over the frozen bodies rank r owns, or None where it owns none.  Returns (new_pos, new_vel); the frozen bodies are
    returned as they came."""
    world = len(rank_plans)
    n = pos.shape[0]
    args = [sym_plan_args(pl) if pl is not None else None for pl in rank_plans]
    shapes = (SymShape * world)(*[a[0] if a is not None else SymShape() for a in args])
    wins = np.zeros(4 * world, np.uint32)
    offs = np.zeros(world, np.uint64)
    tabs, at = [np.zeros(1, np.uint32)], 1
    for r, a in enumerate(args):
        if a is None:
            continue
        assert a[1].shape[0] == 1, "the sharded step is made in one window"
        wins[4 * r:4 * r + 4] = a[1].reshape(-1)
        offs[r] = at
        tabs.append(a[2]); at += a[2].size
    tab = np.ascontiguousarray(np.concatenate(tabs), np.uint32)
    fw = np.array([f[0] if f is not None else 0 for f in rank_frozen], np.uint32)
    fs = np.array([f[1] if f is not None else 0 for f in rank_frozen], np.uint32)
    count = n // world
    G = 8 if count <= 16384 else 4 if count <= 65536 else 1          # exchange_threads_per_body (csrc/mapn_sym.hip)
    pos = np.ascontiguousarray(pos, np.float32); vel = np.ascontiguousarray(vel, np.float32)
    npos, nvel = pos.copy(), vel.copy()
    rc = oracle.lib.mapn_oracle_step_all_pairs_sym_sharded_split(pos, vel, npos, nvel, n, int(n_active), C.byref(params), threads, world, shapes, wins, tab, offs, G, fw, fs)
    assert rc == 0, rc
    return npos, nvel


class OracleSim:
    """Host-array twin of the reference's ``Compute`` object: two ping-pong buffer pairs, a
    buffer index and ``simulate(num_active)`` with Compute.cpp:1009-1055 semantics."""

    def __init__(self, oracle: Oracle, pos, vel, mode=MODE_ALL_PAIRS, params=None, threads=0, sum_spec=None, sym_plan=None, split_plan=None):
        self.o = oracle
        # the device's PARTIALLY ACTIVE step in its split form (ORDER_MATCHED_SPLIT): (split, plan) as the product's binding returns them
        # (duck-typed: split.active / .frozen_waves / .frozen_sb; plan as for sym_plan); steps with that num_active take this order
        self.split = (split_plan[0], sym_plan_args(split_plan[1])) if split_plan is not None else None
        self.sum_spec = sum_spec            # diagnostic summation variant (all-pairs, num_active = N only)
        self.sym = sym_plan_args(sym_plan) if sym_plan is not None else None   # ... or the symmetric kernel's order (ORDER_MATCHED_SYM)
        self.n = pos.shape[0]
        self.pos = [np.array(pos, np.float32, order="C"), np.array(pos, np.float32, order="C")]   # Compute.cpp:881-882
        self.vel = [np.array(vel, np.float32, order="C"), np.array(vel, np.float32, order="C")]   # Compute.cpp:903-904
        self.buffer_index = 0                                                                     # Compute.cpp:80
        self.mode, self.params, self.threads = mode, params or Params(), threads

    def simulate(self, num_active=None, steps=1):
        num_active = self.n if num_active is None else num_active
        if self.split is not None and self.o.active_bodies(num_active, self.n) == self.split[0].active:
            assert self.mode == MODE_ALL_PAIRS
            split, (shape, win, tab) = self.split
            for _ in range(steps):
                w, r = self.buffer_index, 1 - self.buffer_index
                rc = self.o.lib.mapn_oracle_step_all_pairs_sym_split(self.pos[r], self.vel[r], self.pos[w], self.vel[w], self.n, split.active,
                                                                     C.byref(self.params), self.threads, C.byref(shape), win, tab,
                                                                     split.frozen_waves, split.frozen_sb)
                assert rc == 0, rc
                self.buffer_index = 1 - self.buffer_index
            return
        if self.sym is not None:
            assert self.mode == MODE_ALL_PAIRS and self.o.active_bodies(num_active, self.n) == self.n
            shape, win, tab = self.sym
            for _ in range(steps):
                w, r = self.buffer_index, 1 - self.buffer_index
                rc = self.o.lib.mapn_oracle_step_all_pairs_sym(self.pos[r], self.vel[r], self.pos[w], self.vel[w], self.n,
                                                               C.byref(self.params), self.threads, C.byref(shape), win, tab)
                assert rc == 0, rc
                self.buffer_index = 1 - self.buffer_index
            return
        if self.sum_spec is not None and self.mode == MODE_ALL_PAIRS:
            active = self.o.active_bodies(num_active, self.n)
            for _ in range(steps):
                w, r = self.buffer_index, 1 - self.buffer_index
                if active:
                    rc = self.o.lib.mapn_oracle_step_all_pairs_ex(self.pos[r], self.vel[r], self.pos[w], self.vel[w], self.n, 0, active,
                                                                  C.byref(self.params), self.threads, C.byref(self.sum_spec))
                    assert rc == 0, rc
                self.buffer_index = 1 - self.buffer_index
            return
        for _ in range(steps):
            self.buffer_index = int(self.o.lib.mapn_oracle_simulate(
                self.pos[0], self.pos[1], self.vel[0], self.vel[1], self.buffer_index, self.n,
                num_active, self.mode, C.byref(self.params), self.threads))

    @property
    def latest(self):
        """(pos, vel) most recently written = buffer 1 - buffer_index after the flip."""
        r = 1 - self.buffer_index
        return self.pos[r], self.vel[r]


class OracleSim64:
    """The all-pairs step in double on DOUBLE state (mapn_oracle_step_all_pairs_f64): the discrete
    map itself, the yardstick both fp32 paths are measured against.  num_active = N only."""

    def __init__(self, oracle: Oracle, pos, vel, params=None, threads=0):
        self.o, self.n = oracle, pos.shape[0]
        self.pos = [np.array(pos, np.float64, order="C"), np.array(pos, np.float64, order="C")]
        self.vel = [np.array(vel, np.float64, order="C"), np.array(vel, np.float64, order="C")]
        self.buffer_index = 0
        self.params, self.threads = params or Params(), threads

    def simulate(self, steps=1):
        for _ in range(steps):
            w, r = self.buffer_index, 1 - self.buffer_index
            rc = self.o.lib.mapn_oracle_step_all_pairs_f64(self.pos[r], self.vel[r], self.pos[w], self.vel[w], self.n, 0, self.n,
                                                           C.byref(self.params), self.threads)
            assert rc == 0, rc
            self.buffer_index = 1 - self.buffer_index

    @property
    def latest(self):
        r = 1 - self.buffer_index
        return self.pos[r], self.vel[r]
