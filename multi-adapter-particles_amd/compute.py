"""Python mirror of the reference's ``class Compute`` over the C ABI.

Method names, argument meaning and error behaviour follow reference/Particles/Compute.h:33-78
and AdapterShared.h:51-60, so a parity test reads like ``Particles::Draw`` (Particles.cpp:446-448):

    fence = compute.GetFenceValue()
    compute.Simulate(num_simulated, fence)

Errors raise ``MapnError`` (the reference throws HrException, DXSampleHelper.h:29-46).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, KernelStats, MapnError, SharedHandles, check, load_library


def generate_initial_state(num_particles: int, seed: int = 1, spread: float = 400.0, speed: float = 15.0,
                           variant: int = _lib.INIT_LCG):
    """LoadParticles / InitializeParticles made deterministic (csrc/mapn_init.cpp); `variant`
    selects which of the reference's three #if variants supplies the randomness."""
    lib = load_library()
    pos = np.zeros((num_particles, 4), np.float32)
    vel = np.zeros((num_particles, 3), np.float32)
    check(lib.mapn_generate_initial_state_ex(variant, seed, num_particles, spread, speed,
                                             pos.ctypes.data_as(_lib._fp), vel.ctypes.data_as(_lib._fp)))
    return pos, vel


class Compute:
    """``Compute(numParticles, adapter, useIntelCommandQueueExtension, old=None)`` (Compute.h:36-39).

    ``device`` replaces the DXGI adapter.  The keyword arguments are what the reference
    hard-codes (nBodyGravityCS.hlsl:37-38, Compute.cpp:545-546, defines.h:39,42) plus the
    sharding coordinates of the multi-GPU mode.
    """

    def __init__(self, num_particles: int, device: int = 0, use_intel_command_queue_extension: bool = False,
                 old: "Compute | None" = None, *, force_mode: int = _lib.FORCE_ALL_PAIRS,
                 mass: float = 70000.0, softening_squared: float = 25.0, dt: float = 0.1,
                 damping: float = 1.0, seed: int = 1, spread: float = 400.0, initial_speed: float = 15.0,
                 flags: int = 0, kernel: int = _lib.KERNEL_AUTO, rank: int = 0, world_size: int = 1,
                 init_variant: int = _lib.INIT_LCG):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        cfg = Config()
        check(self._lib.mapn_config_default(C.byref(cfg)))
        cfg.num_particles, cfg.device, cfg.force_mode = num_particles, device, force_mode
        cfg.mass, cfg.softening_squared, cfg.dt, cfg.damping = mass, softening_squared, dt, damping
        cfg.seed, cfg.spread, cfg.initial_speed = seed, spread, initial_speed
        cfg.flags, cfg.kernel, cfg.rank, cfg.world_size = flags, kernel, rank, world_size
        cfg.init_variant = init_variant
        self.config = cfg
        self.num_particles = num_particles
        if old is not None:
            check(self._lib.mapn_create_from(C.byref(cfg), old._ctx, C.byref(self._ctx)))   # CopyState
        else:
            check(self._lib.mapn_create(C.byref(cfg), C.byref(self._ctx)))
        self._using_intel_ext = False   # vendor hint only; no AMD analogue

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.mapn_destroy(self._ctx)          # ~Compute: WaitForGpu, then free
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the reference's public surface ----------------------------------------------------------
    def Simulate(self, num_active_particles: int, shared_fence_value: int = 0):
        """Compute.cpp:1009-1055.  Asynchronous enqueue of one step."""
        check(self._lib.mapn_simulate(self._ctx, int(num_active_particles), int(shared_fence_value)))

    def GetFenceValue(self) -> int:
        """Compute.h:64 -- the value the next Simulate will signal."""
        return int(self._lib.mapn_fence_value(self._ctx))

    def GetCompletedValue(self) -> int:
        return int(self._lib.mapn_completed_value(self._ctx))

    def WaitForGpu(self):
        """Compute.cpp:928-940."""
        check(self._lib.mapn_wait_idle(self._ctx))

    def GetSharedHandles(self, consumer_fence: bool = True) -> SharedHandles:
        """Compute.cpp:944-950: export the position ping-pong pair + completion event; attaching
        the consumer's fence makes Simulate honour ``shared_fence_value``."""
        check(self._lib.mapn_set_consumer(self._ctx, 1 if consumer_fence else 0))
        h = SharedHandles()
        check(self._lib.mapn_get_shared_handles(self._ctx, C.byref(h)))
        return h

    def ConsumerSignal(self, value: int, hip_event: int | None = None):
        """The consumer's ``Signal(fence, value)`` (Render.cpp:826): host-side, or device-side
        when given a hipEvent_t already recorded on the consumer's stream."""
        if hip_event is None:
            check(self._lib.mapn_consumer_signal(self._ctx, int(value)))
        else:
            check(self._lib.mapn_consumer_signal_event(self._ctx, int(value), C.c_void_p(hip_event)))

    def set_timeouts(self, p2p_ms: int = 0, consumer_ms: int = 0):
        """Bounds of the device-side waits (0 = unchanged): peer-to-peer exchange, queued consumer fence."""
        check(self._lib.mapn_set_timeouts(self._ctx, int(p2p_ms), int(consumer_ms)))

    def ipc_export(self) -> bytes:
        """GetSharedHandles for a consumer in ANOTHER process: hipIpc handles of the position heap,
        the status / fence block and the step-done event (Compute.cpp:944-950)."""
        buf = C.create_string_buffer(_lib.IPC_BLOB_BYTES)
        check(self._lib.mapn_ipc_export(self._ctx, C.cast(buf, C.c_void_p)))
        return buf.raw

    def SetAsync(self, buffers, buffer_index: int):
        """Compute.cpp:956-987: compute directly into two caller-owned device float4[N] buffers."""
        arr = (C.c_void_p * 2)(C.c_void_p(int(buffers[0])), C.c_void_p(int(buffers[1])))
        check(self._lib.mapn_adopt_position_buffers(self._ctx, C.byref(arr), int(buffer_index)))

    def ResetFromAsyncHelper(self):
        """Compute.cpp:260-298."""
        check(self._lib.mapn_reset_from_async(self._ctx))

    def SetUseIntelCommandQueueExtension(self, desired: bool):
        check(self._lib.mapn_set_use_intel_command_queue_extension(self._ctx, int(bool(desired))))

    def GetUsingIntelCommandQueueExtension(self) -> bool:
        return bool(self._lib.mapn_get_using_intel_command_queue_extension(self._ctx))

    def GetIsUMA(self) -> bool:
        return bool(self._lib.mapn_get_is_uma(self._ctx))

    def GetGpuTimes(self):
        """AdapterShared.h:51 -> vector<pair<float seconds, string name>> (D3D12GpuTimer.h:54-55)."""
        return [(float(self._lib.mapn_last_step_seconds(self._ctx)), self._lib.mapn_timer_name().decode())]

    # -- state hand-off ------------------------------------------------------------------------
    @property
    def buffer_index(self) -> int:
        return int(self._lib.mapn_buffer_index(self._ctx))

    def upload_state(self, pos, vel):
        pos = np.ascontiguousarray(pos, np.float32)
        vel = np.ascontiguousarray(vel, np.float32)
        if pos.shape != (self.num_particles, 4) or vel.shape != (self.num_particles, 3):
            raise ValueError("upload_state: pos must be (N,4) and vel (N,3) float32")
        check(self._lib.mapn_upload_state(self._ctx, pos.ctypes.data_as(_lib._fp), vel.ctypes.data_as(_lib._fp)))

    def download_state(self):
        pos = np.empty((self.num_particles, 4), np.float32)
        vel = np.empty((self.num_particles, 3), np.float32)
        check(self._lib.mapn_download_state(self._ctx, pos.ctypes.data_as(_lib._fp), vel.ctypes.data_as(_lib._fp)))
        return pos, vel

    def download_buffer(self, index: int):
        pos = np.empty((self.num_particles, 4), np.float32)
        vel = np.empty((self.num_particles, 3), np.float32)
        check(self._lib.mapn_download_buffer(self._ctx, index, pos.ctypes.data_as(_lib._fp), vel.ctypes.data_as(_lib._fp)))
        return pos, vel

    def copy_positions_async(self, num_copied: int, dst: int, consumer_stream: int = 0):
        """Render::CopySimulationResults (Render.cpp:789-831) on the consumer's stream."""
        check(self._lib.mapn_copy_positions_async(self._ctx, int(num_copied), C.c_void_p(dst), C.c_void_p(consumer_stream)))

    def save_snapshot(self, path: str):
        check(self._lib.mapn_save_snapshot(self._ctx, str(path).encode()))

    def load_snapshot(self, path: str):
        check(self._lib.mapn_load_snapshot(self._ctx, str(path).encode()))

    def cbuffer(self):
        p, f = (C.c_uint32 * 4)(), (C.c_float * 4)()
        check(self._lib.mapn_get_cbuffer(self._ctx, C.byref(p), C.byref(f)))
        return list(p), np.array(list(f), np.float32)

    # -- sharded mode --------------------------------------------------------------------------
    def shard_range(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(self._lib.mapn_shard_range(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        check(load_library().mapn_comm_get_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    def comm_init(self, unique_id: bytes):
        if len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = C.create_string_buffer(unique_id, _lib.UNIQUE_ID_BYTES)
        check(self._lib.mapn_comm_init(self._ctx, C.cast(buf, C.c_void_p)))

    def comm_init_torch(self):
        """Rendezvous through an already-initialised torch.distributed group: rank 0 creates the
        RCCL unique id, everybody receives it, then all ranks join the communicator."""
        import torch.distributed as dist
        ids, err = [None], None
        if dist.get_rank() == 0:
            try:
                ids = [self.comm_unique_id()]
            except MapnError as e:                 # (the peers are in the broadcast below: send them the verdict, then report)
                err = e
        dist.broadcast_object_list(ids, src=0)
        if err is not None:
            raise err
        if ids[0] is None:
            raise MapnError(-4, "rank 0 could not create the RCCL unique id")      # (MAPN_ERR_COMM)
        self.comm_init(ids[0])

    def set_gather_algorithm(self, algorithm: int):
        """0 = ncclAllGather, 1 = grouped ncclSend/ncclRecv (collective choice: all ranks alike)."""
        check(self._lib.mapn_set_gather_algorithm(self._ctx, int(algorithm)))

    def p2p_export(self) -> bytes:
        buf = C.create_string_buffer(_lib.P2P_BLOB_BYTES)
        check(self._lib.mapn_p2p_export(self._ctx, C.cast(buf, C.c_void_p)))
        return buf.raw

    def p2p_import(self, blobs):
        raw = b"".join(blobs)
        if len(raw) != _lib.P2P_BLOB_BYTES * len(blobs):
            raise ValueError("every blob must be %d bytes" % _lib.P2P_BLOB_BYTES)
        buf = C.create_string_buffer(raw, len(raw))
        check(self._lib.mapn_p2p_import(self._ctx, C.cast(buf, C.c_void_p), len(blobs)))

    def p2p_setup_torch(self):
        """Exchange the hipIpc blobs through an initialised torch.distributed group (any backend)
        and map every peer's buffers; afterwards set_gather_algorithm(2) selects the exchange."""
        import torch.distributed as dist
        blobs = [None] * dist.get_world_size()
        mine, err = None, None
        try:
            mine = self.p2p_export()
        except MapnError as e:                     # (the peers are in the collective below: take part in it, then report)
            err = e
        dist.all_gather_object(blobs, mine)
        if err is not None:
            raise err
        missing = [r for r, b in enumerate(blobs) if b is None]
        if missing:
            raise MapnError(-4, f"peer-to-peer set-up: rank(s) {missing} could not export their buffers")      # (MAPN_ERR_COMM)
        self.p2p_import(blobs)

    def p2p_status(self) -> int:
        return int(self._lib.mapn_p2p_status(self._ctx))

    def replica_checksum(self):
        """(sum of the 32-bit words of position buffer 0, of buffer 1): equal on all ranks after any correct exchange."""
        out = (C.c_uint64 * 2)()
        check(self._lib.mapn_replica_checksum(self._ctx, C.byref(out)))
        return int(out[0]), int(out[1])

    def set_external_gather(self, enabled: bool = True):
        check(self._lib.mapn_set_external_gather(self._ctx, int(bool(enabled))))

    # -- tuning / introspection ----------------------------------------------------------------
    def set_force_plan(self, kernel: int, bodies_per_lane: int = 4, waves: int = 8, sb: int = 1, fused=True):
        """fused: False/0 two launches (rows + reduce_integrate), True/1 one launch, 2 ticket form always."""
        check(self._lib.mapn_set_force_plan(self._ctx, kernel, bodies_per_lane, waves, sb, int(fused)))

    def set_sym_plan(self, waves: int = 0, parts: int = 0, taper1: int = 0, taper2: int = 0, groups_per_window: int = 0, wave_bias=(1, 1)):
        """Shape of the symmetric kernel's launches (waves = parts = 0: the default); re-allocates its scratch.
        wave_bias = (hi, lo): the first half of a workgroup's waves carries hi : lo of its steps against the second half."""
        check(self._lib.mapn_set_sym_plan(self._ctx, waves, parts, taper1, taper2, groups_per_window, int(wave_bias[0]), int(wave_bias[1])))

    def calibrate_sym_xcds(self, steps: int = 4):
        """Relative speeds of the eight XCDs under the symmetric kernel (1024 = the fastest), from `steps` stamped REAL steps."""
        w = (C.c_uint32 * 8)()
        check(self._lib.mapn_calibrate_sym_xcds(self._ctx, int(steps), C.byref(w)))
        return list(w)

    def set_sym_xcd_weights(self, weights=None):
        """Spread the parts of every block over the dies, sized by their speed (None: back to the default plan)."""
        if weights is None:
            check(self._lib.mapn_set_sym_xcd_weights(self._ctx, None))
        else:
            w = (C.c_uint32 * 8)(*[int(x) for x in weights])
            check(self._lib.mapn_set_sym_xcd_weights(self._ctx, C.byref(w)))

    def sym_plan(self) -> "SymPlan":
        """The plan the symmetric kernel runs in this context (raises MapnError if it does not run)."""
        info = _lib.SymPlanInfo()
        check(self._lib.mapn_get_sym_plan(self._ctx, C.byref(info), None, 0, None, 0))
        win = np.zeros((info.windows, 4), np.uint32)
        tab = np.zeros(info.windows * info.table_stride + info.wgmap_entries, np.uint32)
        u32p = C.POINTER(C.c_uint32)
        check(self._lib.mapn_get_sym_plan(self._ctx, C.byref(info), win.ctypes.data_as(u32p), win.size, tab.ctypes.data_as(u32p), tab.size))
        return SymPlan(info, win, tab)

    def split_plan(self):
        """(split, plan) of the latest PARTIALLY ACTIVE step in its split form: `split` names the active / frozen counts and the shape
        of the one-sided launch over the frozen bodies, `plan` is the symmetric plan of the active bodies (raises MapnError before
        such a step has run)."""
        info, split = _lib.SymPlanInfo(), _lib.SplitInfo()
        check(self._lib.mapn_get_split_plan(self._ctx, C.byref(split), C.byref(info), None, 0, None, 0))
        win = np.zeros((info.windows, 4), np.uint32)
        tab = np.zeros(info.windows * info.table_stride + info.wgmap_entries, np.uint32)
        u32p = C.POINTER(C.c_uint32)
        check(self._lib.mapn_get_split_plan(self._ctx, C.byref(split), C.byref(info), win.ctypes.data_as(u32p), win.size, tab.ctypes.data_as(u32p), tab.size))
        return split, SymPlan(info, win, tab)

    def set_shard_overlap(self, enabled: bool):
        check(self._lib.mapn_set_shard_overlap(self._ctx, int(bool(enabled))))

    def measure_clock(self, steps: int = 8) -> "_lib.ClockInfo":
        """The shader clock the chip holds under the force kernel (stamped diagnostic steps)."""
        info = _lib.ClockInfo()
        check(self._lib.mapn_measure_clock(self._ctx, int(steps), C.byref(info)))
        return info

    def set_timers(self, interval: int):
        """0 = off, T >= 1 = time every T-th step."""
        check(self._lib.mapn_set_timers(self._ctx, int(interval)))

    def kernel_stats(self, reset: bool = False) -> KernelStats:
        st = KernelStats()
        check(self._lib.mapn_get_kernel_stats(self._ctx, int(reset), C.byref(st)))
        return st

    def step_samples(self):
        """(step index, step ms, force launch ms) arrays of every timed step since the last statistics reset."""
        n = C.c_uint32()
        check(self._lib.mapn_get_step_samples(self._ctx, None, None, None, 0, C.byref(n)))
        idx = np.zeros(max(n.value, 1), np.uint32)
        sm, fm = np.zeros(max(n.value, 1), np.float32), np.zeros(max(n.value, 1), np.float32)
        check(self._lib.mapn_get_step_samples(self._ctx, idx.ctypes.data_as(C.POINTER(C.c_uint32)), sm.ctypes.data_as(_lib._fp),
                                              fm.ctypes.data_as(_lib._fp), idx.size, C.byref(n)))
        return idx[:n.value], sm[:n.value], fm[:n.value]

    @property
    def compute_stream(self) -> int:
        return int(self._lib.mapn_compute_stream(self._ctx) or 0)


class IpcView:
    """The consumer's side of ``Compute.ipc_export()`` in another process: Render::SetShared +
    CopySimulationResults (Render.cpp:222-251, 789-831) on HIP IPC handles."""

    def __init__(self, blob: bytes, device: int = 0):
        self._lib = load_library()
        self._view = C.c_void_p()
        buf = C.create_string_buffer(blob, _lib.IPC_BLOB_BYTES)
        check(self._lib.mapn_ipc_open(C.cast(buf, C.c_void_p), device, C.byref(self._view)))

    def close(self):
        if getattr(self, "_view", None) and self._view.value:
            self._lib.mapn_ipc_close(self._view)
            self._view = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def latest(self):
        """(fence value signalled by the latest published step, buffer index holding its positions)"""
        f, i = C.c_uint64(), C.c_uint32()
        check(self._lib.mapn_ipc_latest(self._view, C.byref(f), C.byref(i)))
        return f.value, i.value

    def positions_ptr(self, buffer_index: int) -> int:
        return int(self._lib.mapn_ipc_positions(self._view, buffer_index) or 0)

    def copy_positions_async(self, buffer_index: int, num_copied: int, dst: int, wait_fence_value: int = 0, consumer_stream: int = 0):
        """copyQueue.Wait(computeFence, wait_fence_value); CopyBufferRegion(...) (Render.cpp:796,814)"""
        check(self._lib.mapn_ipc_copy_positions_async(self._view, buffer_index, num_copied, C.c_void_p(dst),
                                                      int(wait_fence_value), C.c_void_p(consumer_stream)))

    def consumer_signal(self, value: int, consumer_stream: int = 0):
        check(self._lib.mapn_ipc_consumer_signal(self._view, int(value), C.c_void_p(consumer_stream)))


class SymPlan:
    """The symmetric kernel's launch plan as data (include/mapn.h, mapn_sym_plan_info): `windows[k]` =
    (g0, g1, meetings of a class-0 block, of a class-1 block); `bounds(k, cls)[v]` = first linear step of wave v;
    `split(k, cls)[m]` = part holding the head row of meeting m, or SPLIT_NONE."""
    SPLIT_NONE = 0xffffffff

    def __init__(self, info, windows, tables):
        self.info, self.windows, self.tables = info, windows, tables
        self.wave_bias = (int(info.wave_bias[0]), int(info.wave_bias[1]))
        for k in ("nb", "groups", "parts", "taper1", "taper2", "waves", "brows", "max_meetings", "table_stride", "sets", "a0", "nbl", "active_compute_units", "exchange_workgroups", "scratch_bytes",
                  "xcd_mode", "wgmap_offset", "wgmap_entries", "la_flip"):
            setattr(self, k, int(getattr(info, k)))
        self.nwaves = self.parts * self.waves
        self.xcd_weight = list(info.xcd_weight)
        # class-aware XCD weights (xcd_mode 2): class_die[c] = the four dispatch slots class c's blocks run on; wgmap[y, x] = (block, part) of workgroup (x, y)
        self.class_die = [list(info.class_die)[:4], list(info.class_die)[4:]]
        self.wgmap = None
        if self.wgmap_entries:
            m = np.asarray(tables[self.wgmap_offset:self.wgmap_offset + self.wgmap_entries], np.uint32).reshape(self.parts, -1)
            self.wgmap = np.stack([m >> 16, m & 0xffff], axis=-1)

    def set_of(self, cls: int, block_in_launch: int = 0) -> int:
        """Table set of a block: its class, plus 2 * (block mod 8) when the parts are XCD-weighted."""
        return cls + (2 * (block_in_launch % 8) if self.sets > 2 else 0)

    def bounds(self, window: int, set_: int):
        o = window * self.table_stride + set_ * (self.nwaves + 1)
        return self.tables[o:o + self.nwaves + 1]

    def split(self, window: int, set_: int):
        o = window * self.table_stride + self.sets * (self.nwaves + 1) + set_ * self.max_meetings
        return self.tables[o:o + self.max_meetings]


def describe_shard(num_particles: int, rank: int, world_size: int, num_active: int | None = None) -> "_lib.ShardInfo":
    """The sharded mode's host arithmetic for one rank, computed by the library without a device (mapn_shard_describe)."""
    info = _lib.ShardInfo()
    check(load_library().mapn_shard_describe(int(num_particles), int(rank), int(world_size), int(num_particles if num_active is None else num_active), C.byref(info)))
    return info


def describe_shard_split(num_particles: int, rank: int, world_size: int, num_active: int) -> "_lib.ShardSplitInfo":
    """This rank's part of a PARTIALLY ACTIVE step of a sharded job in its split form, without a device (mapn_shard_split_describe)."""
    info = _lib.ShardSplitInfo()
    check(load_library().mapn_shard_split_describe(int(num_particles), int(rank), int(world_size), int(num_active), C.byref(info)))
    return info


def describe_sym_plan(nb: int, groups_per_window: int = 0, parts: int = 32, taper1: int | None = None, taper2: int = 0, waves: int = 4,
                      xcd_weights=None, launch_blocks: int = 0, wave_bias=(1, 1), launch_a0: int = 0, xcd_mode: int = 0) -> SymPlan:
    """The plan of a shape, computed on the host without a device (csrc/mapn_sym_plan.cpp)."""
    lib = load_library()
    info = _lib.SymPlanInfo()
    t1 = parts if taper1 is None else taper1
    xw = C.byref((C.c_uint32 * 8)(*[int(x) for x in xcd_weights])) if xcd_weights is not None else None
    rc = lib.mapn_sym_plan_describe(nb, groups_per_window, parts, t1, taper2, waves, int(wave_bias[0]), int(wave_bias[1]), xw, launch_blocks, launch_a0, xcd_mode, C.byref(info), None, 0, None, 0)
    if rc:
        raise MapnError(rc, info.error.decode(errors="replace"))
    win = np.zeros((info.windows, 4), np.uint32)
    tab = np.zeros(info.windows * info.table_stride + info.wgmap_entries, np.uint32)
    u32p = C.POINTER(C.c_uint32)
    rc = lib.mapn_sym_plan_describe(nb, groups_per_window, parts, t1, taper2, waves, int(wave_bias[0]), int(wave_bias[1]), xw, launch_blocks, launch_a0, xcd_mode, C.byref(info), win.ctypes.data_as(u32p), win.size, tab.ctypes.data_as(u32p), tab.size)
    if rc:
        raise MapnError(rc, info.error.decode(errors="replace"))
    return SymPlan(info, win, tab)


def device_info(device: int = 0) -> _lib.DeviceInfo:
    info = _lib.DeviceInfo()
    check(load_library().mapn_get_device_info(device, C.byref(info)))
    return info


def device_count() -> int:
    return int(load_library().mapn_device_count())
