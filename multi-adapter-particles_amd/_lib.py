"""ctypes binding of include/mapn.h.  Loads the in-tree ``libmapn.so`` and fails loudly if it
is missing -- there is no Python or CPU implementation of the step behind it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmapn.so")
_CSRC = os.path.join(_HERE, "csrc")

FORCE_ALL_PAIRS = 0        # sum of bodyBodyInteraction, nBodyGravityCS.hlsl:44-57
FORCE_CENTRAL_WELL = 1     # CSMain as shipped, nBodyGravityCS.hlsl:92-101
KERNEL_AUTO, KERNEL_LDS, KERNEL_SCALAR, KERNEL_SYMMETRIC = 0, 1, 2, 3
FLAG_USE_GRAPH = 0x1
INIT_LCG, INIT_SSE, INIT_MT = 0, 1, 2
FLAG_NO_INIT = 0x2
FLAG_SHARD_OVERLAP = 0x4
FLAG_STRICT_CONSUMER = 0x8
FLAG_XCD_CALIBRATE = 0x10
IPC_BLOB_BYTES = 256
UNIQUE_ID_BYTES = 128
P2P_BLOB_BYTES = 192


class MapnError(RuntimeError):
    """Counterpart of the reference's HrException (DXSampleHelper.h:29-46)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"mapn status {status}: {message}")
        self.status = status


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("num_particles", C.c_uint32), ("device", C.c_int32),
        ("force_mode", C.c_int32), ("mass", C.c_float), ("softening_squared", C.c_float),
        ("dt", C.c_float), ("damping", C.c_float), ("seed", C.c_uint32), ("spread", C.c_float),
        ("initial_speed", C.c_float), ("flags", C.c_uint32), ("kernel", C.c_int32),
        ("rank", C.c_int32), ("world_size", C.c_int32), ("init_variant", C.c_int32), ("reserved", C.c_int32 * 4),
    ]


class SharedHandles(C.Structure):
    _fields_ = [
        ("positions", C.c_void_p * 2), ("step_done_event", C.c_void_p),
        ("aligned_data_size", C.c_uint64), ("buffer_index", C.c_uint32), ("reserved", C.c_uint32),
    ]


class DeviceInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128), ("arch", C.c_char * 64), ("compute_units", C.c_int32),
        ("clock_khz", C.c_int32), ("wavefront_size", C.c_int32), ("reserved", C.c_int32),
        ("peak_fp32_flops", C.c_double), ("total_memory_bytes", C.c_uint64),
    ]


class ClockInfo(C.Structure):
    _fields_ = [("shader_clock_ghz", C.c_double), ("shader_clock_ghz_p10", C.c_double), ("shader_clock_ghz_p90", C.c_double),
                ("median_wave_cycles", C.c_double), ("waves_stamped", C.c_uint32), ("steps", C.c_uint32)]


class KernelStats(C.Structure):
    _fields_ = [
        ("kernel_name", C.c_char * 64), ("launches", C.c_uint64), ("avg_seconds", C.c_double),
        ("grid_x", C.c_uint32), ("grid_y", C.c_uint32), ("block_x", C.c_uint32),
        ("bodies_per_lane", C.c_uint32), ("j_splits", C.c_uint32), ("fused", C.c_uint32),
        ("grid_z", C.c_uint32), ("epilogue", C.c_uint32), ("force_launches_per_step", C.c_uint32), ("split_active", C.c_uint32),
        ("split_plans_built", C.c_uint32), ("reserved", C.c_uint32),
    ]


class SymPlanInfo(C.Structure):
    _fields_ = [
        ("nb", C.c_uint32), ("groups", C.c_uint32), ("windows", C.c_uint32),
        ("parts", C.c_uint32), ("taper1", C.c_uint32), ("taper2", C.c_uint32), ("waves", C.c_uint32), ("wave_bias", C.c_uint32 * 2),
        ("brows", C.c_uint32), ("max_meetings", C.c_uint32), ("table_stride", C.c_uint32),
        ("sets", C.c_uint32), ("xcd_weight", C.c_uint32 * 8),
        ("xcd_mode", C.c_uint32), ("wgmap_offset", C.c_uint32), ("wgmap_entries", C.c_uint32), ("la_flip", C.c_uint32), ("class_die", C.c_uint32 * 8),
        ("a0", C.c_uint32), ("nbl", C.c_uint32), ("active_compute_units", C.c_uint32), ("exchange_workgroups", C.c_uint32),
        ("scratch_bytes", C.c_uint64), ("error", C.c_char * 256),
    ]


class ShardInfo(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("first", "count", "active_first", "active_count", "sym_applies", "nb", "nbl", "a0", "send_mask", "recv_mask")] + [("reserved", C.c_uint32 * 2)]


class ShardSplitInfo(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("applies", "active", "ring_blocks", "blocks", "first_block", "active_count", "frozen_first", "frozen_count",
                                          "send_mask", "recv_mask")] + [("reserved", C.c_uint32 * 2)]


class SplitInfo(C.Structure):
    _fields_ = [("active", C.c_uint32), ("frozen", C.c_uint32), ("frozen_kernel", C.c_uint32), ("frozen_bodies_per_lane", C.c_uint32),
                ("frozen_waves", C.c_uint32), ("frozen_sb", C.c_uint32), ("frozen_first", C.c_uint32), ("has_plan", C.c_uint32)]


# every symbol include/mapn.h and include/mapn_tuning.h declare: (name, restype, argtypes)
_fp = C.POINTER(C.c_float)
_ctx = C.c_void_p
SIGNATURES = {
    "mapn_abi_version": (C.c_int, []),
    "mapn_tuning_abi_version": (C.c_int, []),
    "mapn_last_error": (C.c_char_p, []),
    "mapn_config_default": (C.c_int, [C.POINTER(Config)]),
    "mapn_create": (C.c_int, [C.POINTER(Config), C.POINTER(_ctx)]),
    "mapn_create_from": (C.c_int, [C.POINTER(Config), _ctx, C.POINTER(_ctx)]),
    "mapn_destroy": (C.c_int, [_ctx]),
    "mapn_simulate": (C.c_int, [_ctx, C.c_int, C.c_uint64]),
    "mapn_fence_value": (C.c_uint64, [_ctx]),
    "mapn_completed_value": (C.c_uint64, [_ctx]),
    "mapn_wait_idle": (C.c_int, [_ctx]),
    "mapn_buffer_index": (C.c_uint32, [_ctx]),
    "mapn_num_particles": (C.c_uint32, [_ctx]),
    "mapn_get_shared_handles": (C.c_int, [_ctx, C.POINTER(SharedHandles)]),
    "mapn_set_consumer": (C.c_int, [_ctx, C.c_int]),
    "mapn_consumer_signal": (C.c_int, [_ctx, C.c_uint64]),
    "mapn_consumer_signal_event": (C.c_int, [_ctx, C.c_uint64, C.c_void_p]),
    "mapn_set_timeouts": (C.c_int, [_ctx, C.c_uint32, C.c_uint32]),
    "mapn_ipc_export": (C.c_int, [_ctx, C.c_void_p]),
    "mapn_ipc_open": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mapn_ipc_close": (C.c_int, [C.c_void_p]),
    "mapn_ipc_latest": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "mapn_ipc_positions": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "mapn_ipc_copy_positions_async": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mapn_ipc_consumer_signal": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "mapn_adopt_position_buffers": (C.c_int, [_ctx, C.POINTER(C.c_void_p * 2), C.c_uint32]),
    "mapn_reset_from_async": (C.c_int, [_ctx]),
    "mapn_last_step_seconds": (C.c_float, [_ctx]),
    "mapn_timer_name": (C.c_char_p, []),
    "mapn_set_use_intel_command_queue_extension": (C.c_int, [_ctx, C.c_int]),
    "mapn_get_using_intel_command_queue_extension": (C.c_int, [_ctx]),
    "mapn_get_is_uma": (C.c_int, [_ctx]),
    "mapn_upload_state": (C.c_int, [_ctx, _fp, _fp]),
    "mapn_download_state": (C.c_int, [_ctx, _fp, _fp]),
    "mapn_download_buffer": (C.c_int, [_ctx, C.c_uint32, _fp, _fp]),
    "mapn_copy_positions_async": (C.c_int, [_ctx, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mapn_save_snapshot": (C.c_int, [_ctx, C.c_char_p]),
    "mapn_load_snapshot": (C.c_int, [_ctx, C.c_char_p]),
    "mapn_generate_initial_state": (C.c_int, [C.c_uint32, C.c_uint32, C.c_float, C.c_float, _fp, _fp]),
    "mapn_generate_initial_state_ex": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_float, C.c_float, _fp, _fp]),
    "mapn_get_cbuffer": (C.c_int, [_ctx, C.POINTER(C.c_uint32 * 4), C.POINTER(C.c_float * 4)]),
    "mapn_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "mapn_comm_init": (C.c_int, [_ctx, C.c_void_p]),
    "mapn_set_gather_algorithm": (C.c_int, [_ctx, C.c_int]),
    "mapn_p2p_export": (C.c_int, [_ctx, C.c_void_p]),
    "mapn_p2p_import": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "mapn_p2p_status": (C.c_int, [_ctx]),
    "mapn_set_external_gather": (C.c_int, [_ctx, C.c_int]),
    "mapn_replica_checksum": (C.c_int, [_ctx, C.POINTER(C.c_uint64 * 2)]),
    "mapn_shard_range": (C.c_int, [_ctx, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mapn_get_device_info": (C.c_int, [C.c_int, C.POINTER(DeviceInfo)]),
    "mapn_device_count": (C.c_int, []),
    "mapn_get_kernel_stats": (C.c_int, [_ctx, C.c_int, C.POINTER(KernelStats)]),
    "mapn_get_step_samples": (C.c_int, [_ctx, C.POINTER(C.c_uint32), _fp, _fp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mapn_set_force_plan": (C.c_int, [_ctx, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "mapn_set_shard_overlap": (C.c_int, [_ctx, C.c_int]),
    "mapn_step_form_describe": (C.c_int, [C.c_uint32, C.c_int32]),
    "mapn_shard_describe": (C.c_int, [C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(ShardInfo)]),
    "mapn_shard_split_describe": (C.c_int, [C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(ShardSplitInfo)]),
    "mapn_calibrate_sym_xcds": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_uint32 * 8)]),
    "mapn_set_sym_xcd_weights": (C.c_int, [_ctx, C.POINTER(C.c_uint32 * 8)]),
    "mapn_sym_plan_describe": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32 * 8), C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(SymPlanInfo), C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32), C.c_uint64]),
    "mapn_get_sym_plan": (C.c_int, [_ctx, C.POINTER(SymPlanInfo), C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32), C.c_uint64]),
    "mapn_get_split_plan": (C.c_int, [_ctx, C.POINTER(SplitInfo), C.POINTER(SymPlanInfo), C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32), C.c_uint64]),
    "mapn_set_sym_plan": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mapn_measure_clock": (C.c_int, [_ctx, C.c_int, C.POINTER(ClockInfo)]),
    "mapn_set_timers": (C.c_int, [_ctx, C.c_int]),
    "mapn_compute_stream": (C.c_void_p, [_ctx]),
}

_lib = None


def library_path() -> str:
    return _SO


def build_library(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 into the in-tree libmapn.so (hipcc cross-compiles
    without a GPU)."""
    args = ["make", "-C", _CSRC, "-s"] + (["-B"] if force else [])
    subprocess.run(args, check=True)
    return _SO


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise MapnError(-3, f"{_SO} is missing: build it with `make -C {_CSRC}` "
                            "(__graft_entry__.build()); there is no Python/CPU fallback")
    lib = C.CDLL(_SO)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        raise MapnError(status, load_library().mapn_last_error().decode(errors="replace"))
