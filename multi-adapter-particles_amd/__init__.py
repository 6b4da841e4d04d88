"""mapn -- MI355X-native n-body compute step (host-side Python mirror of the C ABI).

The product is ``libmapn.so`` (HIP kernels for gfx950 + the C ABI of ``include/mapn.h``); this
package is a thin ctypes binding that mirrors the reference's ``class Compute``
(reference/Particles/Compute.h:33-78) so tests and the bench harness read like its caller,
``Particles::Draw`` (Particles.cpp:432-456).  There is no CPU fallback: without the built
library, or without a gfx950 device, construction raises.
"""
from ._lib import (  # noqa: F401
    FORCE_ALL_PAIRS, FORCE_CENTRAL_WELL, KERNEL_AUTO, KERNEL_LDS, KERNEL_SCALAR, KERNEL_SYMMETRIC,
    FLAG_NO_INIT, FLAG_SHARD_OVERLAP, FLAG_STRICT_CONSUMER, FLAG_USE_GRAPH, FLAG_XCD_CALIBRATE, INIT_LCG, INIT_MT, INIT_SSE, Config, DeviceInfo, KernelStats, MapnError, SharedHandles,
    build_library, library_path, load_library,
)
from .compute import Compute, IpcView, SymPlan, describe_sym_plan, generate_initial_state  # noqa: F401

__all__ = [
    "Compute", "IpcView", "Config", "MapnError",
    "generate_initial_state", "SymPlan", "describe_sym_plan", "build_library", "load_library", "library_path",
    "FORCE_ALL_PAIRS", "FORCE_CENTRAL_WELL", "KERNEL_AUTO", "KERNEL_LDS", "KERNEL_SCALAR", "KERNEL_SYMMETRIC",
    "FLAG_NO_INIT", "FLAG_USE_GRAPH", "FLAG_SHARD_OVERLAP", "FLAG_STRICT_CONSUMER", "FLAG_XCD_CALIBRATE", "INIT_LCG", "INIT_SSE", "INIT_MT",
]
