"""The consumer PROCESS of tests/test_ipc_consumer.py: Render::SetShared + CopySimulationResults
(Render.cpp:222-251, 789-831) across a process boundary.  Opens the compute process's exported
blob, and for each frame queues on its own stream: wait for the compute fence, copy the first
`ncopy` positions of that frame's buffer, signal the consumer fence.  Captured copies go to an npz.
Host-side HIP calls (malloc / memcpy) through ctypes; no torch, no oracle."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    blob_path, out_path = sys.argv[1], sys.argv[2]
    first_fence, first_index, frames, ncopy = (int(x) for x in sys.argv[3:7])
    delay = float(sys.argv[7])
    import mapn
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipDeviceSynchronize.argtypes = []
    blob = open(blob_path, "rb").read()
    time.sleep(delay)                       # a late consumer: the compute process must park, not run ahead
    with mapn.IpcView(blob, device=0) as view:
        cap = C.c_void_p()
        assert hip.hipMalloc(C.byref(cap), frames * ncopy * 16) == 0
        for f in range(frames):
            fence = first_fence + f
            index = (first_index + f) & 1
            view.copy_positions_async(index, ncopy, cap.value + f * ncopy * 16, wait_fence_value=fence)
            view.consumer_signal(fence)                      # Render.cpp:826 Signal(copyFence)
            if f == frames // 2:
                time.sleep(delay / 2)                        # stall in the middle as well
        assert hip.hipDeviceSynchronize() == 0
        got = np.empty((frames, ncopy, 4), np.float32)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), cap, got.nbytes, 2) == 0
        latest = view.latest()
    np.savez(out_path, got=got, latest=np.array(latest, np.uint64))


if __name__ == "__main__":
    main()
