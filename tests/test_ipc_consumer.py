"""SURVEY 8(f) row 2 across a PROCESS boundary: a renderer / analysis process on
the same GPU attaches to the compute process's position heap through hipIpc and pulls numCopy x 16 B
per step (Render.cpp:789-831) under the reference's fence protocol: its copy waits for the compute
fence (Render.cpp:796), the next Simulate waits for its signal (Compute.cpp:1012).  The consumer
starts late and stalls once; the compute process never host-synchronises inside the loop.  Every
captured copy must equal the lockstep reference's state of THAT frame, bit for bit -- a compute
side that ran ahead would hand the consumer a later frame."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mapn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_consumer_process_pulls_every_frame_under_the_fence_protocol(tmp_path):
    n, ncopy, frames = 8192, 3000, 24
    blob_path, out_path = str(tmp_path / "ipc.blob"), str(tmp_path / "captured.npz")
    with mapn.Compute(n, mass=70000.0 / n) as c, mapn.Compute(n, mass=70000.0 / n) as ref:
        open(blob_path, "wb").write(c.ipc_export())
        first_fence, first_index = c.GetFenceValue(), c.buffer_index
        worker = os.path.join(ROOT, "tests", "ipc_consumer_worker.py")
        proc = subprocess.Popen([sys.executable, worker, blob_path, out_path, str(first_fence), str(first_index),
                                 str(frames), str(ncopy), "0.6"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        c.ConsumerSignal(first_fence - 1)                 # nothing to wait for before frame 0
        want = np.empty((frames, ncopy, 4), np.float32)
        for f in range(frames):
            fence = c.GetFenceValue()                     # Particles.cpp:446
            c.Simulate(n, fence)                          # :448 -- parks on the device until the consumer signalled fence - 1
            ref.Simulate(n, 0)
            want[f] = ref.download_state()[0][:ncopy]
        out = proc.communicate(timeout=300)[0]
        assert proc.returncode == 0, out
        c.WaitForGpu()
        np.testing.assert_array_equal(c.download_state()[0], ref.download_state()[0])
    cap = np.load(out_path)
    bad = [f for f in range(frames) if not np.array_equal(cap["got"][f], want[f])]
    assert not bad, f"frames whose copy differs from the lockstep reference: {bad}"
    assert int(cap["latest"][0]) == first_fence + frames - 1
