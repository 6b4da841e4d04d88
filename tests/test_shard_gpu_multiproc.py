"""Two real processes sharing ONE GPU, each owning half of the bodies (world_size 2): every rank
advances its slice on the device through the C ABI, the slices are exchanged with a gloo
all-gather (RCCL refuses two ranks on one device, so the in-library RCCL exchange itself is
covered by the 1-rank test in test_gpu_parity.py), and the composed trajectory is compared with
the CPU oracle.  Covers the per-rank slicing and velocity locality of the sharded device step in
a true one-process-per-rank setting.  The ranks are separate interpreter processes
(tests/shard_gpu_worker.py); this process never imports torch, so it keeps a single HIP runtime."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_processes_one_gpu_sharded_device_steps(tmp_path, oracle):
    from oracle import OracleSim, Params
    n, steps, world = 4096, 3, 2
    port = 29600 + (os.getpid() % 2000)
    worker = os.path.join(ROOT, "tests", "shard_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(n), str(steps), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    got = np.load(os.path.join(str(tmp_path), "gpu_sharded.npz"))
    pos0, vel0 = oracle.initial_state(n, seed=1)
    sim = OracleSim(oracle, pos0, vel0, params=Params(mass=70000.0 / n))
    sim.simulate(steps=steps)
    dx = np.linalg.norm(got["pos"][:, :3].astype(np.float64) - sim.latest[0][:, :3], axis=1).max() / 400.0
    dv = np.linalg.norm(got["vel"].astype(np.float64) - sim.latest[1], axis=1).max() / 15.0
    assert dx < 3e-6 and dv < 6e-5, (dx, dv)
