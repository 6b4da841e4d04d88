import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import mapn, mapn.compute
real = mapn.Compute
class OneDev(real):
    def __init__(self, n, device=0, *a, **k):
        super().__init__(n, 0, *a, **k)          # every "device" is device 0: a dry run of the test's logic
mapn.compute.device_count = lambda: 2
mapn.Compute = OneDev
import test_gpu_parity as t
t.mapn.Compute = OneDev
t.test_create_from_across_two_devices_when_the_box_has_them()
print("dry run of the two-device CopyState test on one device: ok")
