"""The order of `pytest -m gpu` (tests/conftest.py): every oracle / golden comparison runs BEFORE anything that asserts a
time, a cycle count or a subprocess bench, so that with `-x` a noisy bound cannot leave the parity evidence untested."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_tests_are_collected_before_every_timing_asserting_test():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", "gpu"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    ids = [ln for ln in r.stdout.splitlines() if "::" in ln]
    assert len(ids) > 200, r.stdout[-500:] + r.stderr[-500:]
    files = [i.split("::")[0].split("/")[-1] for i in ids]
    first = ["test_gpu_parity.py", "test_gpu_sym.py", "test_gpu_partial_active.py", "test_parity_1000.py", "test_shard_gpu_multiproc.py", "test_cpp_compat.py", "test_ipc_consumer.py"]
    timing_files = {"test_bench_contract.py", "test_gpu_mfma_ab.py"}
    # the files come in the stated order ...
    seen = [f for k, f in enumerate(files) if k == 0 or files[k - 1] != f]
    assert seen[:len(first)] == first, seen
    # ... and once the first timing-asserting test has come, nothing but timing-asserting tests follows
    k0 = next(k for k, i in enumerate(ids) if files[k] in timing_files or "faster_than" in i or "wins_its_a_b" in i)
    tail = ids[k0:]
    assert all(files[k0 + k] in timing_files or "faster_than" in i or "wins_its_a_b" in i for k, i in enumerate(tail)), tail
    assert len(tail) >= 14 and any("test_bench_json_contract" in i for i in tail)
