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
    # the files come in the stated order ...
    seen = [f for k, f in enumerate(files) if k == 0 or files[k - 1] != f]
    assert seen[:len(first)] == first, seen
    # ... and whatever carries the `timing` marker (on the test or as its file's pytestmark) is exactly the TAIL of the run: once the first
    # timing-asserting test has come, nothing else follows
    t = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", "gpu and timing"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    timing = [ln for ln in t.stdout.splitlines() if "::" in ln]
    assert len(timing) >= 14 and any("test_bench_json_contract" in i for i in timing), timing
    assert ids[len(ids) - len(timing):] == timing, (ids[len(ids) - len(timing):], timing)
    assert not set(ids[:len(ids) - len(timing)]) & set(timing)
    # the tests known to assert a time are among them (a new one must carry the marker: the names below are the ones that did when the order was introduced)
    for i in ids:
        if "faster_than" in i or "wins_its_a_b" in i or i.split("::")[0].endswith(("test_bench_contract.py", "test_gpu_mfma_ab.py")):
            assert i in timing, i


def test_the_last_recorded_gpu_suite_ran_in_well_under_its_time_limit():
    """`pytest -m gpu` is killed at 1200 s, and a kill is a red round whatever the tests say; the suite had grown 30 -> 349 ->
    461 -> 480 -> 686 s.  The latest `pytest -m gpu --durations` record committed under profiles/ (one per round from round 6 on, made on
    the GPU box by tools/evidence.sh suite full) must show a total of at most 600 s -- whoever adds a minute of tests sees it here, on the
    CPU, before the driver's box does -- and must be a record of a green run."""
    import glob
    import re
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pytest_gpu_durations.txt")))
    assert recs, "no profiles/rNN_pytest_gpu_durations.txt on record"
    text = open(recs[-1]).read()
    m = re.findall(r"(\d+) passed(?:, (\d+) skipped)?(?:, \d+ deselected)? in ([0-9.]+)s", text)
    assert m and " failed" not in text.splitlines()[-1], (recs[-1], text.splitlines()[-1:])
    passed, total = int(m[-1][0]), float(m[-1][2])
    assert passed >= 230 and total <= 600.0, f"{recs[-1]}: {passed} passed in {total:.0f} s (limit of this check 600 s; the driver's 1200 s)"
    assert "slowest" in text           # (the --durations table is part of the record: it says where the time goes)
