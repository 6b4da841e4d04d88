#!/usr/bin/env python3
"""Per-wave timeline of ONE sharded symmetric force launch in loopback (development tool).

MAPN_P2P_LOOPBACK=1 maps every peer of rank R of a P-way job to the rank itself, so the real kernels run at
the true shard size on a 1-GPU box; MAPN_STAMP_DUMP makes a stamped diagnostic launch (mapn_measure_clock)
record every wave's entry / loop start / loop end / exit (s_memrealtime, 100 MHz) and its placement.
Prints where the launch's time goes: ramp-up, the spread of loop lengths, the tail, per-SIMD pairing.

Usage: tools/shard_timeline.py [bodies] [world] [rank] [algo]      (algo 4 = sharded symmetric, 0 = unsharded symmetric)
"""
import os
import struct
import sys
import time

import numpy as np

DUMP = "/tmp/mapn_timeline.bin"
os.environ["MAPN_TEST_HOOKS"] = "1"; os.environ["MAPN_P2P_LOOPBACK"] = "1"
os.environ["MAPN_STAMP_DUMP"] = DUMP        # (an experiment switch: honoured with MAPN_TEST_HOOKS=1, set above)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapn  # noqa: E402


def analyse(path, label):
    raw = open(path, "rb").read()
    nw, parts, waves, rank = struct.unpack("<4Q", raw[:32])
    t = np.frombuffer(raw[32:32 + nw * 48], dtype=np.uint64).reshape(nw, 6)
    xt = np.frombuffer(raw[32 + nw * 48:], dtype=np.uint64)
    xt = xt.reshape(-1, 8).astype(np.int64) if xt.size else None
    ok = t[:, 0] > 0
    t = t[ok]
    entry, loop, done, exit_ = (t[:, k].astype(np.int64) for k in range(4))
    hw = t[:, 4]
    steps = t[:, 5].astype(np.int64)
    t0 = entry.min()
    us = lambda x: (x - t0) / 100.0
    simd = (hw & 0x30) >> 4
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    xcc = (hw >> 32) & 0xf
    slot = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
    pct = lambda a: " ".join(f"{np.percentile(a, p):7.2f}" for p in (0, 10, 50, 90, 100))
    print(f"== {label}: {len(t)} waves (parts {parts}, {waves} waves per workgroup, rank {rank}); times in us after the first wave's entry; min p10 p50 p90 max")
    print(f"  entry            {pct(us(entry))}")
    print(f"  loop start       {pct(us(loop))}      prologue (entry -> loop) {pct((loop - entry) / 100.0)}")
    print(f"  loop end         {pct(us(done))}      loop length              {pct((done - loop) / 100.0)}")
    print(f"  exit             {pct(us(exit_))}      epilogue (loop end -> exit) {pct((exit_ - done) / 100.0)}")
    for s in sorted(set(steps)):
        m = steps == s
        print(f"  waves of {s:4d} steps: {m.sum():5d}   loop length {pct((done[m] - loop[m]) / 100.0)}   us per 64 steps p50 {np.median((done[m] - loop[m]) / 100.0) * 64 / s:6.2f}")
    # per SIMD: how many waves, when the last leaves
    order = np.argsort(slot)
    uniq, idx, cnt = np.unique(slot[order], return_index=True, return_counts=True)
    last = np.array([us(exit_[order[i:i + c]]).max() for i, c in zip(idx, cnt)])
    work = np.array([steps[order[i:i + c]].sum() for i, c in zip(idx, cnt)])
    print(f"  SIMDs used {len(uniq)}; waves per SIMD {np.bincount(cnt)[1:].tolist()} (1, 2, ... waves); last exit per SIMD {pct(last)}")
    for wk in sorted(set(work)):
        m = work == wk
        print(f"    SIMDs with {wk:4d} steps in total: {m.sum():5d}, last exit {pct(last[m])}")
    print(f"  launch as the waves saw it: {us(exit_).max():.2f} us from first entry to last exit")
    # per XCD: do the eight dies run at one speed?  (loop time per step of the waves that ran on it, and when its last wave left)
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        per_step = (done[m] - loop[m]) / 100.0 / np.maximum(steps[m], 1)
        print(f"    XCD {x}: {m.sum():5d} waves, loop us per step p50 {np.median(per_step):.4f}, last exit {us(exit_[m]).max():8.2f}, steps {steps[m].sum()}")
    if xt is not None:
        xt = xt[xt[:, 0] > 0]
        if len(xt):
            names = ["entry", "sends issued", "sends acknowledged + own rows summed", "peers' rows arrived (counters)", "integrated, positions stored",
                     "position stores acknowledged", "counter published / exit"]
            x0 = xt[:, 0].min()
            print(f"  exchange launch: {len(xt)} workgroups; first entry {(x0 - t0) / 100.0:.2f} us after the force launch's first wave, "
                  f"{(x0 - exit_.max()) / 100.0:.2f} us after its last exit; stamps in us after that entry; min p10 p50 p90 max")
            for k, nm in enumerate(names):
                col = xt[:, k]
                col = col[col > 0]
                if len(col):
                    print(f"    {nm:40s} {pct((col - x0) / 100.0)}")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    algo = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    kw = dict(rank=rank, world_size=world) if algo else {}
    with mapn.Compute(n, device=0, mass=70000.0 / n, **kw) as c:
        if algo:
            blob = c.p2p_export()
            c.p2p_import([blob] * world)
            c.set_gather_algorithm(algo)
        if os.environ.get("TL_XCD") == "1":                # XCD-aware parts: calibrate on this box, weigh the parts
            w = c.calibrate_sym_xcds(4)
            c.set_sym_xcd_weights(w)
            print(f"# XCD weights {w}")
        pl = c.sym_plan()
        print(f"# plan: {pl.waves} waves x {pl.parts} parts, wave bias {pl.wave_bias}, sets {pl.sets}")
        c.set_timers(4)
        for _ in range(200):
            c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu()
        c.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(400):
            c.Simulate(n, c.GetFenceValue())
        c.WaitForGpu()
        dt = (time.perf_counter() - t0) / 400
        st = c.kernel_stats()
        print(f"# {n} bodies, world {world}, rank {rank}, algo {algo}: step {dt * 1e6:.1f} us, force kernel {st.avg_seconds * 1e6:.1f} us ({st.kernel_name.decode()} grid {st.grid_x} x {st.grid_y})")
        ck = c.measure_clock(1)
        print(f"# held clock {ck.shader_clock_ghz:.3f} GHz (p10 {ck.shader_clock_ghz_p10:.3f}, p90 {ck.shader_clock_ghz_p90:.3f}), median wave cycles {ck.median_wave_cycles:.0f}")
    analyse(DUMP, f"{n} / {world} rank {rank} algo {algo}")


if __name__ == "__main__":
    main()
