#!/usr/bin/env python3
"""Numerical cost of the two MFMA recasts of the pair term (BASELINE configs[4], SURVEY 7 'MFMA
recast'), measured on the real two-shell state against the float64 per-body acceleration:

  r2-expansion   d_ij = (|x_i|^2 + soft^2) + |x_j|^2 - 2 x_i.x_j as a K = 4 contraction in fp32
                 (what v_mfma_f32_16x16x4_f32 computes: an fp32 fma chain per output element);
  sum-form       a_i = sum_j s_ij x_j - x_i sum_j s_ij (the N = 4 accumulation on MFMA) in fp32, with the
                 exact fp32 d_ij;
  both, and for reference the kernel's form (differences first, fp32) = what ships.

fp32 fma is emulated through float64 (product exact, one rounding to fp32) -- an error ESTIMATE of the
recasts, not a bit-exact MFMA model.  CPU only (numpy), test infrastructure / evidence, not product.
usage: mfma_recast_error.py [N=65536] [bodies checked=128]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapn  # noqa: E402  (host-side initial-state generator only; no device needed)

f32 = np.float32


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def recast_errors(n=65536, k=128):
    """{form: (median, max) of |da| / |a| over k bodies of the N-body two-shell state}, plus "d_rel": (median, 99.9 %, max) of the
    relative error of the expanded r^2 over all their pairs."""
    pos, _ = mapn.generate_initial_state(n, seed=1)
    x = pos[:, :3].astype(f32)
    soft2 = f32(25.0)
    idx = np.linspace(0, n - 1, k).astype(int)
    x64 = x.astype(np.float64)
    rows = {"kernel form (differences first, fp32)": [], "r2-expansion on MFMA, difference-form accumulate": [],
            "exact fp32 r2, sum-form accumulate on MFMA": [], "r2-expansion + sum-form (all on MFMA)": []}
    drel = []
    for i in idx:
        xi = x[i]
        r = x - xi                                              # fp32 differences
        d_exact = fma32(r[:, 2], r[:, 2], fma32(r[:, 1], r[:, 1], fma32(r[:, 0], r[:, 0], np.full(n, soft2, f32))))
        # r2 expansion: C_i = |x_i|^2 + soft^2 (fp32), K = 4: (x_i,y_i,z_i,1).(-2x_j,-2y_j,-2z_j,|x_j|^2)
        ci = fma32(xi[2:3], xi[2:3], fma32(xi[1:2], xi[1:2], fma32(xi[0:1], xi[0:1], np.array([soft2], f32))))
        nj = fma32(x[:, 2], x[:, 2], fma32(x[:, 1], x[:, 1], (x[:, 0] * x[:, 0]).astype(f32)))
        m2 = (f32(-2.0) * x).astype(f32)
        d_exp = np.full(n, ci[0], f32)
        for comp in range(3):
            d_exp = fma32(np.full(n, xi[comp], f32), m2[:, comp], d_exp)
        d_exp = fma32(np.ones(n, f32), nj, d_exp)
        d_exp = np.maximum(d_exp, f32(1e-3))                    # a cancelled d can go <= 0: clamp so rsq stays finite
        r64 = x64 - x64[i]
        d64 = (r64 * r64).sum(1) + 25.0
        a64 = (r64 * (d64 ** -1.5)[:, None]).sum(0)
        drel.append(np.abs(d_exp.astype(np.float64) - d64) / d64)

        def s_of(d):
            inv = (f32(1.0) / np.sqrt(d)).astype(f32)
            return ((inv * inv).astype(f32) * inv).astype(f32)

        def diff_form(s):
            return np.array([(r[:, c].astype(np.float64) * s.astype(np.float64)).astype(f32).astype(np.float64).sum() for c in range(3)])

        def sum_form(s):
            # fp32 running sums in chunks of 1024 (like the kernel's j-chunks), combined in fp32
            acc = np.zeros(4, f32)
            for c0 in range(0, n, 1024):
                sl = slice(c0, c0 + 1024)
                part = np.array([np.add.reduce((s[sl] * x[sl, c]).astype(f32), dtype=f32) for c in range(3)] + [np.add.reduce(s[sl], dtype=f32)], f32)
                acc = (acc + part).astype(f32)
            return (acc[:3] - (xi * acc[3]).astype(f32)).astype(np.float64)

        na = np.linalg.norm(a64)
        rows["kernel form (differences first, fp32)"].append(np.linalg.norm(diff_form(s_of(d_exact)) - a64) / na)
        rows["r2-expansion on MFMA, difference-form accumulate"].append(np.linalg.norm(diff_form(s_of(d_exp)) - a64) / na)
        rows["exact fp32 r2, sum-form accumulate on MFMA"].append(np.linalg.norm(sum_form(s_of(d_exact)) - a64) / na)
        rows["r2-expansion + sum-form (all on MFMA)"].append(np.linalg.norm(sum_form(s_of(d_exp)) - a64) / na)
    drel = np.concatenate(drel)
    out = {name: (float(np.median(v)), float(np.max(v))) for name, v in rows.items()}
    out["d_rel"] = (float(np.median(drel)), float(np.quantile(drel, 0.999)), float(drel.max()))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    res = recast_errors(n, k)
    drel = res.pop("d_rel")
    print(f"two-shell state, N = {n}, {k} bodies checked against the float64 acceleration; soft^2 = 25, |x| <= 700")
    print(f"r2-expansion: relative error of d_ij over all pairs  median {drel[0]:.2e}  99.9 % {drel[1]:.2e}  max {drel[2]:.2e}")
    print(f"{'form':58s} {'median |da|/|a|':>16s} {'max |da|/|a|':>14s}")
    for name, (med, mx) in res.items():
        print(f"{name:58s} {med:16.2e} {mx:14.2e}")


if __name__ == "__main__":
    main()
