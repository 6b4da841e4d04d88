"""Independent numpy models of the step (TEST INFRASTRUCTURE ONLY; second opinion on small N).

``accel_fp64``      float64 all-pairs accelerations from float32 positions (the "truth" the fp32
                    paths are measured against in the 1-step tables).
``step_fp32_loop``  float32 numpy restatement with the HLSL's exact operation order, vectorised
                    over i with a Python loop over j ascending -- written without looking at
                    mapn_oracle.c's loop structure, so agreement is bit-for-bit evidence that
                    the C oracle does what its header says.
Follows nBodyGravityCS.hlsl:44-57 (pair term), :92-101 (well), :103-108 (integrator).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def accel_fp64(pos, mass=70000.0, soft2=25.0, first=0, count=None, chunk=512):
    p = np.asarray(pos, np.float64)[:, :3]
    n = p.shape[0]
    count = n - first if count is None else count
    out = np.zeros((count, 3))
    for c0 in range(first, first + count, chunk):
        c1 = min(c0 + chunk, first + count)
        r = p[None, :, :] - p[c0:c1, None, :]
        d = (r * r).sum(-1) + soft2
        s = mass * d ** -1.5
        out[c0 - first:c1 - first] = (r * s[..., None]).sum(1)
    return out


def integrate_fp64(pos, vel, acc, dt=0.1, damping=1.0):
    v = (np.asarray(vel, np.float64) + acc * dt) * damping
    x = np.asarray(pos, np.float64)[:, :3] + v * dt
    w = np.sqrt((acc * acc).sum(-1))
    return np.concatenate([x, w[:, None]], 1), v


def _integrate_fp32(pos, vel, ax, ay, az, dt, damping):
    dt, damping = f32(dt), f32(damping)
    vx = (vel[:, 0] + ax * dt) * damping
    vy = (vel[:, 1] + ay * dt) * damping
    vz = (vel[:, 2] + az * dt) * damping
    npos = np.empty_like(pos)
    npos[:, 0] = pos[:, 0] + vx * dt
    npos[:, 1] = pos[:, 1] + vy * dt
    npos[:, 2] = pos[:, 2] + vz * dt
    npos[:, 3] = np.sqrt((ax * ax + ay * ay) + az * az)
    return npos, np.stack([vx, vy, vz], 1)


def step_fp32_loop(pos, vel, mass=70000.0, soft2=25.0, dt=0.1, damping=1.0):
    pos = np.asarray(pos, f32)
    vel = np.asarray(vel, f32)
    n = pos.shape[0]
    mass, soft2, one = f32(mass), f32(soft2), f32(1.0)
    xi, yi, zi = pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy()
    ax = np.zeros(n, f32); ay = np.zeros(n, f32); az = np.zeros(n, f32)
    for j in range(n):
        rx = pos[j, 0] - xi; ry = pos[j, 1] - yi; rz = pos[j, 2] - zi
        d = ((rx * rx + ry * ry) + rz * rz) + soft2
        inv = one / np.sqrt(d)
        s = (mass * ((inv * inv) * inv)) * one
        ax = ax + rx * s; ay = ay + ry * s; az = az + rz * s
    return _integrate_fp32(pos, vel, ax, ay, az, dt, damping)


def step_central_well_fp32(pos, vel, mass=70000.0, soft2=25.0, dt=0.1, damping=1.0):
    pos = np.asarray(pos, f32)
    vel = np.asarray(vel, f32)
    rx, ry, rz = pos[:, 0], pos[:, 1], pos[:, 2]
    d = ((rx * rx + ry * ry) + rz * rz) + f32(soft2)
    inv = f32(-1.0) / np.sqrt(d)
    s = f32(mass) * ((inv * inv) * inv)
    return _integrate_fp32(pos, vel, rx * s, ry * s, rz * s, dt, damping)
