// mapn_init.cpp -- deterministic initial-condition generator (host side, no device needed).
//
// Replaces LoadParticles / InitializeParticles, reference/Particles/Compute.cpp:667-812 and
// :820-844.  The reference draws from mt19937 seeded by random_device and shared without
// synchronisation across parallel_for threads (:678-684), so its initial state differs from
// run to run; "identical initial conditions" for the device path and the CPU oracle need a
// fully specified generator.  Specification (every op a separately rounded binary32 op; this
// file is built with -ffp-contract=off):
//
//   half = N / 2; bodies [0, half) surround (+0.75*spread, 0, 0), [half, 2*half) surround
//   (-0.75*spread, 0, 0)  (Compute.cpp:831-844); a leftover body of an odd N stays zero.
//   per body i (global index): lcg = fmix32(seed * 0x9E3779B9 + i + 1)      [murmur3 finaliser]
//   draw()  : lcg = 214013*lcg + 2531011; r = (lcg >> 16) & 0x7FFF          (fast_rand, :605-609)
//             value = float(r) * ((1/32767) * 2) - 1                         (:721-725, RAND_MAX 32767)
//   delta = (draw, draw, draw); while |delta|^2 < 10: delta += (draw, draw, draw)   (:727-736)
//   pos = centre + delta / |delta| * spread;  pos.w = 0                      (:738-743)
//   dir = pos.xyz / |pos.xyz|                  (XMVector3NormalizeEst at :746 -> exact normalise)
//   perp = ((1,1,1) - dir) / |(1,1,1) - dir|   (:747)
//   vel = cross(dir, perp) * speed             (:748)
//
// Generation is per-body independent, so it runs on all host cores like the reference's
// parallel_for (:684) and the result does not depend on the thread count.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "mapn.h"

namespace {

inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

struct Lcg {
    uint32_t s;
    float draw()
    {
        s = 214013u * s + 2531011u;
        const float k_scale = (1.0f / 32767.0f) * 2.0f;
        float v = static_cast<float>(static_cast<int>((s >> 16) & 0x7FFFu)) * k_scale;
        return v - 1.0f;
    }
};

inline float len3(float x, float y, float z)
{
    float l = x * x + y * y;
    l = l + z * z;
    return std::sqrt(l);
}

void generate_range(uint32_t seed, uint32_t half, float spread, float speed, uint32_t b0, uint32_t b1,
                    float *pos4, float *vel3)
{
    const float centre = spread * 0.750f;
    for (uint32_t i = b0; i < b1; i++) {
        Lcg g{fmix32(seed * 0x9E3779B9u + i + 1u)};
        float dx = g.draw(), dy = g.draw(), dz = g.draw();
        for (;;) {
            float l = dx * dx + dy * dy;
            l = l + dz * dz;
            if (!(l < 10.0f)) break;
            const float x = g.draw(), y = g.draw(), z = g.draw();
            dx = dx + x; dy = dy + y; dz = dz + z;
        }
        const float dl = len3(dx, dy, dz);
        const float cx = i < half ? centre : -centre;
        const float px = cx + dx / dl * spread;
        const float py = 0.0f + dy / dl * spread;
        const float pz = 0.0f + dz / dl * spread;
        float *p = pos4 + 4 * static_cast<size_t>(i);
        p[0] = px; p[1] = py; p[2] = pz; p[3] = 0.0f;

        const float pl = len3(px, py, pz);
        const float ux = px / pl, uy = py / pl, uz = pz / pl;
        float qx = 1.0f - ux, qy = 1.0f - uy, qz = 1.0f - uz;
        const float ql = len3(qx, qy, qz);
        qx = qx / ql; qy = qy / ql; qz = qz / ql;
        float *v = vel3 + 3 * static_cast<size_t>(i);
        v[0] = (uy * qz - uz * qy) * speed;
        v[1] = (uz * qx - ux * qz) * speed;
        v[2] = (ux * qy - uy * qx) * speed;
    }
}

}  // namespace

extern "C" int mapn_generate_initial_state(uint32_t seed, uint32_t n, float spread, float initial_speed,
                                           float *pos4, float *vel3)
{
    if (!pos4 || !vel3) return MAPN_ERR_INVALID_ARGUMENT;
    std::memset(pos4, 0, static_cast<size_t>(n) * 16);
    std::memset(vel3, 0, static_cast<size_t>(n) * 12);
    const uint32_t half = n / 2, total = 2 * half;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (total < 65536u) nt = 1;
    std::vector<std::thread> th;
    const uint32_t per = (total + nt - 1) / nt;
    for (unsigned t = 1; t < nt; t++) {
        const uint32_t b0 = std::min(total, t * per), b1 = std::min(total, (t + 1) * per);
        if (b0 < b1) th.emplace_back(generate_range, seed, half, spread, initial_speed, b0, b1, pos4, vel3);
    }
    generate_range(seed, half, spread, initial_speed, 0, std::min(total, per), pos4, vel3);
    for (auto &t : th) t.join();
    return MAPN_OK;
}
