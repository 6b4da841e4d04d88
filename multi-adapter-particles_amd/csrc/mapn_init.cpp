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
//
// The reference carries three variants of this loop behind #if (USE_ORIG / USE_SCALAR_OPTIMIZED /
// USE_SIMD_OPTIMIZED, :581-583); they share the distribution and differ in the random source.
// All three are selectable (mapn_config.init_variant), each seeded per body with the same
// lcg = fmix32(seed * 0x9E3779B9 + i + 1):
//   MAPN_INIT_LCG  (0, default)  the scalar LCG above                         (:711-749)
//   MAPN_INIT_SSE  (1)  rand_sse's four LCG lanes, x/y/z from lanes 0/1/2; one 4-lane draw per
//                       do-while round, at least one round               (:619-661, :751-793)
//   MAPN_INIT_MT   (2)  std::mt19937(lcg) + uniform_real_distribution<float>(-1,1) with the
//                       libstdc++ mapping value = float(u32) / 2^32 * 2 - 1 (a result >= 1
//                       clamps to the float below 1), components drawn in x, y, z order (:686-708)
// pos.w = 0 in every variant.
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

// Compute.cpp:619-661: four independent LCG lanes
struct SseLcg {
    uint32_t s[4];
    explicit SseLcg(uint32_t seed) : s{seed + 1u, seed, seed + 1u, seed} {}
    void draw(float out[3])
    {
        static const uint32_t mult[4] = {214013u, 17405u, 214013u, 69069u};
        static const uint32_t gadd[4] = {2531011u, 10395331u, 13737667u, 1u};
        const float k_scale = (1.0f / 32767.0f) * 2.0f;
        for (int l = 0; l < 4; l++) s[l] = s[l] * mult[l] + gadd[l];
        for (int l = 0; l < 3; l++) {
            const int r = (static_cast<int32_t>(s[l]) >> 16) & 0x7FFF;
            out[l] = static_cast<float>(r) * k_scale - 1.0f;
        }
    }
};

// the 32-bit Mersenne Twister of std::mt19937, restated (MT19937, Matsumoto & Nishimura)
struct Mt19937 {
    uint32_t mt[624];
    int idx;
    explicit Mt19937(uint32_t seed)
    {
        mt[0] = seed;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + static_cast<uint32_t>(i);
        idx = 624;
    }
    uint32_t next()
    {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) {
                const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    float uniform_m1_p1()
    {
        float c = static_cast<float>(next()) / 4294967296.0f;
        if (c >= 1.0f) c = std::nextafter(1.0f, 0.0f);
        return 2.0f * c + -1.0f;
    }
};

inline float len3(float x, float y, float z)
{
    float l = x * x + y * y;
    l = l + z * z;
    return std::sqrt(l);
}

void generate_range(int variant, uint32_t seed, uint32_t half, float spread, float speed, uint32_t b0,
                    uint32_t b1, float *pos4, float *vel3)
{
    const float centre = spread * 0.750f;
    for (uint32_t i = b0; i < b1; i++) {
        const uint32_t body_seed = fmix32(seed * 0x9E3779B9u + i + 1u);
        float dx, dy, dz;
        if (variant == MAPN_INIT_SSE) {
            SseLcg g(body_seed);
            dx = dy = dz = 0.0f;
            float l;
            do {                                                  // Compute.cpp:765-777
                float r[3];
                g.draw(r);
                dx = dx + r[0]; dy = dy + r[1]; dz = dz + r[2];
                l = dx * dx + dy * dy;
                l = l + dz * dz;
            } while (l < 10.0f);
        } else if (variant == MAPN_INIT_MT) {
            Mt19937 g(body_seed);
            dx = g.uniform_m1_p1(); dy = g.uniform_m1_p1(); dz = g.uniform_m1_p1();    // Compute.cpp:690
            for (;;) {
                float l = dx * dx + dy * dy;
                l = l + dz * dz;
                if (!(l < 10.0f)) break;                          // :691
                const float x = g.uniform_m1_p1(), y = g.uniform_m1_p1(), z = g.uniform_m1_p1();
                dx = dx + x; dy = dy + y; dz = dz + z;            // :693
            }
        } else {
            Lcg g{body_seed};
            dx = g.draw(); dy = g.draw(); dz = g.draw();
            for (;;) {
                float l = dx * dx + dy * dy;
                l = l + dz * dz;
                if (!(l < 10.0f)) break;
                const float x = g.draw(), y = g.draw(), z = g.draw();
                dx = dx + x; dy = dy + y; dz = dz + z;
            }
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
    return mapn_generate_initial_state_ex(MAPN_INIT_LCG, seed, n, spread, initial_speed, pos4, vel3);
}

extern "C" int mapn_generate_initial_state_ex(int variant, uint32_t seed, uint32_t n, float spread,
                                              float initial_speed, float *pos4, float *vel3)
{
    if (!pos4 || !vel3) return MAPN_ERR_INVALID_ARGUMENT;
    if (variant < MAPN_INIT_LCG || variant > MAPN_INIT_MT) return MAPN_ERR_INVALID_ARGUMENT;
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
        if (b0 < b1) th.emplace_back(generate_range, variant, seed, half, spread, initial_speed, b0, b1, pos4, vel3);
    }
    generate_range(variant, seed, half, spread, initial_speed, 0, std::min(total, per), pos4, vel3);
    for (auto &t : th) t.join();
    return MAPN_OK;
}
