// tools/sym_jpack_probe.hip -- experiment behind DESIGN 7 (round 4): what the lane moves of the symmetric pair loop cost when the
// board's power limit sets the clock.  Two register-resident loops, same pair term (14 packed operations + 2 v_rsq_f32 per packed
// evaluation), two waves per SIMD, random data, no memory traffic inside the loop:
//   A  the shipped arrangement: 16 resident bodies per lane as 8 packed pairs, ONE travelling body per lane;
//      a step = 8 packed evaluations (16 pairs per lane) + 9 ds_bpermute_b32 (3 position words, 6 unfolded reaction words)
//   B  the two lanes of a packed operation given to TWO travelling bodies and one resident one: 16 resident bodies per lane,
//      unpacked; a step = 16 packed evaluations (32 pairs per lane) + 12 ds_bpermute_b32 -- a third fewer moves per pair
// Prints pairs per second, wall time per launch and the clock held (s_memtime / s_memrealtime), after >= 2 s of launches.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-sched-strategy=max-ilp -o tools/sym_jpack_probe tools/sym_jpack_probe.hip   (the schedule the product has)      Run: tools/sym_jpack_probe [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float lane_next(float v, int addr)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

template <int MODE>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const float4 *bodies, float4 *out, unsigned long long *clk, int steps, float soft, int bias)
{
    const uint32_t lane = threadIdx.x & 63u, gw = (blockIdx.x * 512u + threadIdx.x) >> 6;
    const int next = (int)((lane + 1u) & 63u) * 4;
    const v2f soft2 = v2f{soft, soft};
    const float4 *mine = bodies + (size_t)gw * 64u * 20u + lane;
    // the product's wave bias: the older wave of every SIMD (the first half of the workgroup's waves) takes 10 : 3 of the steps, so that
    // the two end together (issue goes oldest-first: run with equal shares the older wave ends early and the younger one runs on alone)
    if (bias) steps = (threadIdx.x >> 6) < 4u ? steps * 20 / 13 : steps * 6 / 13;
    steps &= ~1;
    unsigned long long c0 = 0, r0 = 0;
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if constexpr (MODE == 0) {
        v2f xi[8], yi[8], zi[8], ax[8], ay[8], az[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 b0 = mine[(2 * k) * 64], b1 = mine[(2 * k + 1) * 64];
            xi[k] = v2f{b0.x, b1.x}; yi[k] = v2f{b0.y, b1.y}; zi[k] = v2f{b0.z, b1.z};
            ax[k] = v2f{0.f, 0.f}; ay[k] = v2f{0.f, 0.f}; az[k] = v2f{0.f, 0.f};
        }
        const float4 j = mine[16 * 64];
        float xj = j.x, yj = j.y, zj = j.z;
        v2f bx = v2f{0.f, 0.f}, by = v2f{0.f, 0.f}, bz = v2f{0.f, 0.f};
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(c0));
#pragma nounroll
        for (int t = 0; t < steps; t++) {
            const float x1 = lane_next(xj, next), y1 = lane_next(yj, next), z1 = lane_next(zj, next);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const v2f dx = xj - xi[k], dy = yj - yi[k], dz = zj - zi[k];
                v2f d = __builtin_elementwise_fma(dx, dx, soft2);
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                v2f inv; inv.x = __builtin_amdgcn_rsqf(d.x); inv.y = __builtin_amdgcn_rsqf(d.y);
                const v2f inv3 = inv * inv * inv;
                ax[k] = __builtin_elementwise_fma(dx, inv3, ax[k]); ay[k] = __builtin_elementwise_fma(dy, inv3, ay[k]); az[k] = __builtin_elementwise_fma(dz, inv3, az[k]);
                bx = __builtin_elementwise_fma(-dx, inv3, bx); by = __builtin_elementwise_fma(-dy, inv3, by); bz = __builtin_elementwise_fma(-dz, inv3, bz);
            }
            bx.x = lane_next(bx.x, next); by.x = lane_next(by.x, next); bz.x = lane_next(bz.x, next);
            bx.y = lane_next(bx.y, next); by.y = lane_next(by.y, next); bz.y = lane_next(bz.y, next);
            xj = x1; yj = y1; zj = z1;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { ox += ax[k].x + ax[k].y; oy += ay[k].x + ay[k].y; oz += az[k].x + az[k].y; }
        ox += bx.x + bx.y; oy += by.x + by.y; oz += bz.x + bz.y;
    } else if constexpr (MODE == 2) {
        // A as the product runs it: two steps per trip, the travelling position alternating between two register sets (no copies)
        v2f xi[8], yi[8], zi[8], ax[8], ay[8], az[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 b0 = mine[(2 * k) * 64], b1 = mine[(2 * k + 1) * 64];
            xi[k] = v2f{b0.x, b1.x}; yi[k] = v2f{b0.y, b1.y}; zi[k] = v2f{b0.z, b1.z};
            ax[k] = v2f{0.f, 0.f}; ay[k] = v2f{0.f, 0.f}; az[k] = v2f{0.f, 0.f};
        }
        const float4 j = mine[16 * 64];
        float xj = j.x, yj = j.y, zj = j.z;
        v2f bx = v2f{0.f, 0.f}, by = v2f{0.f, 0.f}, bz = v2f{0.f, 0.f};
        auto step = [&](float x, float y, float z) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const v2f dx = x - xi[k], dy = y - yi[k], dz = z - zi[k];
                v2f d = __builtin_elementwise_fma(dx, dx, soft2);
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                v2f inv; inv.x = __builtin_amdgcn_rsqf(d.x); inv.y = __builtin_amdgcn_rsqf(d.y);
                const v2f inv3 = inv * inv * inv;
                ax[k] = __builtin_elementwise_fma(dx, inv3, ax[k]); ay[k] = __builtin_elementwise_fma(dy, inv3, ay[k]); az[k] = __builtin_elementwise_fma(dz, inv3, az[k]);
                bx = __builtin_elementwise_fma(-dx, inv3, bx); by = __builtin_elementwise_fma(-dy, inv3, by); bz = __builtin_elementwise_fma(-dz, inv3, bz);
            }
            bx.x = lane_next(bx.x, next); by.x = lane_next(by.x, next); bz.x = lane_next(bz.x, next);
            bx.y = lane_next(bx.y, next); by.y = lane_next(by.y, next); bz.y = lane_next(bz.y, next);
        };
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(c0));
#pragma nounroll
        for (int t = 0; t + 2 <= steps; t += 2) {
            const float x1 = lane_next(xj, next), y1 = lane_next(yj, next), z1 = lane_next(zj, next);
            step(xj, yj, zj);
            xj = lane_next(x1, next); yj = lane_next(y1, next); zj = lane_next(z1, next);
            step(x1, y1, z1);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { ox += ax[k].x + ax[k].y; oy += ay[k].x + ay[k].y; oz += az[k].x + az[k].y; }
        ox += bx.x + bx.y; oy += by.x + by.y; oz += bz.x + bz.y;
    } else if constexpr (MODE == 3) {
        // B, two steps per trip as well
        float xi[16], yi[16], zi[16];
        v2f ax[16], ay[16], az[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float4 b0 = mine[k * 64];
            xi[k] = b0.x; yi[k] = b0.y; zi[k] = b0.z;
            ax[k] = v2f{0.f, 0.f}; ay[k] = v2f{0.f, 0.f}; az[k] = v2f{0.f, 0.f};
        }
        const float4 ja = mine[16 * 64], jb = mine[17 * 64];
        v2f xj = v2f{ja.x, jb.x}, yj = v2f{ja.y, jb.y}, zj = v2f{ja.z, jb.z};
        v2f bx = v2f{0.f, 0.f}, by = v2f{0.f, 0.f}, bz = v2f{0.f, 0.f};
        auto step = [&](v2f x, v2f y, v2f z) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const v2f dx = x - xi[k], dy = y - yi[k], dz = z - zi[k];
                v2f d = __builtin_elementwise_fma(dx, dx, soft2);
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                v2f inv; inv.x = __builtin_amdgcn_rsqf(d.x); inv.y = __builtin_amdgcn_rsqf(d.y);
                const v2f inv3 = inv * inv * inv;
                ax[k] = __builtin_elementwise_fma(dx, inv3, ax[k]); ay[k] = __builtin_elementwise_fma(dy, inv3, ay[k]); az[k] = __builtin_elementwise_fma(dz, inv3, az[k]);
                bx = __builtin_elementwise_fma(-dx, inv3, bx); by = __builtin_elementwise_fma(-dy, inv3, by); bz = __builtin_elementwise_fma(-dz, inv3, bz);
            }
            bx.x = lane_next(bx.x, next); by.x = lane_next(by.x, next); bz.x = lane_next(bz.x, next);
            bx.y = lane_next(bx.y, next); by.y = lane_next(by.y, next); bz.y = lane_next(bz.y, next);
        };
        auto rot = [&](v2f v) { v2f r; r.x = lane_next(v.x, next); r.y = lane_next(v.y, next); return r; };
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(c0));
#pragma nounroll
        for (int t = 0; t + 2 <= steps; t += 2) {
            const v2f x1 = rot(xj), y1 = rot(yj), z1 = rot(zj);
            step(xj, yj, zj);
            xj = rot(x1); yj = rot(y1); zj = rot(z1);
            step(x1, y1, z1);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) { ox += ax[k].x + ax[k].y; oy += ay[k].x + ay[k].y; oz += az[k].x + az[k].y; }
        ox += bx.x + bx.y; oy += by.x + by.y; oz += bz.x + bz.y;
    } else {
        float xi[16], yi[16], zi[16];
        v2f ax[16], ay[16], az[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float4 b0 = mine[k * 64];
            xi[k] = b0.x; yi[k] = b0.y; zi[k] = b0.z;
            ax[k] = v2f{0.f, 0.f}; ay[k] = v2f{0.f, 0.f}; az[k] = v2f{0.f, 0.f};
        }
        const float4 ja = mine[16 * 64], jb = mine[17 * 64];
        v2f xj = v2f{ja.x, jb.x}, yj = v2f{ja.y, jb.y}, zj = v2f{ja.z, jb.z};
        v2f bx = v2f{0.f, 0.f}, by = v2f{0.f, 0.f}, bz = v2f{0.f, 0.f};
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(c0));
#pragma nounroll
        for (int t = 0; t < steps; t++) {
            v2f x1, y1, z1;
            x1.x = lane_next(xj.x, next); y1.x = lane_next(yj.x, next); z1.x = lane_next(zj.x, next);
            x1.y = lane_next(xj.y, next); y1.y = lane_next(yj.y, next); z1.y = lane_next(zj.y, next);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const v2f dx = xj - xi[k], dy = yj - yi[k], dz = zj - zi[k];
                v2f d = __builtin_elementwise_fma(dx, dx, soft2);
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                v2f inv; inv.x = __builtin_amdgcn_rsqf(d.x); inv.y = __builtin_amdgcn_rsqf(d.y);
                const v2f inv3 = inv * inv * inv;
                ax[k] = __builtin_elementwise_fma(dx, inv3, ax[k]); ay[k] = __builtin_elementwise_fma(dy, inv3, ay[k]); az[k] = __builtin_elementwise_fma(dz, inv3, az[k]);
                bx = __builtin_elementwise_fma(-dx, inv3, bx); by = __builtin_elementwise_fma(-dy, inv3, by); bz = __builtin_elementwise_fma(-dz, inv3, bz);
            }
            bx.x = lane_next(bx.x, next); by.x = lane_next(by.x, next); bz.x = lane_next(bz.x, next);
            bx.y = lane_next(bx.y, next); by.y = lane_next(by.y, next); bz.y = lane_next(bz.y, next);
            xj = x1; yj = y1; zj = z1;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) { ox += ax[k].x + ax[k].y; oy += ay[k].x + ay[k].y; oz += az[k].x + az[k].y; }
        ox += bx.x + bx.y; oy += by.x + by.y; oz += bz.x + bz.y;
    }
    asm volatile("" :: "v"(ox), "v"(oy), "v"(oz));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)gw * 64u + lane] = make_float4(ox, oy, oz, 0.f);
    if (lane == 0) { clk[2 * gw] = c1 - c0; clk[2 * gw + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
static int run(const char *name, const float4 *bodies, float4 *out, unsigned long long *clk, int steps, int pairs_per_step, int waves, int bias)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    int launches = 0;
    // settle: >= 2 s of launches, then time 20
    CK(hipEventRecord(e0));
    do {
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(probe<MODE>, dim3(waves / 8), dim3(512), 0, 0, bodies, out, clk, steps, 25.0f, bias);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    } while (ms < 2000.f);
    CK(hipEventRecord(e0));
    for (launches = 0; launches < 20; launches++) hipLaunchKernelGGL(probe<MODE>, dim3(waves / 8), dim3(512), 0, 0, bodies, out, clk, steps, 25.0f, bias);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * (size_t)waves);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int w = 0; w < waves; w++) if (h[2 * w + 1]) ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double per_launch = ms / launches, pairs = (double)waves * 64.0 * (bias ? ((steps * 20 / 13) & ~1) * 0.5 + ((steps * 6 / 13) & ~1) * 0.5 : (double)(steps & ~1)) * pairs_per_step;
    printf("%-44s %8.3f ms per launch  %.4e ordered interactions/s (2 per evaluated pair)  clock %.3f GHz\n", name, per_launch, 2.0 * pairs / (per_launch * 1e-3),
           ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
    return 0;
}

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 2048, waves = 2048;
    std::vector<float4> h((size_t)waves * 64 * 20);
    unsigned s = 12345u;
    for (auto &b : h) {
        auto r = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 800.f - 400.f; };
        b = make_float4(r(), r(), r(), 0.f);
    }
    float4 *bodies, *out;
    unsigned long long *clk;
    CK(hipMalloc(&bodies, h.size() * sizeof(float4))); CK(hipMalloc(&out, (size_t)waves * 64 * sizeof(float4))); CK(hipMalloc(&clk, 2 * (size_t)waves * 8));
    CK(hipMemcpy(bodies, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
    for (int bias = 0; bias < 2; bias++) {
        printf("== %s\n", bias ? "steps shared 10 : 3 between the older and the younger wave of a SIMD (the product's plan)" : "equal steps for every wave");
        for (int rep = 0; rep < 2; rep++) {
            if (run<0>("A: 8 packed pairs resident, 1 travelling", bodies, out, clk, steps, 16, waves, bias)) return 1;
            if (run<1>("B: 16 resident, 2 travelling (J-packed)", bodies, out, clk, steps / 2, 32, waves, bias)) return 1;
            if (run<2>("A2: as A, two steps per trip (the product)", bodies, out, clk, steps, 16, waves, bias)) return 1;
            if (run<3>("B2: as B, two steps per trip", bodies, out, clk, steps / 2, 32, waves, bias)) return 1;
        }
    }
    return 0;
}
