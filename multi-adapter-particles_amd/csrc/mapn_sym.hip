// mapn_sym.hip -- the all-pairs force with Newton's third law: every UNORDERED pair is evaluated
// once and feeds both bodies (a_i += s r, a_j -= s r), i.e. 14 packed ops + 2 v_rsq_f32 per FOUR
// ordered interactions instead of 11 + 2 per two (mapn_kernels.hip).  Same pair term
// (nBodyGravityCS.hlsl:46-56), same integrator (:103-108), different summation order -- parity with
// the oracle is by the same tolerances as the one-sided kernels (tests/test_gpu_sym.py).
//
// Why it is not free: the reaction -s r belongs to body j, and in the one-sided kernels every lane
// of a wave works on the SAME j (broadcast from the scalar cache), so collecting it would need a
// 64-lane reduction per j.  Here the roles are arranged systolically instead:
//   * a lane owns 16 bodies i (eight packed pairs: positions + accumulators in registers, 246 VGPRs,
//     two waves per SIMD) -- an I-block of 1024 bodies per wave;
//   * a J-block is 64 bodies, ONE per lane, travelling with its reaction accumulator; after each step
//     (8 packed evaluations = 32 interactions per lane) the travelling body and its reaction move one
//     lane on (9 ds_bpermute_b32: through the LDS crossbar), so after 64 steps every lane has met
//     every body of the J-block and each body is back home with its complete reaction.
// Measured (tools/ubench.hip, profiles/r02_ubench.txt, profiles/r02_sym_loop_variants.txt): the moves go through
// the LDS pipe beside the VALU (ds_bpermute_b32; a v_mov_b32_dpp would take VALU cycles), so what a step costs is
// its 131 VALU instructions; the loop gets faster the more bodies i a lane owns per move: 4 bodies 5.3e12
// interactions/s, 8 bodies 7.4e12 in the microbenchmark (one-sided pair term: 4.9e12); in the kernel 16 bodies
// per lane at two waves per SIMD beat 12 at three (-8 %) and 8 at four (-5 %: fewer cycles, but the chip holds a
// lower clock under the denser variants).  While two waves are resident the SIMD's VALU is busy all the time
// (profiles/r02_sym_issue_counters.txt).  LDS float atomics for the reaction (ds_add_f32, ~195 cycles per
// wave-instruction): 1.1e12.
//
// Coverage of the N^2 ordered pairs (N padded to a multiple of 1024 with stand-in bodies that exert no
// force): I-block a meets, symmetrically, the
// I-blocks a+1 .. a+D (mod NB, D = (NB-1)/2, NB = N/1024) 64 bodies at a time, for even NB also
// a+NB/2 when a < NB/2; and itself one-sidedly (no reaction kept).  Every unordered pair of
// blocks is met exactly once; every body collects its force as: rows of its own I-block (role i)
// + one row per meeting of its J-block (role j), all written to scratch and summed in a FIXED
// order by sym_reduce_integrate_kernel -- no float atomics, bit-reproducible.
//
// Sharded over ranks (gather algorithm 4, bottom of this file): a rank launches the kernel for ITS I-blocks
// only (a0, shard_nbl), the reaction rows are kept per (J-block, local I-block); sym_shard_exchange_kernel adds
// them per destination rank, stores them into the owner's receive region, waits for the rows owed to this rank
// and integrates its bodies from its own rows plus the rows received.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "mapn_kernels.h"

namespace mapn {

typedef float v2f __attribute__((ext_vector_type(2)));

namespace {

constexpr uint32_t SYM_IB = SYM_BLOCK;     // bodies per I-block (1024)
constexpr uint32_t JPI = SYM_JPI;          // J-blocks per I-block (16)

__device__ __forceinline__ float lane_next(float v, int addr)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

struct SymBodies {
    v2f xi[SYM_K2], yi[SYM_K2], zi[SYM_K2];
    v2f ax[SYM_K2], ay[SYM_K2], az[SYM_K2];
};

// one step against the travelling body (xj, yj, zj): SYMMETRIC -- b collects the reaction on j.
// The reaction travels UNFOLDED, as the register pair the packed fma chain accumulates in (the even and
// the odd bodies i of the lane): the moves cost no VALU cycles, a fold per step would (3 v_pk_add + the
// re-zeroing copies; measured 0.4-1.3 % slower, profiles/r02_sym_loop_variants.txt).  The meeting's end
// folds the pair once.
__device__ __forceinline__ void sym_step(SymBodies &b, float xj, float yj, float zj, v2f soft2,
                                         v2f &bx, v2f &by, v2f &bz)
{
    v2f rx = bx, ry = by, rz = bz;                       // (reaction so far, 0)
#pragma unroll
    for (int k = 0; k < SYM_K2; k++) {
        const v2f dx = xj - b.xi[k];
        const v2f dy = yj - b.yi[k];
        const v2f dz = zj - b.zi[k];
        v2f d = __builtin_elementwise_fma(dx, dx, soft2);
        d = __builtin_elementwise_fma(dy, dy, d);
        d = __builtin_elementwise_fma(dz, dz, d);
        v2f inv;
        inv.x = __builtin_amdgcn_rsqf(d.x);
        inv.y = __builtin_amdgcn_rsqf(d.y);
        const v2f inv3 = inv * inv * inv;
        b.ax[k] = __builtin_elementwise_fma(dx, inv3, b.ax[k]);
        b.ay[k] = __builtin_elementwise_fma(dy, inv3, b.ay[k]);
        b.az[k] = __builtin_elementwise_fma(dz, inv3, b.az[k]);
        rx = __builtin_elementwise_fma(-dx, inv3, rx);
        ry = __builtin_elementwise_fma(-dy, inv3, ry);
        rz = __builtin_elementwise_fma(-dz, inv3, rz);
    }
    bx = rx; by = ry; bz = rz;
}

// one step, ONE-SIDED (the I-block against itself: every ordered pair is met from both sides anyway)
__device__ __forceinline__ void one_step(SymBodies &b, float xj, float yj, float zj, v2f soft2)
{
#pragma unroll
    for (int k = 0; k < SYM_K2; k++) {
        const v2f dx = xj - b.xi[k];
        const v2f dy = yj - b.yi[k];
        const v2f dz = zj - b.zi[k];
        v2f d = __builtin_elementwise_fma(dx, dx, soft2);
        d = __builtin_elementwise_fma(dy, dy, d);
        d = __builtin_elementwise_fma(dz, dz, d);
        v2f inv;
        inv.x = __builtin_amdgcn_rsqf(d.x);
        inv.y = __builtin_amdgcn_rsqf(d.y);
        const v2f inv3 = inv * inv * inv;
        b.ax[k] = __builtin_elementwise_fma(dx, inv3, b.ax[k]);
        b.ay[k] = __builtin_elementwise_fma(dy, inv3, b.ay[k]);
        b.az[k] = __builtin_elementwise_fma(dz, inv3, b.az[k]);
    }
}

}  // namespace

// grid = (NB, S)   block = 64 * WAVES
// Workgroup (s, a): I-block a, part s of S of its meetings.  The meetings of an I-block are numbered
// m = 0 .. M-1: m < 16 -> itself, J-block a*16 + m, one-sided; then 16 per partner block a + d.
// The workgroup's part is dealt to its waves: whole meetings first, the remainder shared step-wise.  Every wave keeps the I-block's
// accumulators in registers for all of its meetings; at the end the WAVES copies are combined in LDS
// in ascending wave order into ONE row arow[a][s][1024]; each symmetric meeting writes ONE row
// brow[jblock][d-1][64] with the reactions of the J-block's bodies.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void force_sym_kernel(const SymArgs p)
{
    __shared__ float comb[WAVES][3][SYM_IB];               // 48 KiB at 4 waves: the two workgroups a CU holds fit

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // grid = (I-blocks, parts): dispatch order is part-major -- all blocks' part 0, then part 1, ... -- so that the
    // LATE workgroups are the small ones when the parts taper (below)
    const uint32_t s = blockIdx.y, la = blockIdx.x, a = p.a0 + la;    // a: the I-block in the whole job; la: among this launch's
    const uint32_t nb = p.nb, half = p.half_d;             // half_d = NB/2 when NB is even, else 0
    const uint32_t D = (nb - 1u) / 2u;
    const uint32_t M = JPI * (1u + D) + ((half && a < half) ? JPI : 0u);
    // part s of S (even split: the first `rem` parts take one more), then the part's `cnt` meetings to its
    // waves: q = cnt / WAVES whole meetings each, and the remaining r < WAVES meetings are SHARED -- every
    // wave runs 64 / WAVES of such a meeting's 64 steps (wave w starts with the J-block rotated by w * SEG
    // lanes), the waves' partial reactions are added in LDS in ascending wave order.  All waves of the
    // workgroup finish together (before: a wave with one meeting more kept the other three waiting -- at
    // 65 536 bodies a quarter of the workgroups ran 5 meeting-times for 4.25 of work).
    constexpr uint32_t SEG = 64u / WAVES;
    // Part sizes TAPER: the first taper1 parts weigh 4 units, the next taper2 parts 2, the rest 1 (taper1 = parts:
    // all equal).  The CUs do not all run at one speed and a launch is only a few rounds of workgroups, so with
    // equal parts the slots that finish first idle for up to a whole workgroup time at the end (65 536 bodies, 32
    // equal parts: 11 % of the wave slots empty over the launch); small workgroups LAST keep that tail short.
    auto units = [&](uint32_t x) {
        return x <= p.taper1 ? 4u * x : x <= p.taper1 + p.taper2 ? 4u * p.taper1 + 2u * (x - p.taper1) : 4u * p.taper1 + 2u * p.taper2 + (x - p.taper1 - p.taper2);
    };
    const uint32_t U = units(p.parts);
    const uint32_t pm0 = (uint32_t)(((uint64_t)M * units(s)) / U), pm1 = (uint32_t)(((uint64_t)M * units(s + 1u)) / U);
    const uint32_t cnt = pm1 - pm0;
    // (p.whole_only: the A/B form -- whole meetings only, the first cnt % WAVES waves take one more)
    const uint32_t wm0 = pm0 + (uint32_t)(((uint64_t)cnt * w) / WAVES), wm1 = pm0 + (uint32_t)(((uint64_t)cnt * (w + 1u)) / WAVES);
    const uint32_t q = p.whole_only ? wm1 - wm0 : cnt / WAVES, r = p.whole_only ? 0u : cnt % WAVES, items = q + r;
    const uint32_t first = p.whole_only ? wm0 : pm0 + w * q, first_shared = pm0 + WAVES * q;
    __shared__ float part[WAVES - 1][WAVES][3][64];

    const float4 *__restrict__ pos = p.pos_old;
    // N need not be a multiple of the block: bodies past the end are stand-ins so far away that
    // (|r|^2 + soft^2)^(-3/2) underflows to zero -- they exert and feel exactly nothing, and what the
    // kernel accumulates for them is never read (sym_reduce_integrate_kernel stops at n)
    const float4 far = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);
    auto body = [&](uint32_t i) { return i < p.n ? pos[i] : far; };
    SymBodies b;
#pragma unroll
    for (int k = 0; k < SYM_K2; k++) {
        const float4 b0 = body(a * SYM_IB + (2 * k) * 64u + lane), b1 = body(a * SYM_IB + (2 * k + 1) * 64u + lane);
        b.xi[k] = v2f{b0.x, b1.x}; b.yi[k] = v2f{b0.y, b1.y}; b.zi[k] = v2f{b0.z, b1.z};
        b.ax[k] = v2f{0.f, 0.f}; b.ay[k] = v2f{0.f, 0.f}; b.az[k] = v2f{0.f, 0.f};
    }
    const v2f soft2 = v2f{p.soft2, p.soft2};
    const int next = (int)((lane + 1u) & 63u) * 4;         // ds_bpermute: take the value of lane + 1

    // J-block of meeting m (and whether it is symmetric); the NEXT meeting's bodies are fetched while
    // the current one is computed (a meeting is ~9 us of a wave's life, a global load ~1-2 us)
    auto meeting = [&](uint32_t m, uint32_t &jb, uint32_t &d) {
        const uint32_t grp = m / JPI, t = m % JPI;         // grp 0: own block; grp g: partner a + g (the last may be the half ring)
        d = grp <= D ? grp : half;
        uint32_t ap = a + d;
        ap = ap >= nb ? ap - nb : ap;
        jb = ap * JPI + t;
    };
    // where the reactions of J-block jb from this I-block go: one row per partner distance, or -- sharded --
    // one row per I-block of this rank (sym_shard_send_kernel adds them up per destination rank)
    auto brow_row = [&](uint32_t jb, uint32_t d) -> size_t {
        return p.shard_nbl ? (size_t)jb * p.shard_nbl + la : (size_t)jb * p.brows + (d - 1u);
    };
    // diagnostic launches only (mapn_measure_clock): stamps around the wave's meetings; null otherwise
    unsigned long long st_c = 0, st_r = 0;
    if (p.stamps) asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(st_r), "=s"(st_c));
    auto item = [&](uint32_t it, uint32_t &jb, uint32_t &d, uint32_t &rot) {
        meeting(it < q ? first + it : first_shared + (it - q), jb, d);
        rot = it < q ? 0u : w * SEG;
    };
    uint32_t jb = 0, d = 0, rot = 0;
    float4 pn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (items) { item(0u, jb, d, rot); pn = body(jb * 64u + ((lane + rot) & 63u)); }
    for (uint32_t it = 0; it < items; it++) {
        const uint32_t jb_cur = jb, d_cur = d, rot_cur = rot;
        const int steps = it < q ? 64 : (int)SEG;
        float xj = pn.x, yj = pn.y, zj = pn.z;
        if (it + 1u < items) { item(it + 1u, jb, d, rot); pn = body(jb * 64u + ((lane + rot) & 63u)); }
        // (Alternatives to moving the position, measured on one box each: a wave-private LDS copy of the J-block
        //  read with one ds_read_b128 per step, also one step ahead: 2-3 % slower; re-reading body (lane + k) % 64
        //  from global memory every step, fetched one step ahead: 9 % slower.)
        if (d_cur == 0u) {
#pragma nounroll
            for (int k = 0; k < steps; k++) {
                const float nx = lane_next(xj, next), ny = lane_next(yj, next), nz = lane_next(zj, next);
                one_step(b, xj, yj, zj, soft2);
                xj = nx; yj = ny; zj = nz;
            }
        } else {
            v2f bx = v2f{0.f, 0.f}, by = v2f{0.f, 0.f}, bz = v2f{0.f, 0.f};
#pragma nounroll                                           // (unrolled by two: no loop-carried copies, but the moves issue late: 2 % slower)
            for (int k = 0; k < steps; k++) {
                // the travelling position does not change during the step: its move overlaps the step
                const float nx = lane_next(xj, next), ny = lane_next(yj, next), nz = lane_next(zj, next);
                sym_step(b, xj, yj, zj, soft2, bx, by, bz);
                xj = nx; yj = ny; zj = nz;
                bx.x = lane_next(bx.x, next); by.x = lane_next(by.x, next); bz.x = lane_next(bz.x, next);
                bx.y = lane_next(bx.y, next); by.y = lane_next(by.y, next); bz.y = lane_next(bz.y, next);
            }
            const float fx = bx.x + bx.y, fy = by.x + by.y, fz = bz.x + bz.y;
            if (it < q) {
                // 64 moves: every body is back in its home lane with its complete reaction from this I-block
                p.brow[brow_row(jb_cur, d_cur) * 64u + lane] = make_float4(fx, fy, fz, 0.f);
            } else {
                // SEG moves: this lane holds body (lane + rot + SEG) % 64 with this wave's share of its reaction
                const uint32_t home = (lane + rot_cur + SEG) & 63u;
                part[it - q][w][0][home] = fx; part[it - q][w][1][home] = fy; part[it - q][w][2][home] = fz;
            }
        }
    }

    if (p.stamps) {
        asm volatile("" :: "v"(b.ax[0]), "v"(b.ay[0]), "v"(b.az[0]));
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            const size_t wave = ((size_t)la * p.parts + s) * WAVES + w;
            p.stamps[2 * wave] = c1 - st_c;
            p.stamps[2 * wave + 1] = r1 - st_r;
        }
    }
    // combine the WAVES copies of the I-block's accumulators: every wave parks its copy in LDS, then
    // each thread sums one body's WAVES values in ascending wave order (a fixed order: bit-reproducible)
#pragma unroll
    for (int k = 0; k < SYM_K2; k++) {
        const uint32_t e0 = (2 * k) * 64u + lane, e1 = e0 + 64u;
        comb[w][0][e0] = b.ax[k].x; comb[w][0][e1] = b.ax[k].y;
        comb[w][1][e0] = b.ay[k].x; comb[w][1][e1] = b.ay[k].y;
        comb[w][2][e0] = b.az[k].x; comb[w][2][e1] = b.az[k].y;
    }
    __syncthreads();
    float4 *row = p.arow + ((size_t)la * p.parts + s) * SYM_IB;
    for (uint32_t e = threadIdx.x; e < SYM_IB; e += 64u * WAVES) {
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ww++) { ax += comb[ww][0][e]; ay += comb[ww][1][e]; az += comb[ww][2][e]; }
        row[e] = make_float4(ax, ay, az, 0.f);
    }
    // the shared meetings' reactions: wave t adds the WAVES shares of shared meeting t in ascending wave order
    if (w < r) {
        uint32_t jbs, ds;
        meeting(first_shared + w, jbs, ds);
        if (ds != 0u) {
            float fx = 0.f, fy = 0.f, fz = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ww++) { fx += part[w][ww][0][lane]; fy += part[w][ww][1][lane]; fz += part[w][ww][2][lane]; }
            p.brow[brow_row(jbs, ds) * 64u + lane] = make_float4(fx, fy, fz, 0.f);
        }
    }
}

// One thread per body: rows of its I-block (role i) in ascending part order, then the rows of its
// J-block (role j) in ascending partner distance, then mass, kick, damp, drift (hlsl:103-108).
__global__ __launch_bounds__(256) void sym_reduce_integrate_kernel(const SymArgs p)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= p.n) return;
    const uint32_t a = i / SYM_IB, jb = i >> 6;
    float ax = 0.f, ay = 0.f, az = 0.f;
    const float4 *ar = p.arow + (size_t)a * p.parts * SYM_IB + (i - a * SYM_IB);
    uint32_t s = 0;
    for (; s + 8u <= p.parts; s += 8u) {                   // 8 loads in flight (one wave per SIMD: nothing else hides the latency), summed in ascending order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = ar[(size_t)(s + u) * SYM_IB];
#pragma unroll
        for (int u = 0; u < 8; u++) { ax += v[u].x; ay += v[u].y; az += v[u].z; }
    }
    for (; s < p.parts; s++) {
        const float4 v = ar[(size_t)s * SYM_IB];
        ax += v.x; ay += v.y; az += v.z;
    }
    // rows d-1 = 0 .. D-1 always exist; the half-ring row D exists for the blocks that were the far partner
    const uint32_t D = (p.nb - 1u) / 2u;
    const uint32_t rows = D + ((p.half_d && a >= p.half_d) ? 1u : 0u);
    const float4 *br = p.brow + (size_t)jb * p.brows * 64u + (i & 63u);
    uint32_t r = 0;
    for (; r + 8u <= rows; r += 8u) {                      // 8 loads in flight, summed in ascending order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = br[(size_t)(r + u) * 64u];
#pragma unroll
        for (int u = 0; u < 8; u++) { ax += v[u].x; ay += v[u].y; az += v[u].z; }
    }
    for (; r < rows; r++) {
        const float4 v = br[(size_t)r * 64u];
        ax += v.x; ay += v.y; az += v.z;
    }
    ax *= p.mass; ay *= p.mass; az *= p.mass;
    const float4 pos = p.pos_old[i];
    const float *v = p.vel_old + 3 * (size_t)i;
    float vx = v[0], vy = v[1], vz = v[2];
    vx = __builtin_fmaf(ax, p.dt, vx) * p.damping;
    vy = __builtin_fmaf(ay, p.dt, vy) * p.damping;
    vz = __builtin_fmaf(az, p.dt, vz) * p.damping;
    float4 o;
    o.x = __builtin_fmaf(vx, p.dt, pos.x);
    o.y = __builtin_fmaf(vy, p.dt, pos.y);
    o.z = __builtin_fmaf(vz, p.dt, pos.z);
    o.w = __builtin_sqrtf(__builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)));
    p.pos_new[i] = o;
    float *vo = p.vel_new + 3 * (size_t)i;
    vo[0] = vx; vo[1] = vy; vo[2] = vz;
}


// ---- the symmetric step sharded over ranks ----------------------------------------------------------
namespace {
typedef float f4v __attribute__((ext_vector_type(4)));

// does I-block a meet I-block b symmetrically (b's bodies travelling)?  The schedule of force_sym_kernel.
__device__ __forceinline__ bool sym_meets(uint32_t a, uint32_t b, uint32_t nb, uint32_t half)
{
    const uint32_t d = b >= a ? b - a : b + nb - a, D = (nb - 1u) / 2u;
    return (d >= 1u && d <= D) || (half && d == half && a < half);
}
}  // namespace

// grid <= 1024 workgroups (all co-resident: they wait for each other through the ticket)   block = 256
// One launch does both halves of the reaction exchange.
//  (1) SEND: for every rank q this rank produced reactions for and every body of q: the rows of this rank's
//      I-blocks that met the body's block, added in ascending block order, stored as ONE float4 into rank q's
//      receive region (row [this rank]) with a system-scope write-through store -- over xGMI when q is another
//      GPU.  Once acknowledged (vmcnt(0)) the stores are in q's memory: no cache write-back is owed (a release
//      fence here would write back the whole L2, full of this step's rows: measured 10+ us per step).  The
//      last workgroup through the ticket stores the arrival flags.
//  (2) REDUCE: lanes 0 .. world-1 of the first wave wait (bounded) for the arrival flags of the ranks that owe
//      this rank rows; then G threads per body (a rank's slice is small -- 8192 bodies at 65 536 / 8 -- so one
//      thread per body would leave the rows' loads latency-bound): thread (body, g) adds the a-rows of parts
//      [g P/G, (g+1) P/G) in ascending order, thread (body, 0) adds the G sums in ascending g, then the rows
//      received, nearest sender first (this rank, rank - 1, rank - 2, ...), then mass, kick, damp, drift
//      (hlsl:103-108) -- a fixed order throughout, so the replicas stay bit-identical.
template <int G>
__global__ __launch_bounds__(256) void sym_shard_exchange_kernel(const SymShardArgs p)
{
    constexpr uint32_t B = 256u / G;                       // bodies per workgroup and pass
    __shared__ uint32_t ok;
    __shared__ float part[G][3][B];

    const uint32_t total = p.world * p.count;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
        const uint32_t q = t / p.count, jl = t - q * p.count;
        if (!((p.send_mask >> q) & 1u)) continue;
        const uint32_t b = t / SYM_BLOCK, jb = t >> 6;     // t is the body's index in the whole job
        const float4 *rows = p.brow + (size_t)jb * p.nbl * 64u + (t & 63u);
        float fx = 0.f, fy = 0.f, fz = 0.f;
        for (uint32_t la = 0; la < p.nbl; la += 8u) {      // eight loads in flight, added in ascending block order
            float4 v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const bool met = la + u < p.nbl && sym_meets(p.a0 + la + u, b, p.nb, p.half_d);
                v[u] = met ? rows[(size_t)(la + u) * 64u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) { fx += v[u].x; fy += v[u].y; fz += v[u].z; }
        }
        const f4v o = {fx, fy, fz, 0.f};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p.recv_peer[q] + (size_t)p.rank * p.count + jl), "v"(o) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64u) {
        if (threadIdx.x == 0) {
            const uint32_t prev = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1u == gridDim.x) {
                __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
                for (uint32_t q = 0; q < p.world; q++)
                    if ((p.send_mask >> q) & 1u)
                        __hip_atomic_store(p.flags_peer[q] + SYM_FLAG_BASE + p.rank, p.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        const uint32_t q = threadIdx.x;
        const bool need = q < p.world && ((p.recv_mask >> q) & 1u);
        uint32_t good = 1u;
        if (need) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while ((int32_t)(__hip_atomic_load(p.flags_mine + SYM_FLAG_BASE + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - p.step) < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                    good = 0u;
                    __hip_atomic_store(p.status, 1u + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        const uint32_t all_good = __builtin_amdgcn_ballot_w64(good == 0u) == 0ull ? 1u : 0u;
        if (threadIdx.x == 0) ok = all_good;
    }
    __syncthreads();
    if (!ok) return;

    const uint32_t bl = threadIdx.x % B, g = threadIdx.x / B;
    for (uint32_t base = blockIdx.x * B; base < p.count; base += gridDim.x * B) {
        const uint32_t il = base + bl;
        const bool live = il < p.count;
        const uint32_t la = live ? il / SYM_BLOCK : 0u;
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (live) {
            const float4 *ar = p.arow + (size_t)la * p.parts * SYM_BLOCK + (il - la * SYM_BLOCK);
            const uint32_t s0 = (uint32_t)(((uint64_t)p.parts * g) / G), s1 = (uint32_t)(((uint64_t)p.parts * (g + 1u)) / G);
            uint32_t s = s0;
            for (; s + 8u <= s1; s += 8u) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = ar[(size_t)(s + u) * SYM_BLOCK];
#pragma unroll
                for (int u = 0; u < 8; u++) { ax += v[u].x; ay += v[u].y; az += v[u].z; }
            }
            for (; s < s1; s++) {
                const float4 v = ar[(size_t)s * SYM_BLOCK];
                ax += v.x; ay += v.y; az += v.z;
            }
        }
        __syncthreads();                                   // (the previous pass has read `part`)
        part[g][0][bl] = ax; part[g][1][bl] = ay; part[g][2][bl] = az;
        __syncthreads();
        if (g != 0u || !live) continue;
        ax = ay = az = 0.f;
#pragma unroll
        for (int gg = 0; gg < G; gg++) { ax += part[gg][0][bl]; ay += part[gg][1][bl]; az += part[gg][2][bl]; }
        // the rows received: uncached region, read past this GPU's caches with system-scope loads (global_load_dwordx2
        // sc0 sc1, issued by the compiler so that it places the waits), all in flight, added nearest sender first
        unsigned long long lo[P2P_MAX_RANKS], hi[P2P_MAX_RANKS];
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)P2P_MAX_RANKS; k++) {
            const uint32_t q = p.rank >= k ? p.rank - k : p.rank + p.world - k;
            const bool need = k < p.world && ((p.recv_mask >> q) & 1u);
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(p.recv_mine + (size_t)(need ? q : p.rank) * p.count + il);
            lo[k] = need ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
            hi[k] = need ? __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)P2P_MAX_RANKS; k++) {
            ax += __builtin_bit_cast(float, (uint32_t)lo[k]);
            ay += __builtin_bit_cast(float, (uint32_t)(lo[k] >> 32));
            az += __builtin_bit_cast(float, (uint32_t)hi[k]);
        }
        ax *= p.mass; ay *= p.mass; az *= p.mass;
        const uint32_t i = p.rank * p.count + il;
        const float4 pos = p.pos_old[i];
        const float *v = p.vel_old + 3 * (size_t)i;
        float vx = v[0], vy = v[1], vz = v[2];
        vx = __builtin_fmaf(ax, p.dt, vx) * p.damping;
        vy = __builtin_fmaf(ay, p.dt, vy) * p.damping;
        vz = __builtin_fmaf(az, p.dt, vz) * p.damping;
        float4 o;
        o.x = __builtin_fmaf(vx, p.dt, pos.x);
        o.y = __builtin_fmaf(vy, p.dt, pos.y);
        o.z = __builtin_fmaf(vz, p.dt, pos.z);
        o.w = __builtin_sqrtf(__builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)));
        p.pos_new[i] = o;
        float *vo = p.vel_new + 3 * (size_t)i;
        vo[0] = vx; vo[1] = vy; vo[2] = vz;
    }
}

hipError_t launch_force_sym(const SymArgs &a, uint32_t waves, hipStream_t st)
{
    const dim3 grid(a.shard_nbl ? a.shard_nbl : a.nb, a.parts);
    // (experiments only: MAPN_SYM_PAD_LDS=bytes of unused dynamic LDS, e.g. 60000 leaves room for ONE workgroup per CU)
    static const unsigned pad = [] { const char *e = getenv("MAPN_SYM_PAD_LDS"); return e ? (unsigned)strtoul(e, nullptr, 10) : 0u; }();
    if (waves == 4) hipLaunchKernelGGL((force_sym_kernel<4>), grid, dim3(256), pad, st, a);
    else if (waves == 8) hipLaunchKernelGGL((force_sym_kernel<8>), grid, dim3(512), 0, st, a);
    else return hipErrorInvalidConfiguration;
    return hipGetLastError();
}

hipError_t launch_sym_shard_exchange(const SymShardArgs &a, hipStream_t st)
{
    // enough threads to keep the rows' loads in flight: 8 per body up to 16 384 bodies, 4 up to 65 536, else 1;
    // never more workgroups than are resident together (they wait for each other through the ticket)
    if (a.count <= 16384u) hipLaunchKernelGGL((sym_shard_exchange_kernel<8>), dim3(std::min(1024u, (a.count + 31u) / 32u)), dim3(256), 0, st, a);
    else if (a.count <= 65536u) hipLaunchKernelGGL((sym_shard_exchange_kernel<4>), dim3(std::min(1024u, (a.count + 63u) / 64u)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((sym_shard_exchange_kernel<1>), dim3(std::min(1024u, (a.count + 255u) / 256u)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_sym_reduce(const SymArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(sym_reduce_integrate_kernel, dim3((a.n + 255u) / 256u), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace mapn
