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
// a+NB/2 when a is the runner of that pair (sym_runs_half: the pairs alternate between the two halves of the ring); and itself (the reaction is dropped).  Every unordered pair of
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

// A row leaves the wave: rows are read by OTHER workgroups in the next launch, mostly on other XCDs (through memory), so
// keeping them dirty in this XCD's L2 until the end-of-kernel write-back only lengthens the launch's tail; wt != 0 stores
// them write-through (sc1) as they are produced.
typedef float f4r __attribute__((ext_vector_type(4)));
typedef float f3r __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void store_row(SymRow *dst, float x, float y, float z, uint32_t wt)
{
    if (wt) {
        const f3r o = {x, y, z};
        asm volatile("global_store_dwordx3 %0, %1, off sc1" : : "v"(dst), "v"(o) : "memory");
    } else {
        *dst = SymRow{x, y, z};
    }
}

// write-through store at system scope: once acknowledged (vmcnt) the data is in the owner's memory
__device__ __forceinline__ void store_sys(float4 *dst, float x, float y, float z, float w)
{
    const f4r o = {x, y, z, w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(o) : "memory");
}

// lanes 0 .. 63 of the calling wave each wait (bounded) for one counter to reach `need`; returns 1 when all did.  `dead`: this rank's
// SYM_DEAD_WORD -- once a wait of the rank has given up (or a check has failed) no later one waits again: read only when the counter is
// not there yet, so it costs nothing in a healthy step
__device__ __forceinline__ uint32_t wait_counters(const uint32_t *counters, uint32_t index, bool need_it, uint32_t need,
                                                  uint64_t timeout_ticks, uint32_t *status, uint32_t code, uint32_t *dead)
{
    uint32_t good = 1u;
    if (need_it) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while ((int32_t)(__hip_atomic_load(counters + index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - need) < 0) {
            if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { good = 0u; break; }
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                good = 0u;
                __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    return __builtin_amdgcn_ballot_w64(good == 0u) == 0ull ? 1u : 0u;
}

// a body read past this GPU's caches (system-scope loads): a line of a buffer the PEERS store into must not come from a cache
__device__ __forceinline__ float4 load_sys(const float4 *src)
{
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(src);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_float4(__builtin_bit_cast(float, (uint32_t)lo), __builtin_bit_cast(float, (uint32_t)(lo >> 32)),
                       __builtin_bit_cast(float, (uint32_t)hi), __builtin_bit_cast(float, (uint32_t)(hi >> 32)));
}

// XOR over the 32 lanes of a half-wave (all of them active), result in every lane: four row rotations (DPP: VALU only, no wait)
// inside the rows of 16, one ds_swizzle_b32 (swap the two rows of a half-wave) across -- one trip through the LDS crossbar
// instead of the five of a shuffle tree
__device__ __forceinline__ uint32_t xor_reduce32(uint32_t h)
{
#define MAPN_ROR(n) h ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x120 + (n), 0xf, 0xf, false)
    MAPN_ROR(8); MAPN_ROR(4); MAPN_ROR(2); MAPN_ROR(1);
#undef MAPN_ROR
    return h ^ (uint32_t)__builtin_amdgcn_ds_swizzle((int)h, 0x401F);     // bit mode: and 0x1f, or 0, xor 0x10 = lane ^ 16
}

// Gather algorithm 5, receiver side: re-compute the checksums of what the peers PUSHED into `replica` (read past the caches, after
// their counters) and compare them with the words the pushers stored behind the data.  Called by whole waves: half-wave h of wave
// `wv` (of `nw`) takes the 32-body groups 2 wv + h, 2 wv + h + 2 nw, ... of the (world - 1) x count / 32 pushed groups.
// `active`: the publication was that of a PARTIALLY ACTIVE step -- only the bodies [0, active) of the whole job were pushed (a multiple of 64: whole groups).
__device__ __forceinline__ void verify_pushed(const float4 *replica, const uint32_t *sums, uint32_t epoch, uint32_t count, uint32_t world, uint32_t rank,
                                              uint32_t self, uint32_t wv, uint32_t nw, uint32_t lane, uint32_t *status, uint32_t *dead, uint32_t active)
{
    const uint32_t per = count / 32u, ng = (world - 1u) * per;            // (count is a multiple of 1024: ng is even)
    for (uint32_t base = 2u * wv; base < ng; base += 2u * nw) {
        const uint32_t gi = base + (lane >> 5), k = gi / per, grp = gi - k * per;
        const uint32_t qp = k < rank ? k : k + 1u, q = self ? rank : qp;   // (loopback timing: the "peers" are this rank -- the data is its own slice,
        const uint32_t body = q * count + grp * 32u + (lane & 31u);        //  the checksum row the one it stored for "peer" qp)
        if (body >= active) continue;                                      // (a frozen body: nobody pushed it)
        const uint32_t want = __hip_atomic_load(sums + ((size_t)(epoch & 1u) * world + qp) * per + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const float4 v = load_sys(replica + body);
        uint32_t h = sym_push_checksum(__builtin_bit_cast(uint32_t, v.x), __builtin_bit_cast(uint32_t, v.y), __builtin_bit_cast(uint32_t, v.z),
                                       __builtin_bit_cast(uint32_t, v.w), body);
        h = xor_reduce32(h);
        if ((h ^ sym_push_epoch_mix(epoch)) != want) {
            __hip_atomic_store(status, 0x200u + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
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

}  // namespace

// grid = (I-blocks of the launch, parts)   block = 64 * WAVES
// Workgroup (la, s) = part s of I-block a = a0 + la.  Which STEPS each of its waves runs comes from the host-built
// plan (mapn_sym_plan.h): wave v = s * WAVES + w runs the linear steps [bounds[v], bounds[v + 1]) of the block's
// meetings in this launch (step 64 m + k = step k of meeting m; every wave carries the same cost, at least 64
// steps).  A wave keeps the I-block's accumulators in registers for all of its steps; at the end the WAVES copies
// are combined in LDS in ascending wave order into ONE row arow[la][s][1024].  A symmetric meeting run whole by
// one wave writes its row brow[...][64] directly; a meeting cut between two waves of the workgroup is put together
// in LDS (first steps + last steps); one cut between two workgroups leaves its first steps in the meeting's row
// and its last steps in the head row brow1[la][s] of the later workgroup.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void force_sym_kernel(const SymArgs p)
{
    __shared__ float comb[WAVES][3][SYM_IB];               // 48 KiB at 4 waves: the two workgroups a CU holds fit
    __shared__ float edge[2][WAVES][3][64];                // [0] a wave's partial of the meeting its range ENDS in, [1] of the one it STARTS in

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // diagnostic launches only (MAPN_STAMP_DUMP): wall-clock stamps of the wave's life; null otherwise
    unsigned long long tl_entry = 0, tl_loop = 0;
    if (p.timeline) asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tl_entry));
    // dispatch order is part-major -- all blocks' part 0, then part 1, ... -- so that the LATE workgroups are the
    // small ones when the parts taper
    // (XCD-weighted parts, sets == 16: the parts of a block are spread over the dies -- workgroup (x, y) lands on XCD x mod 8 and
    //  runs part y of block (x + y) mod blocks, so part s of block la runs on die (la - s) mod 8, which its share of the steps was sized for)
    // (class-aware XCD weights: which (block, part) this workgroup runs is a host-built table -- the heavy blocks' parts on the fast dies)
    uint32_t s = blockIdx.y, la = p.sets > 2u ? (blockIdx.x + blockIdx.y) % gridDim.x : (blockIdx.x ^ p.la_flip);
    if (p.wgmap) { const uint32_t m = p.wgmap[blockIdx.y * gridDim.x + blockIdx.x]; la = m >> 16; s = m & 0xffffu; }
    const uint32_t a = p.a0 + la;                          // a: the I-block in the whole job; la: among this launch's
    const uint32_t nb = p.nb, half = p.half_d;             // half_d = NB/2 when NB is even, else 0
    const uint32_t D = (nb - 1u) / 2u;
    const uint32_t cls = sym_runs_half(a, half) ? 0u : 1u; // class 0: the blocks that also run the half-ring group (alternating between the two halves of the ring)
    const uint32_t set = cls + (p.sets > 2u ? 2u * (la & 7u) : 0u);
    const uint32_t *bounds = p.tab + set * (p.nwaves + 1u);
    const uint32_t t0 = bounds[s * WAVES + w], t1 = bounds[s * WAVES + w + 1u];

    const float4 *__restrict__ pos = p.pos_old;
    // N need not be a multiple of the block: bodies past the end are stand-ins so far away that
    // (|r|^2 + soft^2)^(-3/2) underflows to zero -- they exert and feel exactly nothing, and what the
    // kernel accumulates for them is never read (sym_reduce_integrate_kernel stops at n)
    const float4 far = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);
    auto body = [&](uint32_t i) { return i < p.n ? pos[i] : far; };
    // a travelling body: under gather algorithm 5 it may lie in a slice a PEER stored into this rank's buffer
    auto body_j = [&](uint32_t i) { return i >= p.n ? far : p.wait_counters ? load_sys(pos + i) : pos[i]; };
    const v2f soft2 = v2f{p.soft2, p.soft2};
    const int next = (int)((lane + 1u) & 63u) * 4;         // ds_bpermute: take the value of lane + 1

    // J-block and partner distance of meeting m of this launch (d = 0: the block itself, one-sided)
    auto meeting = [&](uint32_t m, uint32_t &jb, uint32_t &d, uint32_t &g) {
        g = p.g0 + m / JPI;
        d = g <= D ? g : half;
        uint32_t ap = a + d;
        ap = ap >= nb ? ap - nb : ap;
        jb = ap * JPI + m % JPI;
    };
    // where the reactions of J-block jb from this I-block go: one row per group of the launch, or -- sharded --
    // one row per I-block of this rank (sym_shard_exchange_kernel adds them up per destination rank)
    auto brow_row = [&](uint32_t jb, uint32_t g) -> size_t {
        return p.shard_nbl ? (size_t)jb * p.shard_nbl + la : (size_t)jb * p.brows + (g - (p.g0 ? p.g0 : 1u));
    };
    // diagnostic launches only (mapn_measure_clock): stamps around the wave's meetings; null otherwise
    unsigned long long st_c = 0, st_r = 0;
    if (p.stamps) asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(st_r), "=s"(st_c));
    uint32_t jb = 0, d = 0, g = 0;
    float4 pn = make_float4(0.f, 0.f, 0.f, 0.f);
    // The wave's start is a chain of memory round trips (~1 us each); they are issued so that they overlap: the I-block's
    // loads go out first, the wait for the peers' counters (gather algorithm 5) polls while they are in flight, the first
    // J-block is requested as soon as the counters allow, and only then are the I-block's slices taken through LDS
    // (entry -> first step 2.6 us when these ran one after the other).
    // The first piece may start inside a meeting (k0 = t0 % 64 steps of it ran in the previous wave): the lane then
    // starts with body (lane + k0) % 64; every later piece starts a meeting.  The NEXT piece's bodies are fetched
    // while the current one is computed (a meeting is ~9 us of a wave's life, a global load ~1-2 us).
    uint32_t arrived = 1u;                                 // wave-uniform: the peers' position counters were there (nothing to wait for: 1)
    auto first_piece = [&]() {
        if (p.wait_counters) {
            // (the I-block is this rank's own slice; everything else waits for the peers' pushes -- the counters have
            //  normally been there since before this launch started)
            const bool need = lane < p.wait_world && lane != p.wait_rank;
            arrived = wait_counters(p.wait_counters, p.wait_self ? p.wait_rank : lane, need, p.wait_need, p.wait_timeout_ticks, p.wait_status, 1u + lane, p.wait_dead);
        }
        if (t0 < t1) { meeting(t0 >> 6, jb, d, g); pn = body_j(jb * 64u + ((lane + t0) & 63u)); }
        // what the peers pushed is CHECKED here, once per launch, spread over the launch's waves (a few loads per wave, in flight
        // together with the first J-block: the wave waits for that one anyway)
        // (only when the counters arrived: after a timeout or the dead word the slices are simply not there, and the check would
        //  overwrite "peer q is late" with "peer q pushed corrupted data": ADVICE r4)
        if (p.verify_sums && arrived)
            verify_pushed(pos, p.verify_sums, p.verify_epoch, p.verify_count, p.wait_world, p.wait_rank, p.wait_self,
                          (la * p.parts + s) * WAVES + w, gridDim.x * gridDim.y * WAVES, lane, p.wait_status, p.wait_dead, p.verify_active);
    };
    SymBodies b;
    if (p.stage_iblock) {
        // the workgroup's waves all hold the SAME I-block: each fetches 16 / WAVES of its 16 lane-slices, the slices meet in LDS
        // (the space the closing combination uses) and every wave takes all 16 from there -- a quarter of the global loads in
        // the moment when every wave of the launch starts at once
        float4 *stage = reinterpret_cast<float4 *>(&comb[0][0][0]);
        constexpr uint32_t PER = 2u * SYM_K2 / WAVES;
        float4 mine[PER];
#pragma unroll
        for (uint32_t c = 0; c < PER; c++) mine[c] = body(a * SYM_IB + (w * PER + c) * 64u + lane);
        first_piece();
#pragma unroll
        for (uint32_t c = 0; c < PER; c++) stage[(w * PER + c) * 64u + lane] = mine[c];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SYM_K2; k++) {
            const float4 b0 = stage[(2 * k) * 64u + lane], b1 = stage[(2 * k + 1) * 64u + lane];
            b.xi[k] = v2f{b0.x, b1.x}; b.yi[k] = v2f{b0.y, b1.y}; b.zi[k] = v2f{b0.z, b1.z};
            b.ax[k] = v2f{0.f, 0.f}; b.ay[k] = v2f{0.f, 0.f}; b.az[k] = v2f{0.f, 0.f};
        }
        __syncthreads();                                   // (the space is written again only by the closing combination)
    } else {
#pragma unroll
        for (int k = 0; k < SYM_K2; k++) {
            const float4 b0 = body(a * SYM_IB + (2 * k) * 64u + lane), b1 = body(a * SYM_IB + (2 * k + 1) * 64u + lane);
            b.xi[k] = v2f{b0.x, b1.x}; b.yi[k] = v2f{b0.y, b1.y}; b.zi[k] = v2f{b0.z, b1.z};
            b.ax[k] = v2f{0.f, 0.f}; b.ay[k] = v2f{0.f, 0.f}; b.az[k] = v2f{0.f, 0.f};
        }
        first_piece();
    }
    asm volatile("" :: "v"(pn.x), "v"(pn.y), "v"(pn.z));   // (the first piece's bodies are waited for HERE: see the note in the loop)
    if (p.timeline) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the I-block and the first J-block have arrived
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tl_loop));
    }
    for (uint32_t t = t0; t < t1;) {
        const uint32_t k0 = t & 63u, steps = min(64u - k0, t1 - t), jb_cur = jb, d_cur = d, g_cur = g;
        float xj = pn.x, yj = pn.y, zj = pn.z;
        t += steps;
        if (t < t1) {
            uint32_t lf = lane;
            asm volatile("" : "+v"(lf));                   // (as below: keep the load's address out of the loops' registers)
            meeting(t >> 6, jb, d, g);
            pn = body_j(jb * 64u + lf);
        }
        // (Alternatives to moving the position, measured on one box each: a wave-private LDS copy of the J-block
        //  read with one ds_read_b128 per step, also one step ahead: 2-3 % slower; re-reading body (lane + k) % 64
        //  from global memory every step, fetched one step ahead: 9 % slower.)
        {
            // ONE loop for every kind of meeting.  The block against itself (d = 0) needs no reaction -- every ordered pair is met
            // from both sides anyway -- but runs the same step and drops it: with a second, one-sided loop beside this one the
            // compiler kept two copies of the accumulators (220 register moves per meeting), ran out of registers and serialised
            // the step (36 s_nop, 1436 instead of 1183 cycles); those meetings are 16 of ~530, 6 packed fma dearer each.
            v2f bx = v2f{0.f, 0.f}, by = v2f{0.f, 0.f}, bz = v2f{0.f, 0.f};
            // two steps per trip, the travelling position alternating between two register sets: the move's result IS the next
            // step's operand (one step per trip needs three register copies per step to carry it round the loop: 2 % of the VALU work)
            uint32_t k = 0;
#pragma nounroll
            for (; k + 2u <= steps; k += 2u) {
                const float x1 = lane_next(xj, next), y1 = lane_next(yj, next), z1 = lane_next(zj, next);
                sym_step(b, xj, yj, zj, soft2, bx, by, bz);
                bx.x = lane_next(bx.x, next); by.x = lane_next(by.x, next); bz.x = lane_next(bz.x, next);
                bx.y = lane_next(bx.y, next); by.y = lane_next(by.y, next); bz.y = lane_next(bz.y, next);
                xj = lane_next(x1, next); yj = lane_next(y1, next); zj = lane_next(z1, next);
                sym_step(b, x1, y1, z1, soft2, bx, by, bz);
                bx.x = lane_next(bx.x, next); by.x = lane_next(by.x, next); bz.x = lane_next(bz.x, next);
                bx.y = lane_next(bx.y, next); by.y = lane_next(by.y, next); bz.y = lane_next(bz.y, next);
            }
            if (k < steps) {
                sym_step(b, xj, yj, zj, soft2, bx, by, bz);
                bx.x = lane_next(bx.x, next); by.x = lane_next(by.x, next); bz.x = lane_next(bz.x, next);
                bx.y = lane_next(bx.y, next); by.y = lane_next(by.y, next); bz.y = lane_next(bz.y, next);
            }
            // The next piece's bodies were fetched a whole piece ago: take them NOW, before this piece's row is stored.  Left
            // to the compiler the wait sits at the head of the next piece as s_waitcnt vmcnt(0) -- loads and stores share the
            // counter, so every piece would also wait for the row store just issued (1-2 us each; SQ_WAIT_ANY was 11 % of
            // the wave-cycles).
            asm volatile("" :: "v"(pn.x), "v"(pn.y), "v"(pn.z));
            if (d_cur == 0u) continue;
            const float fx = bx.x + bx.y, fy = by.x + by.y, fz = bz.x + bz.y;
            // (the row addresses are formed HERE, once per meeting, from a lane id the compiler cannot see through: hoisted out
            //  of the loop they would sit in registers across the steps, and the loop has none to spare -- they were spilled)
            uint32_t ln = lane;
            asm volatile("" : "+v"(ln));
            if (steps == 64u) {
                // 64 moves: every body is back in its home lane with its complete reaction from this I-block
                store_row(p.brow + brow_row(jb_cur, g_cur) * 64u + ln, fx, fy, fz, p.row_wt);
            } else {
                // part of a meeting: its LAST 64 - k0 steps (the first k0 ran in the previous wave; the bodies are home) go to
                // this wave's slot [1], its FIRST `steps` steps (the next wave runs the rest; this lane holds body
                // (lane + steps) % 64) to slot [0]; the closing section puts the meeting together
                const uint32_t which = k0 != 0u ? 1u : 0u, home = (ln + (k0 != 0u ? 0u : steps)) & 63u;
                edge[which][w][0][home] = fx; edge[which][w][1][home] = fy; edge[which][w][2][home] = fz;
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
    unsigned long long tl_done = 0;
    if (p.timeline) {
        asm volatile("" :: "v"(b.ax[0]), "v"(b.ay[0]), "v"(b.az[0]));
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tl_done));
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
    SymRow *row = p.arow + ((size_t)la * p.parts + s) * SYM_IB;
    for (uint32_t e = threadIdx.x; e < SYM_IB; e += 64u * WAVES) {
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ww++) { ax += comb[ww][0][e]; ay += comb[ww][1][e]; az += comb[ww][2][e]; }
        store_row(row + e, ax, ay, az, p.row_wt);
    }
    // a symmetric meeting this wave's range STARTED in: cut between wave w - 1 and this wave -- first steps + last steps make
    // its row; cut between the previous WORKGROUP and this one (w = 0) -- the last steps are this workgroup's head row
    if ((t0 & 63u) != 0u && t0 < t1) {
        uint32_t jbs, ds, gs;
        meeting(t0 >> 6, jbs, ds, gs);
        if (ds != 0u) {
            if (w == 0u) {
                store_row(p.brow1 + ((size_t)la * p.parts + s) * 64u + lane, edge[1][0][0][lane], edge[1][0][1][lane], edge[1][0][2][lane], p.row_wt);
            } else {
                const float fx = edge[0][w - 1u][0][lane] + edge[1][w][0][lane];
                const float fy = edge[0][w - 1u][1][lane] + edge[1][w][1][lane];
                const float fz = edge[0][w - 1u][2][lane] + edge[1][w][2][lane];
                store_row(p.brow + brow_row(jbs, gs) * 64u + lane, fx, fy, fz, p.row_wt);
            }
        }
    }
    // a symmetric meeting the workgroup's LAST wave ended in: the next workgroup runs the rest, its first steps are the row
    if (w == WAVES - 1u && (t1 & 63u) != 0u && t0 < t1) {
        uint32_t jbs, ds, gs;
        meeting(t1 >> 6, jbs, ds, gs);
        if (ds != 0u)
            store_row(p.brow + brow_row(jbs, gs) * 64u + lane, edge[0][w][0][lane], edge[0][w][1][lane], edge[0][w][2][lane], p.row_wt);
    }
    if (p.timeline) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows have left the wave
        unsigned long long tl_exit;
        uint32_t hw, xcc;
        asm volatile("s_memrealtime %0\n s_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n s_waitcnt lgkmcnt(0)"
                     : "=s"(tl_exit), "=s"(hw), "=s"(xcc));
        if (lane == 0) {
            unsigned long long *o = p.timeline + 6u * (((size_t)la * p.parts + s) * WAVES + w);
            o[0] = tl_entry; o[1] = tl_loop; o[2] = tl_done; o[3] = tl_exit; o[4] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            o[5] = (unsigned long long)(t1 - t0);
        }
    }
}

namespace {
// the partner distance group under which I-block a meets I-block b symmetrically (b's bodies travelling), 0 if it does not:
// the schedule of force_sym_kernel
__device__ __forceinline__ uint32_t sym_group(uint32_t a, uint32_t b, uint32_t nb, uint32_t half)
{
    const uint32_t d = b >= a ? b - a : b + nb - a, D = (nb - 1u) / 2u;
    return (d >= 1u && d <= D) ? d : (half && d == half && sym_runs_half(a, half)) ? D + 1u : 0u;
}
}  // namespace

// One thread per body: what the earlier windows of this step summed (if any), the frozen bodies' rows (a partially active step's first window), the rows of its I-block (role i) in
// ascending part order, then the rows of its J-block (role j) in ascending group order -- a meeting's row, then its
// head row when the meeting was cut between two workgroups -- and, in the step's last window, mass, kick, damp,
// drift (hlsl:103-108).
__global__ __launch_bounds__(256) void sym_reduce_integrate_kernel(const SymArgs p)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= p.n_integrate) return;                        // (bodies past num_active exert force but are not advanced: Compute.cpp:1041)
    const uint32_t a = i / SYM_IB, jb = i >> 6;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (p.acc_in) { const float4 v = p.acc_in[i]; ax = v.x; ay = v.y; az = v.z; }
    if (p.extra) {
        // a partially active step: what the FROZEN bodies do to this one -- the partial rows of the one-sided launch in front of
        // this step's symmetric launches, in ascending row order (eight loads in flight)
        const float4 *xr = p.extra + i;
        uint32_t s = 0;
        for (; s + 8u <= p.extra_rows; s += 8u) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = xr[(size_t)(s + u) * p.extra_stride];
#pragma unroll
            for (int u = 0; u < 8; u++) { ax += v[u].x; ay += v[u].y; az += v[u].z; }
        }
        for (; s < p.extra_rows; s++) {
            const float4 v = xr[(size_t)s * p.extra_stride];
            ax += v.x; ay += v.y; az += v.z;
        }
    }
    // The launch is ONE wave per SIMD walking a chain of dependent memory round trips (table lookup -> row loads -> sums), so what it
    // costs is the NUMBER of trips (round 4: two per batch of eight groups + the a-rows + the state: 12 at 65 536 bodies, 10 us at 3.1
    // TB/s).  Round 5: everything that depends on nothing goes out first (the body's state, the first batch's lookups, the a-rows),
    // batches are 16 groups, and the NEXT batch's lookups are in flight with this batch's rows -- 4 trips.  The SUMS are taken in
    // exactly the order they always were (the order-matched oracle does not change).
    const bool last = p.acc_out == nullptr;
    float4 pos = make_float4(0.f, 0.f, 0.f, 0.f);
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (last) {
        pos = p.pos_old[i];
        const float *v = p.vel_old + 3 * (size_t)i;
        vx = v[0]; vy = v[1]; vz = v[2];
    }
    // the symmetric groups of this window; group g's row exists when some block meets this one under g: always for
    // g <= D, for the half-ring group D + 1 only if this block's half-ring PARTNER runs the pair's meetings (sym_runs_half)
    const uint32_t D = (p.nb - 1u) / 2u, gs0 = p.g0 ? p.g0 : 1u;
    const uint32_t gend = (p.g1 == D + 2u && sym_runs_half(a, p.half_d)) ? D + 1u : p.g1;
    const SymRow *br = p.brow + (size_t)jb * p.brows * 64u + (i & 63u);
    // split table of the block that ran a meeting: set = class (+ 2 * (block mod 8) with XCD-weighted parts)
    const uint32_t *splits = p.tab + p.sets * (p.nwaves + 1u);
    auto split_of = [&](uint32_t blk) { return splits + (size_t)((sym_runs_half(blk, p.half_d) ? 0u : 1u) + (p.sets > 2u ? 2u * (blk & 7u) : 0u)) * p.max_meetings; };
    const uint32_t t = jb % SYM_JPI;
    constexpr uint32_t GB = 16u;                           // groups per batch
    uint32_t sp[GB], apv[GB];
    auto lookup = [&](uint32_t g, uint32_t (&spo)[GB], uint32_t (&apo)[GB]) {
#pragma unroll
        for (uint32_t u = 0; u < GB; u++) {
            const uint32_t gu = g + u;
            const bool live = gu < gend;
            const uint32_t d = gu <= D ? gu : p.half_d;
            apo[u] = a >= d ? a - d : a + p.nb - d;                                // the I-block that ran this meeting
            spo[u] = live ? split_of(apo[u])[(gu - p.g0) * SYM_JPI + t] : 0xffffffffu;
        }
    };
    if (gs0 < gend) lookup(gs0, sp, apv);
    const SymRow *ar = p.arow + (size_t)a * p.parts * SYM_IB + (i - a * SYM_IB);
    uint32_t s = 0;
    for (; s + 8u <= p.parts; s += 8u) {                   // 8 loads in flight, summed in ascending order
        SymRow v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = ar[(size_t)(s + u) * SYM_IB];
#pragma unroll
        for (int u = 0; u < 8; u++) { ax += v[u].x; ay += v[u].y; az += v[u].z; }
    }
    if (s < p.parts) {                                     // (the usual case: 4 parts -- all in flight together)
        SymRow v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) v[u] = ar[(size_t)(s + u < p.parts ? s + u : s) * SYM_IB];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++)
            if (s + u < p.parts) { ax += v[u].x; ay += v[u].y; az += v[u].z; }
    }
    for (uint32_t g = gs0; g < gend; g += GB) {            // 16 meetings in flight, summed in ascending order
        SymRow v[GB], h[GB];
#pragma unroll
        for (uint32_t u = 0; u < GB; u++) {
            // (no branch around a load: a lane without a row / without a head row re-reads a row that IS there and drops it --
            //  a branch per load serialised them: the launch took 18.7 instead of 13.9 us at 65 536 bodies)
            const bool live = g + u < gend;
            const SymRow *vsrc = br + (size_t)((live ? g + u : gs0) - gs0) * 64u;
            const SymRow *hsrc = sp[u] != 0xffffffffu ? p.brow1 + ((size_t)apv[u] * p.parts + sp[u]) * 64u + (i & 63u) : vsrc;
            v[u] = *vsrc;
            h[u] = *hsrc;
        }
        uint32_t spn[GB], apn[GB];
        const bool more = g + GB < gend;
        if (more) lookup(g + GB, spn, apn);                // the next batch's lookups travel with this batch's rows
#pragma unroll
        for (uint32_t u = 0; u < GB; u++) {
            if (!(g + u < gend)) v[u] = SymRow{0.f, 0.f, 0.f};
            if (sp[u] == 0xffffffffu) h[u] = SymRow{0.f, 0.f, 0.f};
        }
#pragma unroll
        for (uint32_t u = 0; u < GB; u++) {
            ax += v[u].x; ay += v[u].y; az += v[u].z;
            ax += h[u].x; ay += h[u].y; az += h[u].z;
        }
        if (more) {
#pragma unroll
            for (uint32_t u = 0; u < GB; u++) { sp[u] = spn[u]; apv[u] = apn[u]; }
        }
    }
    if (!last) { p.acc_out[i] = make_float4(ax, ay, az, 0.f); return; }
    ax *= p.mass; ay *= p.mass; az *= p.mass;
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

}  // namespace

// grid <= the workgroups the device holds at once (they wait for the peers' rows; a rank's waiting workgroups must not keep its unsent ones out)   block = 256
// One launch does the reaction exchange, the integration and the exchange of the new positions.
//  (1) SEND: for every rank q this rank produced reactions for and every body of q: the rows of this rank's
//      I-blocks that met the body's block (a meeting's row, then its head row if it was cut between two workgroups),
//      added in ascending block order, stored as ONE float4 into rank q's receive region (row [this rank]) with a
//      system-scope write-through store -- over xGMI when q is another GPU; the unused .w carries the row's tag (see (3)),
//      which the receiver checks.  Once acknowledged (vmcnt(0)) the stores are in q's memory: no cache write-back is
//      owed (a release fence here would write back the whole L2, full of this step's rows: measured 10+ us per step).
//  (2) OWN ROWS, before anything is waited for: G threads per body (a rank's slice is small -- 8192 bodies at
//      65 536 / 8 -- so one thread per body would leave the rows' loads latency-bound): thread (body, g) adds the
//      a-rows of parts [g P/G, (g+1) P/G) in ascending order.
//  (3) ARRIVAL.  poll_rows (round 4, the default): there is no arrival step -- every row validates ITSELF: its .w is
//      sym_row_tag(x, y, z, exchange number), the sender stores it and goes on (no acknowledgement wait, no flag), and in (4) the
//      receiving thread re-reads its body's rows (bounded) until every one carries the tag its contents demand.  One trip through
//      memory instead of three; a torn or stale row cannot pass.  The flag forms of round 3, kept for the A/B: chunk_flags != 0 --
//      every workgroup, its sends acknowledged (vmcnt(0)), stores a flag per (destination, 256-body chunk) it sent and lanes
//      0 .. world-1 of a workgroup's first wave wait (bounded) for the senders' flags of the chunk it integrates next;
//      chunk_flags == 0 -- the last workgroup through a ticket stores ONE flag per destination.
//  (4) INTEGRATE: thread (body, 0) adds the G sums in ascending g, then the rows received, nearest sender first
//      (this rank, rank - 1, rank - 2, ...), then mass, kick, damp, drift (hlsl:103-108) -- a fixed order throughout,
//      so the replicas stay bit-identical.  The new position is stored write-through.
//  (5) POSITIONS (pos_step != 0; else they travel in p2p_gather_kernel): every workgroup, its stores acknowledged, adds its share
//      to this rank's position counter at every peer.  PULL form: every workgroup waits
//      for the peers' counters and pulls its share of their slices with cache-bypassing system-scope loads.  PUSH form
//      (gather algorithm 5): (4) has already stored every new position into every peer's replica as well, the counter
//      says so, and the launch ends here -- the wait moves to the head of the peers' next force launch.
// PARTIAL: a partially active step (p.active < the job's bodies; enqueue_sym_shard_split) -- frozen bodies are skipped, the frozen rows added, count_active
// bodies integrated.  The all-active step is instantiated WITHOUT any of it: its code is round 5's (same box: exchange launch 12.0 - 12.1 us
// either way; with the checks compiled in unconditionally 12.3).
template <int G, bool PARTIAL>
__global__ __launch_bounds__(256) void sym_shard_exchange_kernel(const SymShardArgs p)
{
    const uint32_t bid = blockIdx.x, nblk = gridDim.x;
    const uint32_t count_active = PARTIAL ? p.count_active : p.count;
    constexpr uint32_t B = 256u / G;                       // bodies per workgroup and pass
    __shared__ uint32_t ok;
    __shared__ float part[G][3][B];

    // diagnostic launches only (MAPN_STAMP_DUMP): wall-clock stamps of the workgroup's phases; null otherwise
    auto stamp = [&](uint32_t k) {
        if (p.timeline && threadIdx.x == 0) p.timeline[(size_t)bid * 8u + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    // split table of the block that ran a meeting: set = class (+ 2 * (LOCAL block mod 8) with XCD-weighted parts)
    const uint32_t *splits = p.tab + p.sets * (p.nwaves + 1u);
    auto split_of = [&](uint32_t a, uint32_t la) { return splits + (size_t)((sym_runs_half(a, p.half_d) ? 0u : 1u) + (p.sets > 2u ? 2u * (la & 7u) : 0u)) * p.max_meetings; };
    const uint32_t total = p.phase == 2u ? 0u : p.world * p.count;
    for (uint32_t t = bid * 256u + threadIdx.x; t < total; t += nblk * 256u) {
        const uint32_t q = t / p.count, jl = t - q * p.count;
        if (!((p.send_mask >> q) & 1u) || (PARTIAL && t >= p.active)) continue;   // (t is the body's index in the whole job; a frozen body collects nothing)
        const uint32_t b = t / SYM_BLOCK, jb = t >> 6, tt = jb % SYM_JPI;
        const SymRow *rows = p.brow + (size_t)jb * p.nbl * 64u + (t & 63u);
        float fx = 0.f, fy = 0.f, fz = 0.f;
        if (PARTIAL && p.extra) {
            // a PARTIALLY ACTIVE step: what this rank's FROZEN bodies do to body t -- the partial rows of the one-sided launch in front of this
            // one, in ascending row order from zero (eight loads in flight); the reactions of this rank's active blocks follow
            const float4 *xr = p.extra + t;
            uint32_t r = 0;
            for (; r + 8u <= p.extra_rows; r += 8u) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = xr[(size_t)(r + u) * p.extra_stride];
#pragma unroll
                for (int u = 0; u < 8; u++) { fx += v[u].x; fy += v[u].y; fz += v[u].z; }
            }
            for (; r < p.extra_rows; r++) {
                const float4 v = xr[(size_t)r * p.extra_stride];
                fx += v.x; fy += v.y; fz += v.z;
            }
        }
        for (uint32_t la = 0; la < p.nbl; la += 8u) {      // eight meetings in flight, added in ascending block order
            SymRow v[8], h[8];
            uint32_t gg[8], sp[8];
            // first ALL the table lookups of the batch, then all its row loads: the waits count in order, so a lookup issued
            // behind a row load would wait for that row too
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t a = p.a0 + la + u;
                gg[u] = la + u < p.nbl ? sym_group(a, b, p.nb, p.half_d) : 0u;
                sp[u] = split_of(a, la + u)[gg[u] * SYM_JPI + tt];   // (group 0: the block itself, never cut)
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                // no branch around a load (eight branches per batch serialise the loads):
                // a lane without a row / without a head row re-reads a row that is there and drops it
                const bool cut = gg[u] && sp[u] != 0xffffffffu;
                const SymRow *vsrc = rows + (size_t)(gg[u] ? la + u : 0u) * 64u;
                const SymRow *hsrc = cut ? p.brow1 + ((size_t)(la + u) * p.parts + sp[u]) * 64u + (t & 63u) : vsrc;
                v[u] = *vsrc;
                h[u] = *hsrc;
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const bool cut = gg[u] && sp[u] != 0xffffffffu;
                if (!gg[u]) v[u] = SymRow{0.f, 0.f, 0.f};
                if (!cut) h[u] = SymRow{0.f, 0.f, 0.f};
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                fx += v[u].x; fy += v[u].y; fz += v[u].z;
                fx += h[u].x; fy += h[u].y; fz += h[u].z;
            }
        }
        const uint32_t tag = p.poll_rows ? sym_row_tag(__builtin_bit_cast(uint32_t, fx), __builtin_bit_cast(uint32_t, fy), __builtin_bit_cast(uint32_t, fz), p.step) : p.step;
        if (p.corrupt_row && t == (p.rank * p.count + 5u)) fx = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, fx) ^ 1u);   // TEST HOOK: after the tag
        store_sys(p.recv_peer[q] + (size_t)p.send_row * p.count + jl, fx, fy, fz, __builtin_bit_cast(float, tag));
    }
    if (p.phase == 1u) return;                             // PACK: a collective library moves the rows, another launch reduces

    // this rank's own rows for the first pass of (4): nothing here depends on a peer
    const uint32_t bl = threadIdx.x % B, g = threadIdx.x / B;
    auto own_rows = [&](uint32_t il, float &ax, float &ay, float &az) {
        ax = ay = az = 0.f;
        if (il >= count_active) return;
        const uint32_t la = il / SYM_BLOCK;
        const SymRow *ar = p.arow + (size_t)la * p.parts * SYM_BLOCK + (il - la * SYM_BLOCK);
        const uint32_t s0 = (uint32_t)(((uint64_t)p.parts * g) / G), s1 = (uint32_t)(((uint64_t)p.parts * (g + 1u)) / G);
        uint32_t s = s0;
        for (; s + 8u <= s1; s += 8u) {
            SymRow v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = ar[(size_t)(s + u) * SYM_BLOCK];
#pragma unroll
            for (int u = 0; u < 8; u++) { ax += v[u].x; ay += v[u].y; az += v[u].z; }
        }
        for (; s < s1; s++) {
            const SymRow v = ar[(size_t)s * SYM_BLOCK];
            ax += v.x; ay += v.y; az += v.z;
        }
    };
    float ax, ay, az;
    stamp(1);                                              // sends issued
    own_rows(bid * B + bl, ax, ay, az);

    if (!(p.poll_rows && p.phase == 0u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (self-validating rows: nothing is published behind the sends, nobody waits for their acknowledgement here)
    if (p.release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // buffer_wbl2 sc0 sc1 + wait: the sends have LEFT this GPU's L2 (see the note at sync 2)
    __syncthreads();
    stamp(2);                                              // sends acknowledged (flag forms) / issued (self-validating rows), own rows summed
    if (p.phase == 2u || p.poll_rows) {
        if (threadIdx.x == 0) ok = 1u;                     // REDUCE: the rows were delivered in stream order; POLL: every row says itself when it is there
    } else if (p.chunk_flags) {
        // ARRIVAL FLAGS PER CHUNK: this workgroup's sends are acknowledged -- it says so itself, per destination and 256-body
        // chunk (one word each in the destination's uncached region: [sender][chunk]), and a receiver waits only for the
        // chunk it is about to integrate.  No ticket round trip, no workgroup waits for another one of its own rank (so the
        // launch's workgroups need not all be resident for it to make progress), and the first receivers start while the last
        // senders are still at work.
        const uint32_t tasks = p.world * (p.count / 256u), nchunks = p.count / 256u;
        for (uint32_t T = bid + threadIdx.x * nblk; T < tasks; T += 256u * nblk) {
            const uint32_t q = T / nchunks, c = T - q * nchunks;
            if ((p.send_mask >> q) & 1u)
                __hip_atomic_store(p.flags_peer[q] + p.chunk_flags + p.send_row * nchunks + c, p.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (threadIdx.x == 0) ok = 1u;
    } else if (threadIdx.x < 64u) {
        // the last workgroup through the ticket tells every rank this one sent to.  (Every workgroup adding a share to the peers'
        // counters instead -- no ticket round trip -- was 0.8 us SLOWER: 256 system-scope atomics on one uncached word are served
        // one after the other, 3.3 us from the last add to the counter being seen against 1.45 us this way.)
        if (threadIdx.x == 0) {
            const uint32_t prev = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1u == nblk) {
                __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
                for (uint32_t q = 0; q < p.world; q++)
                    if ((p.send_mask >> q) & 1u)
                        __hip_atomic_store(p.flags_peer[q] + SYM_FLAG_BASE + p.rank, p.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        const uint32_t q = threadIdx.x;
        const uint32_t all_good = wait_counters(p.flags_mine + SYM_FLAG_BASE, q, q < p.world && ((p.recv_mask >> q) & 1u), p.step,
                                                p.timeout_ticks, p.status, 1u + q, p.flags_mine + SYM_DEAD_WORD);
        if (threadIdx.x == 0) ok = all_good;
    }
    __syncthreads();
    if (!ok) return;
    if (!p.chunk_flags && !p.poll_rows) stamp(3);          // the peers' rows are here

    for (uint32_t base = bid * B; base < count_active; base += nblk * B) {       // (count_active: this rank's bodies that advance -- all of them, or fewer in a partially active step)
        const uint32_t il = base + bl;
        const bool live = il < count_active;
        if (base != bid * B) own_rows(il, ax, ay, az);
        if (p.chunk_flags && !p.poll_rows && p.phase == 0u && threadIdx.x < 64u) {
            // the senders' flags of THIS chunk (one lane per sender; bounded)
            const uint32_t q = threadIdx.x, nchunks = p.count / 256u;
            const uint32_t all_good = wait_counters(p.flags_mine + p.chunk_flags, q * nchunks + base / 256u, q < p.world && ((p.recv_mask >> q) & 1u), p.step,
                                                    p.timeout_ticks, p.status, 1u + q, p.flags_mine + SYM_DEAD_WORD);
            if (threadIdx.x == 0) ok = all_good;
        }
        __syncthreads();                                   // (the previous pass has read `part`)
        part[g][0][bl] = ax; part[g][1][bl] = ay; part[g][2][bl] = az;
        __syncthreads();
        if (!ok) return;
        if (p.chunk_flags && !p.poll_rows && base == bid * B) stamp(3);    // the peers' rows of the first chunk are here
        if (g != 0u || !live) continue;
        ax = ay = az = 0.f;
#pragma unroll
        for (int gg = 0; gg < G; gg++) { ax += part[gg][0][bl]; ay += part[gg][1][bl]; az += part[gg][2][bl]; }
        // the rows received: uncached region, read past this GPU's caches with system-scope loads (global_load_dwordx2
        // sc0 sc1, issued by the compiler so that it places the waits), all in flight, added nearest sender first
        unsigned long long lo[P2P_MAX_RANKS], hi[P2P_MAX_RANKS];
        uint32_t late = 0u;
        const bool poll = p.poll_rows && p.phase == 0u;
        const uint64_t poll_t0 = poll ? __builtin_amdgcn_s_memrealtime() : 0ull;
        for (;;) {
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)P2P_MAX_RANKS; k++) {
                const uint32_t q = p.rank >= k ? p.rank - k : p.rank + p.world - k;
                const bool need = k < p.world && ((p.recv_mask >> q) & 1u);
                const unsigned long long *src = reinterpret_cast<const unsigned long long *>(p.recv_mine + (size_t)(need ? q : p.rank) * p.count + il);
                lo[k] = need ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
                hi[k] = need ? __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ((unsigned long long)p.step << 32);
            }
            late = 0u;
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)P2P_MAX_RANKS; k++) {
                const uint32_t q = p.rank >= k ? p.rank - k : p.rank + p.world - k;
                const bool need = k < p.world && ((p.recv_mask >> q) & 1u);
                // what the row's .w must be: the exchange number (flag forms: the flag said the row is there) -- or, self-validating
                // rows, the hash of the very words just read
                const uint32_t want = (p.poll_rows && need) ? sym_row_tag((uint32_t)lo[k], (uint32_t)(lo[k] >> 32), (uint32_t)hi[k], p.step) : p.step;
                if ((uint32_t)(hi[k] >> 32) != want) late = 0x100u + k;        // not (yet) this exchange's row, or a torn one
            }
            if (!late || !poll) break;
            // self-validating rows: not all there yet -- read them again (bounded: a sender that never sends is reported, with its place;
            // not at all once an earlier wait or check of this rank has failed)
            if (__hip_atomic_load(p.flags_mine + SYM_DEAD_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
            if (__builtin_amdgcn_s_memrealtime() - poll_t0 > p.timeout_ticks) break;
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)P2P_MAX_RANKS; k++) {
            ax += __builtin_bit_cast(float, (uint32_t)lo[k]);
            ay += __builtin_bit_cast(float, (uint32_t)(lo[k] >> 32));
            az += __builtin_bit_cast(float, (uint32_t)hi[k]);
        }
        if (late) {
            __hip_atomic_store(p.status, late, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (p.phase == 0u) {
                // A rank whose rows never validated must not go on as if they had: until round 4 it integrated the incomplete sum,
                // published the result with valid checksums and counted itself complete -- only IT reported, the peers held bit-identical
                // wrong replicas.  The failing thread POISONS this rank's position counter at every peer (a value 1.5 x 2^30 short of what
                // this publication must reach: no number of shares makes that up, and every later launch of the dead rank poisons again), so
                // the peers' bounded waits for this rank's slice give up too and every rank of the job reports -- and (round 6) the BODY is
                // left alone: not integrated, neither stored here nor pushed, no checksum word for its group (`continue` below; until then it
                // fell through and the incomplete sum landed in every peer's replica behind valid checksums).  Bodies of the slice whose rows
                // DID validate are published as ever (they are right, and were on their way long before this thread gave up); what keeps the
                // peers from consuming the slice is the poisoned counter.  All of it sits in this cold branch (same box: exchange launch
                // 11.98 -> 12.14 us; a workgroup flag that withholds the share instead: 11.80 -> 12.16).
                __hip_atomic_store(p.flags_mine + SYM_DEAD_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (p.pos_step) {
                    const uint32_t poison = p.pos_step * SYM_COUNT_PER_LAUNCH - 0x60000000u;
                    for (uint32_t q = 0; q < p.world; q++)
                        if (p.pull_self ? q == p.rank : q != p.rank)
                            __hip_atomic_store(p.flags_peer[q] + SYM_POS_BASE + p.rank, poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            continue;                                      // nothing of this body is published (phase 2, rows delivered by RCCL: reported, not stored either)
        }
        ax *= p.mass; ay *= p.mass; az *= p.mass;
        const uint32_t i = p.rank * p.count + il;
        const float4 pos = p.pos_old[i];
        const float *v = p.vel_old + 3 * (size_t)i;
        float vx = v[0], vy = v[1], vz = v[2];
        vx = __builtin_fmaf(ax, p.dt, vx) * p.damping;
        vy = __builtin_fmaf(ay, p.dt, vy) * p.damping;
        vz = __builtin_fmaf(az, p.dt, vz) * p.damping;
        const float ox = __builtin_fmaf(vx, p.dt, pos.x), oy = __builtin_fmaf(vy, p.dt, pos.y), oz = __builtin_fmaf(vz, p.dt, pos.z);
        const float ow = __builtin_sqrtf(__builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)));
        store_sys(p.pos_new + i, ox, oy, oz, ow);
        if (p.push) {                                      // ... and into every peer's replica (over xGMI when q is another GPU)
            // TEST HOOK: one bit of one pushed position flipped AFTER the checksum below was formed from the true value
            const float oxp = (p.corrupt_push && il == 0u) ? __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, ox) ^ 1u) : ox;
            for (uint32_t q = 0; q < p.world; q++)
                if (q != p.rank) store_sys(p.pos_peer[q] + i, oxp, oy, oz, ow);
            if (p.pos_sums) {
                // behind the data: one checksum word per 32 bodies, this publication's number mixed in, into every peer's row [this rank]
                // (the receivers re-compute it from what they read: verify_pushed).  The threads here are whole, aligned half-waves.
                uint32_t hs = sym_push_checksum(__builtin_bit_cast(uint32_t, ox), __builtin_bit_cast(uint32_t, oy), __builtin_bit_cast(uint32_t, oz),
                                                __builtin_bit_cast(uint32_t, ow), i);
                hs = xor_reduce32(hs);
                if ((threadIdx.x & 31u) == 0u) {
                    const uint32_t cs = hs ^ sym_push_epoch_mix(p.pos_step);
                    for (uint32_t q = 0; q < p.world; q++)
                        if (q != p.rank)
                            // (loopback timing: every "peer" is this rank -- one row per peer then, so that the stores go to distinct
                            //  addresses as they would on a node instead of queueing up behind each other at ONE uncached word)
                            __hip_atomic_store(p.flags_peer[q] + p.pos_sums + ((p.pos_step & 1u) * p.world + (p.pull_self ? q : p.send_row)) * (p.count / 32u) + il / 32u, cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        float *vo = p.vel_new + 3 * (size_t)i;
        vo[0] = vx; vo[1] = vy; vo[2] = vz;
    }
    stamp(4);                                              // integrated, position stores issued
    if (!p.pos_step) return;

    // (5) this rank's new slice is in memory once every workgroup's stores are acknowledged: then the counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (the system-scope write-through stores are acknowledged once performed; p.release puts a release fence -- buffer_wbl2 +
    //  wait -- in front of the counter as well: belt and braces, +22 us per step measured, off by default: DESIGN 5 "Visibility")
    if (p.release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    stamp(5);                                              // position stores acknowledged
    if (threadIdx.x < 64u) {
        // every workgroup adds ITS share of SYM_COUNT_PER_LAUNCH to this rank's position counter at every peer (fire and forget: the
        // workgroup leaves at once instead of waiting for a ticket to come back; the shares of a launch sum to
        // SYM_COUNT_PER_LAUNCH whatever its grid, a waiter needs launch number x that): the launch ends 2 us earlier
        const uint32_t share = (uint32_t)(((uint64_t)SYM_COUNT_PER_LAUNCH * (bid + 1u)) / nblk) - (uint32_t)(((uint64_t)SYM_COUNT_PER_LAUNCH * bid) / nblk);
        if (threadIdx.x < p.world && (p.pull_self ? threadIdx.x == p.rank : threadIdx.x != p.rank))
            (void)__hip_atomic_fetch_add(p.flags_peer[threadIdx.x] + SYM_POS_BASE + p.rank, share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (pushed positions: the peers' NEXT force launch waits for the counter, nothing to pull -- unless that next launch cannot wait
        //  itself: wait_tail, below)
        if (p.push && !(PARTIAL && p.wait_tail)) { stamp(6); return; }
        const uint32_t q = threadIdx.x;
        const uint32_t all_good = wait_counters(p.flags_mine + SYM_POS_BASE, p.pull_self ? p.rank : q, q < p.world && q != p.rank, p.pos_step * SYM_COUNT_PER_LAUNCH,
                                                p.timeout_ticks, p.status, 1u + q, p.flags_mine + SYM_DEAD_WORD);
        if (threadIdx.x == 0) ok = all_good;
    }
    if (p.push && !(PARTIAL && p.wait_tail)) return;
    __syncthreads();
    if (!ok) return;
    if (PARTIAL && p.push) {
        // WAIT TAIL (a partially active step on a rank that owns frozen bodies): its next launch is the ONE-SIDED one over its frozen bodies,
        // which reads the peers' new positions and cannot wait for them itself -- until round 6's last day a separate wait launch in front of
        // it (4.5 us per step of a frozen rank).  The wait moves HERE, to the tail of this step's exchange launch (every workgroup, like the
        // pulled form; the dependency is the same, a launch less), together with the check of what the peers pushed: every wave its share.
        if (p.pos_sums)
            verify_pushed(p.pos_new, p.flags_mine + p.pos_sums, p.pos_step, p.count, p.world, p.rank, p.pull_self, bid * 4u + (threadIdx.x >> 6), nblk * 4u, threadIdx.x & 63u,
                          p.status, p.flags_mine + SYM_DEAD_WORD, p.active);
        return;
    }
    // the peers' slices: 16 bytes per lane per access, eight in flight, past this GPU's caches (a line of q's buffer
    // cached here two steps ago must not be returned)
    const uint32_t others = (p.world - 1u) * p.count;
    for (uint32_t e = bid * 256u + threadIdx.x; e < others; e += nblk * 256u * 8u) {
        unsigned long long lo[8], hi[8];
        uint32_t at[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t eu = e + u * nblk * 256u;
            const bool in = eu < others;
            const uint32_t k = in ? eu / p.count : 0u, q = k < p.rank ? k : k + 1u;        // skip self
            const bool live = in && (!PARTIAL || q * p.count + (eu - k * p.count) < p.active);   // (a partially active step: the frozen bodies did not move)
            at[u] = live ? q * p.count + (eu - k * p.count) : 0xffffffffu;
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(p.pos_peer[q] + (live ? at[u] : 0u));
            lo[u] = live ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
            hi[u] = live ? __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++)
            if (at[u] != 0xffffffffu) {
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.pos_new + at[u]);
                dst[0] = lo[u]; dst[1] = hi[u];
            }
    }
}

// stream operation (gather algorithm 5, wherever something other than the next sharded symmetric force launch is about to
// read the replica): one wave waits (bounded) until every peer has pushed its slice
// ... and (verify_sums != null) checks their slices against the pushers' checksums like the force launch does; grid = waves that share the check
__global__ __launch_bounds__(64) void p2p_wait_kernel(const uint32_t *counters, uint32_t need, uint32_t world, uint32_t rank, uint32_t self,
                                                      uint64_t timeout_ticks, uint32_t *status, uint32_t *dead, const float4 *replica, const uint32_t *verify_sums,
                                                      uint32_t verify_epoch, uint32_t count, uint32_t verify_active)
{
    const uint32_t q = threadIdx.x;
    const uint32_t good = wait_counters(counters, self ? rank : q, q < world && q != rank, need, timeout_ticks, status, 1u + q, dead);
    if (verify_sums && good) verify_pushed(replica, verify_sums, verify_epoch, count, world, rank, self, blockIdx.x, gridDim.x, threadIdx.x, status, dead, verify_active);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

hipError_t launch_p2p_wait(const uint32_t *counters, uint32_t need, uint32_t world, uint32_t rank, uint32_t self, uint64_t timeout_ticks,
                           uint32_t *status, uint32_t *dead, const float4 *replica, const uint32_t *verify_sums, uint32_t verify_epoch, uint32_t count,
                           uint32_t verify_active, hipStream_t st)
{
    const uint32_t groups = verify_sums ? (world - 1u) * (count / 32u) : 0u;
    const uint32_t grid = std::max(1u, std::min(64u, groups / 2u));
    hipLaunchKernelGGL(p2p_wait_kernel, dim3(grid), dim3(64), 0, st, counters, need, world, rank, self, timeout_ticks, status, dead, replica, verify_sums, verify_epoch, count, verify_active);
    return hipGetLastError();
}

hipError_t launch_force_sym(const SymArgs &a, uint32_t waves, hipStream_t st)
{
    const dim3 grid(a.shard_nbl ? a.shard_nbl : a.nb, a.parts);
    if (waves == 4) hipLaunchKernelGGL((force_sym_kernel<4>), grid, dim3(256), 0, st, a);
    else if (waves == 8) hipLaunchKernelGGL((force_sym_kernel<8>), grid, dim3(512), 0, st, a);
    else return hipErrorInvalidConfiguration;
    return hipGetLastError();
}

namespace {
// enough threads to keep the rows' loads in flight: 8 per body up to 16 384 bodies, 4 up to 65 536, else 1
int exchange_threads_per_body(uint32_t count) { return count <= 16384u ? 8 : count <= 65536u ? 4 : 1; }
}  // namespace

// Which compute units actually take this process's workgroups (a CU mask, a partition mode or a reservation can leave
// fewer than the device properties promise): every probe wave marks the unit it runs on.
__global__ __launch_bounds__(64) void cu_probe_kernel(uint32_t *bitmap)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    for (int k = 0; k < 8; k++) __builtin_amdgcn_s_sleep(127);            // stay a moment, so that the dispatcher has to use every unit
    const uint32_t id = ((((xcc & 15u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u) + ((hw >> 8) & 15u);
    if (threadIdx.x == 0) atomicOr(bitmap + id / 32u, 1u << (id % 32u));
}

int probe_active_compute_units(hipStream_t st)
{
    uint32_t *bits = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&bits), 512) != hipSuccess) { (void)hipGetLastError(); return 0; }
    uint32_t host[128] = {};
    int n = 0;
    if (hipMemsetAsync(bits, 0, 512, st) == hipSuccess) {
        hipLaunchKernelGGL(cu_probe_kernel, dim3(16384), dim3(64), 0, st, bits);
        if (hipMemcpyAsync(host, bits, 512, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
            for (uint32_t w : host) n += __builtin_popcount(w);
    }
    (void)hipGetLastError();
    (void)hipFree(bits);
    return n;
}

// The exchange kernel's workgroups wait for each other (tickets) and for the peers: all of them must be resident at
// once.  What ONE unit can hold is asked of the runtime (registers, LDS), how many units there are is probed
// (probe_active_compute_units), and half of the product is left to whatever else is running (a queued fence wait, the comm
// stream, another rank's kernels on a shared device).
uint32_t sym_shard_exchange_resident_workgroups(uint32_t count, int cus)
{
    int per_cu = 0;
    const int g = exchange_threads_per_body(count);
    // (the PARTIAL instantiation: it holds no more registers than the plain one would allow -- the smaller of the two answers)
    int plain = 0, part = 0;
    hipError_t e = g == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&plain, sym_shard_exchange_kernel<8, false>, 256, 0)
                 : g == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&plain, sym_shard_exchange_kernel<4, false>, 256, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&plain, sym_shard_exchange_kernel<1, false>, 256, 0);
    if (e == hipSuccess)
        e = g == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&part, sym_shard_exchange_kernel<8, true>, 256, 0)
          : g == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&part, sym_shard_exchange_kernel<4, true>, 256, 0)
                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&part, sym_shard_exchange_kernel<1, true>, 256, 0);
    per_cu = std::min(plain, part);
    if (e != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
    return (uint32_t)std::max(1, per_cu * cus / 2);
}

hipError_t launch_sym_shard_exchange(const SymShardArgs &a, uint32_t max_workgroups, hipStream_t st)
{
    const int g = exchange_threads_per_body(a.count);
    const uint32_t want = (a.count * (uint32_t)g + 255u) / 256u, grid = std::max(1u, std::min(max_workgroups, want));
    const bool partial = a.active < a.world * a.count;     // (phase 1 / 2 launches of the RCCL form: always all bodies)
    if (partial) {
        if (g == 8) hipLaunchKernelGGL((sym_shard_exchange_kernel<8, true>), dim3(grid), dim3(256), 0, st, a);
        else if (g == 4) hipLaunchKernelGGL((sym_shard_exchange_kernel<4, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((sym_shard_exchange_kernel<1, true>), dim3(grid), dim3(256), 0, st, a);
    } else if (g == 8) hipLaunchKernelGGL((sym_shard_exchange_kernel<8, false>), dim3(grid), dim3(256), 0, st, a);
    else if (g == 4) hipLaunchKernelGGL((sym_shard_exchange_kernel<4, false>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((sym_shard_exchange_kernel<1, false>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_sym_reduce(const SymArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(sym_reduce_integrate_kernel, dim3((a.n_integrate + 255u) / 256u), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace mapn
