// mapn_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the n-body step.
//
// What the reference dispatches: CSMain, reference/Particles/nBodyGravityCS.hlsl:85-109, one
// thread per body in groups of 64 (Compute.cpp:1041).  The all-pairs force is that kernel's
// integrator (:103-108) fed by the sum of bodyBodyInteraction (:44-57) over all bodies of the
// old position buffer.
//
// Design:
//   * a lane owns 2*K2 bodies i (registers), two per packed-fp32 op; a wave walks a contiguous
//     j-chunk; the j-range of a launch is split S = gridDim.y * WAVES ways so that N = 65 536
//     still fills 1024 SIMDs with several rounds of 8 waves each;
//   * j-bodies reach the lanes either through the scalar cache (s_load_dwordx8 -> SGPR-pair
//     operands of the packed ops, op_sel picking x/y/z) or through a wave-private,
//     double-buffered LDS tile (coalesced global_load_dwordx4 -> ds_write_b64 + ds_write_b32
//     into an (x0,y0,x1,y1)/(z0..z3) split layout, then broadcast ds_read_b128);
//   * per two pairs: 3 v_pk_add, 3 v_pk_fma, 2 v_rsq_f32, 2 v_pk_mul, 3 v_pk_fma -- 13 VALU
//     instructions, mass hoisted out of the sum (measured: SQ_INSTS_VALU = 13 per 2 pairs);
//   * the chunk sums of a body are combined in FIXED ascending order (LDS inside a workgroup,
//     one scratch row per workgroup row across workgroups) -- no float atomics, so a run is
//     bit-reproducible;
//   * the kick-drift integrator runs inside the force launch: directly when one workgroup sees all
//     of a body's chunks (EPI_FUSED), otherwise by the LAST workgroup to arrive at the i-tile's
//     ticket (EPI_TICKET: rows published write-through, summed in ascending row order, so the bits
//     equal those of the two-kernel form EPI_ROWS + reduce_integrate_kernel, kept for A/B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mapn_kernels.h"

namespace mapn {

// ---------------------------------------------------------------------------------------------
// shared device helpers

typedef float v2f __attribute__((ext_vector_type(2)));

// accumulators of TWO bodies i, one per half of a 64-bit register pair, so that every VALU op of
// the pair term except v_rsq_f32 is a packed-fp32 instruction (v_pk_add/mul/fma_f32).
// Measured on MI355X (tools/ubench.hip, profiles/r01_ubench.txt): the packed form of the
// 12-op pair term sustains 4.9e12 pairs/s against 3.8e12 for scalar fp32 ops.
struct Acc2 { v2f x, y, z; };

// one softened pair term for two bodies i against one body j, without the mass factor:
//   acc += r * (|r|^2 + soft2)^(-3/2)         (nBodyGravityCS.hlsl:46-56)
// mass * particles (hlsl:54) is applied once after the sum.  13 VALU instructions per 2 pairs:
// 3 v_pk_add (x_j broadcast by op_sel), 3 v_pk_fma, 2 v_rsq, 2 v_pk_mul, 3 v_pk_fma.
__device__ __forceinline__ void pair_term2(Acc2 &a, v2f xi, v2f yi, v2f zi, float xj, float yj,
                                           float zj, v2f soft2)
{
    const v2f dx = xj - xi;
    const v2f dy = yj - yi;
    const v2f dz = zj - zi;
    v2f d = __builtin_elementwise_fma(dx, dx, soft2);
    d = __builtin_elementwise_fma(dy, dy, d);
    d = __builtin_elementwise_fma(dz, dz, d);
    v2f inv;
    inv.x = __builtin_amdgcn_rsqf(d.x);
    inv.y = __builtin_amdgcn_rsqf(d.y);
    const v2f inv3 = inv * inv * inv;
    a.x = __builtin_elementwise_fma(dx, inv3, a.x);
    a.y = __builtin_elementwise_fma(dy, inv3, a.y);
    a.z = __builtin_elementwise_fma(dz, inv3, a.z);
}

// r = (x_j, x_j) - x_i with x_j taken from the low (SEL 0) or high (SEL 1) half of an aligned
// 64-bit register pair: one v_pk_add_f32 whose op_sel broadcasts the half and whose neg bits
// negate x_i.  Spelled as asm for the LDS path only: left to itself hipcc copies every second
// broadcast operand into a fresh pair first (v_mov_b32; +10 % VALU instructions measured with
// SQ_INSTS_VALU, profiles/r01_ab_lds_vs_sgpr_pmc.txt).
template <int SEL>
__device__ __forceinline__ v2f bcast_sub(v2f pair, v2f xi)
{
    v2f r;
    if constexpr (SEL == 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(pair), "v"(xi));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(pair), "v"(xi));
    return r;
}

// pair term with the j-body given as register pairs: xy = (x_j, y_j), zp = the pair holding z_j
template <int ZSEL>
__device__ __forceinline__ void pair_term2_pairs(Acc2 &a, v2f xi, v2f yi, v2f zi, v2f xy, v2f zp, v2f soft2)
{
    const v2f dx = bcast_sub<0>(xy, xi);
    const v2f dy = bcast_sub<1>(xy, yi);
    const v2f dz = bcast_sub<ZSEL>(zp, zi);
    v2f d = __builtin_elementwise_fma(dx, dx, soft2);
    d = __builtin_elementwise_fma(dy, dy, d);
    d = __builtin_elementwise_fma(dz, dz, d);
    v2f inv;
    inv.x = __builtin_amdgcn_rsqf(d.x);
    inv.y = __builtin_amdgcn_rsqf(d.y);
    const v2f inv3 = inv * inv * inv;
    a.x = __builtin_elementwise_fma(dx, inv3, a.x);
    a.y = __builtin_elementwise_fma(dy, inv3, a.y);
    a.z = __builtin_elementwise_fma(dz, inv3, a.z);
}

// nBodyGravityCS.hlsl:103-108: kick, damp, drift; w = |accel|
// (STREAM: the velocity is read and the new state written with the non-temporal hint -- a state far larger than the 256 MiB
//  Infinity Cache passes through once per step, nothing of it is there again when the next step comes: central_well_kernel)
template <bool STREAM = false>
__device__ __forceinline__ void integrate_store(const StepArgs &p, uint32_t i, float4 pos,
                                                float ax, float ay, float az)
{
    const float *v = p.vel_old + 3 * (size_t)i;
    float vx, vy, vz;
    if constexpr (STREAM) { vx = __builtin_nontemporal_load(v); vy = __builtin_nontemporal_load(v + 1); vz = __builtin_nontemporal_load(v + 2); }
    else { vx = v[0]; vy = v[1]; vz = v[2]; }
    vx = __builtin_fmaf(ax, p.dt, vx) * p.damping;
    vy = __builtin_fmaf(ay, p.dt, vy) * p.damping;
    vz = __builtin_fmaf(az, p.dt, vz) * p.damping;
    float4 o;
    o.x = __builtin_fmaf(vx, p.dt, pos.x);
    o.y = __builtin_fmaf(vy, p.dt, pos.y);
    o.z = __builtin_fmaf(vz, p.dt, pos.z);
    o.w = __builtin_sqrtf(__builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)));
    typedef float f4s __attribute__((ext_vector_type(4)));
    float *vo = p.vel_new + 3 * (size_t)i;
    if constexpr (STREAM) {
        const f4s w = {o.x, o.y, o.z, o.w};
        __builtin_nontemporal_store(w, reinterpret_cast<f4s *>(p.pos_new + i));
        __builtin_nontemporal_store(vx, vo); __builtin_nontemporal_store(vy, vo + 1); __builtin_nontemporal_store(vz, vo + 2);
        return;
    }
    if (p.flow_arrived) {
        // flow mode: peers pull this slice over xGMI while the launch is still running -- the new
        // position goes write-through to memory at system scope (one global_store_dwordx4 sc0 sc1)
        const f4s v = {o.x, o.y, o.z, o.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p.pos_new + i), "v"(v) : "memory");
    } else {
        p.pos_new[i] = o;
    }
    vo[0] = vx; vo[1] = vy; vo[2] = vz;
}

// Blocks are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an L2).  Remap the linear
// block id so that the blocks sharing an XCD walk the SAME j-chunk rows: each XCD's 4 MiB L2
// then holds 1/8 of the j-range instead of all of it (matters from N = 262 144 up, where the
// position buffer no longer fits one L2).  Speed only, never correctness.
__device__ __forceinline__ void xcd_remap(uint32_t &bx, uint32_t &by, uint32_t enabled)
{
    const uint32_t gx = gridDim.x, gy = gridDim.y;
    if (enabled && (gy & 7u) == 0u) {
        const uint32_t lin = blockIdx.y * gx + blockIdx.x;
        const uint32_t xcd = lin & 7u, slot = lin >> 3;
        const uint32_t rows_per_xcd = gy >> 3;
        bx = slot / rows_per_xcd;
        by = xcd * rows_per_xcd + (slot - bx * rows_per_xcd);
    } else {
        bx = blockIdx.x; by = blockIdx.y;
    }
}

// tiles of 64 j-bodies [t0, t1) that chunk c owns: the host splits the segment's tiles into S
// chunks of seg_tiles_base tiles, the first seg_tiles_rem chunks taking one more.
__device__ __forceinline__ void chunk_tiles(const StepArgs &p, uint32_t seg, uint32_t c, uint32_t &t0,
                                            uint32_t &t1)
{
    const uint32_t base = p.seg_tiles_base[seg], rem = p.seg_tiles_rem[seg];
    t0 = c * base + min(c, rem);
    t1 = t0 + base + (c < rem ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------
// flow mode (sharded, gather algorithm 3): device-side hand-offs inside the force launch

// physical block row -> logical chunk row: rows are rotated so that the rows of this rank's OWN slice
// are dispatched first (they need no exchange); the logical row still names the chunk and the
// partial-sum slot, so the summation order -- and every bit of the result -- is that of the
// unrotated launch.
__device__ __forceinline__ uint32_t flow_row(const StepArgs &p, uint32_t by)
{
    if (!p.flow_arrived) return by;
    const uint32_t r = by + p.flow_row_rot;
    return r >= gridDim.y ? r - gridDim.y : r;
}

// ONE lane of the workgroup waits until the slices of every peer that owns part of the workgroup's
// j-range (its WAVES consecutive chunks) have been pulled into the local replica by
// flow_pull_kernel (exchange number >= flow_need); the other waves park at the barrier.  The flags
// live in ordinary device memory and are polled with L1-bypassing loads (sc1: served by the L2) and
// s_sleep, so a few hundred parked workgroups cost the computing ones no memory bandwidth.  Then this
// CU's vector L1 and scalar cache drop whatever they may hold of those addresses.  Workgroups whose
// chunks lie in the rank's own slice do not wait at all.  Bounded by a wall-clock timeout that sets
// *flow_status (the host reports it) instead of hanging.
template <int WAVES>
__device__ __forceinline__ void flow_wait_block(const StepArgs &p, uint32_t seg, uint32_t by)
{
    uint32_t t0, t1, tl0, tl1;
    chunk_tiles(p, seg, by * WAVES, t0, t1);
    chunk_tiles(p, seg, by * WAVES + (WAVES - 1), tl0, tl1);
    const uint32_t j_first = p.seg_first[seg], j_count = p.seg_count[seg];
    const uint32_t j0 = j_first + t0 * 64u, j1 = j_first + min(tl1 * 64u, j_count);
    if (threadIdx.x == 0 && j1 > j0) {
        const uint32_t q0 = j0 / p.flow_count, q1 = (j1 - 1u) / p.flow_count;
        for (uint32_t q = q0; q <= q1 && q < p.flow_world; q++) {
            if (q == p.flow_rank) continue;
            const uint64_t ts = __builtin_amdgcn_s_memrealtime();
            uint32_t *dead = p.flow_peer_flags[p.flow_rank] + SYM_DEAD_WORD;   // (this rank's: set once a wait has given up -- nothing waits again)
            while ((int32_t)(__hip_atomic_load(p.flow_arrived + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.flow_need) < 0) {
                if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
                __builtin_amdgcn_s_sleep(32);
                if (__builtin_amdgcn_s_memrealtime() - ts > p.flow_timeout_ticks) {
                    __hip_atomic_store(p.flow_status, 1u + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        // the slices were stored write-through by another workgroup DURING this launch: nothing of them
        // may be served from this CU's vector L1 (nobody reads a remote slice before its flag, so the
        // caches were clean of it at launch; the invalidates make that independent of prefetching)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
}

// after a tile's last arriver has stored its bodies' new positions (system-scope write-through):
// count the tile; the LAST tile of the launch publishes the exchange number to every peer's flag
// array over xGMI.  Every tile's stores were acknowledged before its count (vmcnt(0) + barrier), so
// the publisher's flag stores are ordered behind ALL of the slice.
__device__ __forceinline__ void flow_tile_done(const StepArgs &p)
{
    if (!p.flow_arrived) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(p.flow_tiles_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == gridDim.x) {
            __hip_atomic_store(p.flow_tiles_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t q = 0; q < p.flow_world; q++)
                if (q != p.flow_rank)
                    __hip_atomic_store(p.flow_peer_flags[q] + p.flow_rank, p.flow_publish, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int K2>
struct Bodies {
    v2f xi[K2], yi[K2], zi[K2];
    Acc2 acc[K2];
};

// lane's 2*K2 bodies: element e of pair k is local body (bx*2*K2 + 2*k + e)*64 + lane
template <int K2>
__device__ __forceinline__ void load_bodies(Bodies<K2> &b, const StepArgs &p, uint32_t bx, uint32_t lane)
{
#pragma unroll
    for (int k = 0; k < K2; k++) {
        uint32_t l0 = (bx * (2 * K2) + 2 * k) * 64u + lane, l1 = l0 + 64u;
        l0 = l0 < p.i_count ? l0 : (p.i_count - 1u);      // clamp: tail lanes redo a valid body
        l1 = l1 < p.i_count ? l1 : (p.i_count - 1u);
        const float4 b0 = p.pos_old[p.i_first + l0], b1 = p.pos_old[p.i_first + l1];
        b.xi[k] = v2f{b0.x, b1.x}; b.yi[k] = v2f{b0.y, b1.y}; b.zi[k] = v2f{b0.z, b1.z};
        b.acc[k] = Acc2{v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    }
}

// 16-byte write-through stores / L1-bypassing loads (global_store/load_dwordx4 sc1) for data handed
// from one workgroup to another INSIDE a launch (MI355X_MICROARCH.md, inter-workgroup visibility):
// the bytes leave the writer's L2 and are never served from the reader's L1.  Inline asm because HIP
// has no 16-byte agent-scope access; one dwordx4 per row entry writes whole 64-B segments per wave --
// split into two 8-byte stores the same rows cost twice the memory-side write traffic
// (WRITE_SIZE 18.3 MiB instead of 10 MiB per 65 536-body launch, profiles/r02_pmc_summary.txt).
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_row_sc1(float4 *dst, float ax, float ay, float az)
{
    const f4v v = {ax, ay, az, 0.f};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
}

// eight rows in flight, then ONE wait: the compiler does not count inline-asm loads in vmcnt
__device__ __forceinline__ void load_rows8_sc1(const float4 *in, size_t stride, f4v (&r)[8])
{
    const float4 *p0 = in, *p1 = in + stride, *p2 = in + 2 * stride, *p3 = in + 3 * stride;
    const float4 *p4 = in + 4 * stride, *p5 = in + 5 * stride, *p6 = in + 6 * stride, *p7 = in + 7 * stride;
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
                 : "memory");
}

__device__ __forceinline__ f4v load_row_sc1(const float4 *src)
{
    f4v r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(src) : "memory");
    return r;
}

// epilogue shared by both force kernels: the WAVES chunk sums of a workgroup are combined in LDS
// in ascending wave order (fixed order, no atomics); then, by EPI,
//   EPI_FUSED  the workgroup has seen every chunk of its bodies and integrates them in place;
//   EPI_ROWS   it stores ONE partial sum per body into row seg_slot[seg] + by of the scratch
//              buffer for reduce_integrate_kernel (second launch);
//   EPI_TICKET it publishes that row write-through, takes a ticket of its i-tile, and the workgroup
//              whose ticket is the last one sums ALL rows in ascending row order and integrates --
//              the same additions in the same order as EPI_ROWS + reduce_integrate_kernel, in one
//              launch and without float atomics.  Hand-off form: sc1 stores -> every storing wave's
//              s_waitcnt vmcnt(0) -> workgroup barrier -> one agent-scope atomic add; last arriver:
//              agent acquire -> barrier -> sc1 loads.  The last arriver re-arms the ticket (0).
template <int K2, int WAVES, int EPI>
__device__ __forceinline__ void finish(const Bodies<K2> &b, const StepArgs &p, uint32_t bx, uint32_t by,
                                       uint32_t w, uint32_t lane, uint32_t seg,
                                       float (*red)[3][128 * K2], uint32_t *last_flag)
{
#pragma unroll
    for (int k = 0; k < K2; k++) {
        red[w][0][(2 * k) * 64 + lane] = b.acc[k].x.x; red[w][0][(2 * k + 1) * 64 + lane] = b.acc[k].x.y;
        red[w][1][(2 * k) * 64 + lane] = b.acc[k].y.x; red[w][1][(2 * k + 1) * 64 + lane] = b.acc[k].y.y;
        red[w][2][(2 * k) * 64 + lane] = b.acc[k].z.x; red[w][2][(2 * k + 1) * 64 + lane] = b.acc[k].z.y;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < 128u * K2; e += 64u * WAVES) {
        const uint32_t li = bx * (128u * K2) + e;
        if (li < p.i_count) {
            float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ww++) { ax += red[ww][0][e]; ay += red[ww][1][e]; az += red[ww][2][e]; }
            if constexpr (EPI == EPI_FUSED) {
                const uint32_t i = p.i_first + li;
                integrate_store(p, i, p.pos_old[i], ax * p.mass, ay * p.mass, az * p.mass);
            } else if constexpr (EPI == EPI_ROWS) {
                p.partial[(size_t)(p.seg_slot[seg] + by) * p.partial_stride + li] = make_float4(ax, ay, az, 0.f);
            } else {
                store_row_sc1(p.partial + (size_t)(p.seg_slot[seg] + by) * p.partial_stride + li, ax, ay, az);
            }
        }
    }
    if constexpr (EPI == EPI_FUSED) flow_tile_done(p);
    if constexpr (EPI == EPI_TICKET) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's row stores have left the chip's caches
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t prev = __hip_atomic_fetch_add(p.ticket + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t last = (prev + 1u == p.ticket_total) ? 1u : 0u;
            if (last) {
                __hip_atomic_store(p.ticket + bx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *last_flag = last;
        }
        __syncthreads();
        if (*last_flag == 0u) return;
        const uint32_t rows = p.ticket_total;
        for (uint32_t e = threadIdx.x; e < 128u * K2; e += 64u * WAVES) {
            const uint32_t li = bx * (128u * K2) + e;
            if (li >= p.i_count) continue;
            const float4 *in = p.partial + li;
            const size_t stride = p.partial_stride;
            float ax = 0.f, ay = 0.f, az = 0.f;
            uint32_t s = 0;
            for (; s + 8u <= rows; s += 8u) {                  // 8 rows in flight, summed in ascending order
                f4v r[8];
                load_rows8_sc1(in + (size_t)s * stride, stride, r);
#pragma unroll
                for (int u = 0; u < 8; u++) { ax += r[u].x; ay += r[u].y; az += r[u].z; }
            }
            for (; s < rows; s++) {
                const f4v r = load_row_sc1(in + (size_t)s * stride);
                ax += r.x; ay += r.y; az += r.z;
            }
            const uint32_t i = p.i_first + li;
            integrate_store(p, i, p.pos_old[i], ax * p.mass, ay * p.mass, az * p.mass);
        }
        flow_tile_done(p);
    }
}

// ---------------------------------------------------------------------------------------------
// all-pairs force, LDS-tiled
//
// grid  = (ceil(i_count / (128*K2)), SB, nseg)      block = 64 * WAVES
// wave w of block (bx, by, z) accumulates its lanes' 2*K2 bodies against chunk c = by*WAVES + w
// of segment z.  A tile of 64 j-bodies is fetched with one coalesced global_load_dwordx4 per
// lane and written to a WAVE-PRIVATE LDS slot (no s_barrier in the loop), split as
//   xy[32] = (x0,y0,x1,y1) per two bodies,  zz[16] = (z0,z1,z2,z3) per four bodies,
// so that four j-bodies cost three broadcast ds_read_b128 (3 LDS cycles per j per wave instead
// of 8 for the ds_read_b96 the float4 layout compiles to) and x_j / y_j / z_j always sit in
// the low or high half of an aligned register pair, which v_pk_*_f32 selects with op_sel for
// free.  The next tile's global load is in flight while the current tile is consumed.
template <int K2, int WAVES, int EPI>
__global__ __launch_bounds__(64 * WAVES) void force_lds_kernel(const StepArgs p)
{
    __shared__ float4 tile_xy[WAVES][2][32];
    __shared__ float4 tile_zz[WAVES][2][16];
    __shared__ float red[WAVES][3][128 * K2];
    __shared__ uint32_t last_flag;

    uint32_t bx, by;
    xcd_remap(bx, by, p.xcd_remap);
    by = flow_row(p, by);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t seg = blockIdx.z;
    const uint32_t c = by * WAVES + w;
    const uint32_t j_first = p.seg_first[seg], j_count = p.seg_count[seg];
    const float4 *__restrict__ pos = p.pos_old;

    Bodies<K2> b;
    load_bodies<K2>(b, p, bx, lane);
    if (p.flow_arrived) flow_wait_block<WAVES>(p, seg, by);

    uint32_t t0, t1;
    chunk_tiles(p, seg, c, t0, t1);
    const v2f soft2 = v2f{p.soft2, p.soft2};

    if (t0 < t1) {
        float4 cur;
        {
            uint32_t j = t0 * 64u + lane;
            j = j < j_count ? j : (j_count - 1u);
            cur = pos[j_first + j];
        }
        for (uint32_t t = t0; t < t1; t++) {
            const uint32_t buf = (t - t0) & 1u;
            float2 *wxy = reinterpret_cast<float2 *>(tile_xy[w][buf]);
            float *wz = reinterpret_cast<float *>(tile_zz[w][buf]);
            wxy[lane] = make_float2(cur.x, cur.y);             // ds_write_b64
            wz[lane] = cur.z;                                  // ds_write_b32
            // prefetch the next tile into registers while this one is consumed
            if (t + 1u < t1) {
                uint32_t j = (t + 1u) * 64u + lane;
                j = j < j_count ? j : (j_count - 1u);
                cur = pos[j_first + j];
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t nj = min(64u, j_count - t * 64u);   // wave-uniform; < 64 only on a segment's last tile
            const float4 *txy = tile_xy[w][buf];
            const float4 *tzz = tile_zz[w][buf];
            if (nj == 64u) {
#pragma unroll 4
                for (int q = 0; q < 16; q++) {
                    const float4 zz = tzz[q];                  // ds_read_b128, every lane the same address
                    const float4 xa = txy[2 * q], xb = txy[2 * q + 1];
                    const v2f z01 = v2f{zz.x, zz.y}, z23 = v2f{zz.z, zz.w};
                    const v2f j0 = v2f{xa.x, xa.y}, j1 = v2f{xa.z, xa.w}, j2 = v2f{xb.x, xb.y}, j3 = v2f{xb.z, xb.w};
#pragma unroll
                    for (int k = 0; k < K2; k++) pair_term2_pairs<0>(b.acc[k], b.xi[k], b.yi[k], b.zi[k], j0, z01, soft2);
#pragma unroll
                    for (int k = 0; k < K2; k++) pair_term2_pairs<1>(b.acc[k], b.xi[k], b.yi[k], b.zi[k], j1, z01, soft2);
#pragma unroll
                    for (int k = 0; k < K2; k++) pair_term2_pairs<0>(b.acc[k], b.xi[k], b.yi[k], b.zi[k], j2, z23, soft2);
#pragma unroll
                    for (int k = 0; k < K2; k++) pair_term2_pairs<1>(b.acc[k], b.xi[k], b.yi[k], b.zi[k], j3, z23, soft2);
                }
            } else {
                const float2 *sxy = reinterpret_cast<const float2 *>(txy);
                const float *sz = reinterpret_cast<const float *>(tzz);
                for (uint32_t jj = 0; jj < nj; jj++) {
                    const float2 xy = sxy[jj];
                    const float z = sz[jj];
#pragma unroll
                    for (int k = 0; k < K2; k++) pair_term2(b.acc[k], b.xi[k], b.yi[k], b.zi[k], xy.x, xy.y, z, soft2);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    finish<K2, WAVES, EPI>(b, p, bx, by, w, lane, seg, red, &last_flag);
}

// ---------------------------------------------------------------------------------------------
// all-pairs force, j-bodies through the scalar cache (no LDS, no VGPRs for j)
//
// Same decomposition as force_lds_kernel.  pos[j] with a wave-uniform j compiles to
// s_load_dwordx8/x16 and the packed VALU ops take (x_j,y_j) / (z_j,w_j) as their one SGPR-pair
// source, op_sel picking the half.  hipcc selects scalar loads only for memory it can prove no
// instruction of the kernel clobbers before the load: the j-source is therefore its own
// `const __restrict__` kernel parameter (= p.pos_old; nothing in this kernel writes it), the stamps
// are read with inline asm, and the scalar-cache invalidate of flow mode is the builtin -- with
// p.pos_old, __builtin_amdgcn_s_memtime() or an asm "memory" clobber in front of the loop every j-load
// silently becomes a uniform global_load_dwordx3 and the kernel runs 27 % slower (round 2, measured:
// 1.124 vs 0.888 ms).  tests/test_abi.py checks the disassembly for the scalar loads.  The default path: measured 62 % of the fp32 peak at 65 536
// bodies against 60 % for the LDS-tiled kernel, with SQ_INSTS_VALU exactly 13 per
// two pairs.
template <int K2, int WAVES, int EPI>
__global__ __launch_bounds__(64 * WAVES) void force_sgpr_kernel(const StepArgs p, const float4 *__restrict__ pos_j)
{
    __shared__ float red[WAVES][3][128 * K2];
    __shared__ uint32_t last_flag;

    uint32_t bx, by;
    xcd_remap(bx, by, p.xcd_remap);
    by = flow_row(p, by);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t seg = blockIdx.z;
    const uint32_t c = by * WAVES + w;
    const uint32_t j_first = p.seg_first[seg], j_count = p.seg_count[seg];

    Bodies<K2> b;
    load_bodies<K2>(b, p, bx, lane);
    if (p.flow_arrived) flow_wait_block<WAVES>(p, seg, by);

    uint32_t t0, t1;
    chunk_tiles(p, seg, c, t0, t1);
    const v2f soft2 = v2f{p.soft2, p.soft2};
    const uint32_t j1 = min(t1 * 64u, j_count);
    const float4 *__restrict__ pj = pos_j + j_first;

    // diagnostic launches only (mapn_measure_clock): shader-clock and 100 MHz wall-clock stamps around
    // the pair loop, written to a buffer nothing else reads; p.stamps is null in every normal launch
    unsigned long long st_c = 0, st_r = 0;
    if (p.stamps) asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(st_r), "=s"(st_c));

    uint32_t j = t0 * 64u;
    for (; j + 8u <= j1; j += 8u) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float4 bj = pj[j + u];                       // uniform address -> s_load
#pragma unroll
            for (int k = 0; k < K2; k++) pair_term2(b.acc[k], b.xi[k], b.yi[k], b.zi[k], bj.x, bj.y, bj.z, soft2);
        }
    }
    for (; j < j1; j++) {
        const float4 bj = pj[j];
#pragma unroll
        for (int k = 0; k < K2; k++) pair_term2(b.acc[k], b.xi[k], b.yi[k], b.zi[k], bj.x, bj.y, bj.z, soft2);
    }
    if (p.stamps) {
        // the accumulators must be complete before the closing stamp: tie it to them
        asm volatile("" :: "v"(b.acc[0].x), "v"(b.acc[0].y), "v"(b.acc[0].z));
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            const size_t wave = ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * WAVES + w;
            p.stamps[2 * wave] = c1 - st_c;
            p.stamps[2 * wave + 1] = r1 - st_r;
        }
    }
    finish<K2, WAVES, EPI>(b, p, bx, by, w, lane, seg, red, &last_flag);
}

// ---------------------------------------------------------------------------------------------
// combine the partial sums of a body in ascending slot order, then integrate (second kernel of
// the non-fused path).  HBM-bound: (16*slots + 28) B read + 28 B written per body.
__global__ __launch_bounds__(256) void reduce_integrate_kernel(const StepArgs p, uint32_t slots)
{
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= p.i_count) return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    const float4 *in = p.partial + li;
    const size_t stride = p.partial_stride;
    uint32_t s = 0;
    for (; s + 8u <= slots; s += 8u) {                     // 8 independent loads in flight, summed in order
        float4 a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = in[(size_t)(s + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) { ax += a[u].x; ay += a[u].y; az += a[u].z; }
    }
    for (; s < slots; s++) {
        const float4 a = in[(size_t)s * stride];
        ax += a.x; ay += a.y; az += a.z;
    }
    const uint32_t i = p.i_first + li;
    integrate_store(p, i, p.pos_old[i], ax * p.mass, ay * p.mass, az * p.mass);
}

// ---------------------------------------------------------------------------------------------
// CSMain exactly as shipped (nBodyGravityCS.hlsl:86-109): one gravity well at the origin.
// HBM-bound, 56 B per body.  One thread per body like the reference.
// STREAM (chosen by the host where a step's 56 B per body exceed 320 MiB, i.e. from 6 Mi bodies on): non-temporal loads and stores.
// Measured (round 4, a probe of this kernel: profiles/r04_central_well_stream.txt): inside the Infinity Cache -- the reference's default
// 4 Mi bodies, 235 MB per step -- the plain form runs at 6.9 TB/s and the hint costs 12 %; from 6 Mi bodies on the plain form falls to
// 5.6 - 6.1 TB/s (with the placement of the buffers: +- 4 %) and the hint gives 6.1 - 6.45 at every size and placement tried
// (the guide's float4-copy rate: 6.29).  More bodies per thread change nothing.
template <bool STREAM>
__global__ __launch_bounds__(256) void central_well_kernel(const StepArgs p)
{
    const uint32_t li = blockIdx.x * 256u + threadIdx.x;
    if (li >= p.i_count) return;
    const uint32_t i = p.i_first + li;
    float4 pos;
    if constexpr (STREAM) {
        typedef float f4s __attribute__((ext_vector_type(4)));
        const f4s v = __builtin_nontemporal_load(reinterpret_cast<const f4s *>(p.pos_old + i));
        pos = make_float4(v.x, v.y, v.z, v.w);
    } else {
        pos = p.pos_old[i];
    }
    float d = __builtin_fmaf(pos.x, pos.x, p.soft2);     // :94-95
    d = __builtin_fmaf(pos.y, pos.y, d);
    d = __builtin_fmaf(pos.z, pos.z, d);
    const float inv = -__builtin_amdgcn_rsqf(d);         // :97
    const float s = p.mass * (inv * inv * inv);          // :98-99
    integrate_store<STREAM>(p, i, pos, pos.x * s, pos.y * s, pos.z * s);
}

// ---------------------------------------------------------------------------------------------
// Direct exchange of the new position slices between the GPUs of a node, without a collective
// library: workgroup b serves peer q (the b-th rank other than this one).
//   1. publish : one lane stores `step` into q's flag array, slot [rank], over xGMI.  The slice
//                itself was written by the PREVIOUS kernel of this stream (reduce_integrate), so
//                it is in this GPU's memory before this kernel starts; nothing to flush here.
//   2. wait    : the same lane polls this GPU's own flag array, slot [q] (uncached memory, relaxed
//                system-scope loads, s_sleep between polls, bounded by a wall-clock timeout that
//                sets *status instead of hanging), then ONE system-scope acquire so that this
//                CU's L1 and this XCD's L2 hold no stale copy of q's buffer from two steps ago.
//   3. pull    : the workgroup copies q's slice (count float4) from q's buffer into the same
//                offset of the local buffer.
// No end-of-step handshake is needed: q overwrites its slice of this buffer again only at step
// s+2, after its force(s+2), which needs this rank's slice(s+1), which this rank produces only
// after its own gather(s) -- i.e. after this pull -- has completed.
__global__ __launch_bounds__(1024) void p2p_gather_kernel(const P2PArgs p)
{
    const uint32_t b = blockIdx.x;
    const uint32_t q = b < p.rank ? b : b + 1u;            // skip self
    __shared__ uint32_t ok;
    if (threadIdx.x == 0) {
        __hip_atomic_store(p.peer_flags[q] + p.rank, p.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        uint32_t good = 1u;
        while ((int32_t)(__hip_atomic_load(p.my_flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - p.step) < 0) {
            // (SYM_DEAD_WORD: an earlier wait of this rank has given up -- the launches queued behind it do not wait a time-out each)
            if (__hip_atomic_load(p.my_flags + SYM_DEAD_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { good = 0u; break; }
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                good = 0u;
                __hip_atomic_store(p.status, 1u + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(p.my_flags + SYM_DEAD_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       // system scope: buffer_inv sc0 sc1
        ok = good;
    }
    __syncthreads();
    if (!ok) return;
    // payload as 8-byte relaxed SYSTEM-scope loads (global_load_dwordx2 sc0 sc1): they bypass this
    // GPU's L1/L2 altogether, so a line of q's buffer cached here two steps ago cannot be returned
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(p.peer[q] + (size_t)q * p.count);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.local + (size_t)q * p.count);
    const uint32_t words = p.count * 2u;
    uint32_t i = threadIdx.x;
    for (; i + 7u * 1024u < words; i += 8u * 1024u) {          // 8 remote loads in flight per lane
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __hip_atomic_load(src + i + u * 1024u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int u = 0; u < 8; u++) dst[i + u * 1024u] = v[u];
    }
    for (; i < words; i += 1024u)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// flow mode, the pull half on the comm stream: workgroup b serves peer q.  One lane waits for q's
// flag (q's LAST integrated tile stored it, behind all of q's slice), the workgroup pulls the slice
// with cache-bypassing system-scope loads and stores it write-through (agent scope), drains, and one
// lane marks arrived[q] = step -- what the force launch's remote-chunk waves wait for.
__global__ __launch_bounds__(512) void flow_pull_kernel(const P2PArgs p, uint32_t *arrived)
{
    const uint32_t b = blockIdx.x;
    const uint32_t q = b < p.rank ? b : b + 1u;            // skip self
    __shared__ uint32_t ok;
    if (threadIdx.x == 0) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        uint32_t good = 1u;
        while ((int32_t)(__hip_atomic_load(p.my_flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - p.step) < 0) {
            // (SYM_DEAD_WORD: an earlier wait of this rank has given up -- the launches queued behind it do not wait a time-out each)
            if (__hip_atomic_load(p.my_flags + SYM_DEAD_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { good = 0u; break; }
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > p.timeout_ticks) {
                good = 0u;
                __hip_atomic_store(p.status, 1u + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(p.my_flags + SYM_DEAD_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        ok = good;
    }
    __syncthreads();
    if (ok) {
        // 16 bytes per lane per access: system-scope loads that bypass this GPU's caches (a line of q's
        // buffer cached two steps ago cannot be returned), agent-scope write-through stores
        const float4 *src = p.peer[q] + (size_t)q * p.count;
        float4 *dst = p.local + (size_t)q * p.count;
        uint32_t i = threadIdx.x;
        for (; i + 7u * 512u < p.count; i += 8u * 512u) {          // 8 remote loads (64 KiB per workgroup) in flight
            f4v v[8];
            asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
                         "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
                         "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                         "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                         : "v"(src + i), "v"(src + i + 512u), "v"(src + i + 1024u), "v"(src + i + 1536u),
                           "v"(src + i + 2048u), "v"(src + i + 2560u), "v"(src + i + 3072u), "v"(src + i + 3584u)
                         : "memory");
#pragma unroll
            for (int u = 0; u < 8; u++)
                asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + i + u * 512u), "v"(v[u]) : "memory");
        }
        for (; i < p.count; i += 512u) {
            f4v v;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src + i) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + i), "v"(v) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave: its write-through stores are acknowledged
    __syncthreads();
    // also after a timed-out wait: the force launch must not hang on top of it (the status word reports it)
    if (threadIdx.x == 0) __hip_atomic_store(arrived + q, p.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// flow mode, the publish half alone: a rank whose slice is not advanced by a force launch in this step
// (no active body in it, or the central-well step) still owes its peers the flag.  Stream-ordered
// behind whatever wrote the slice (the kernel boundary made it visible).
__global__ __launch_bounds__(64) void flow_publish_kernel(uint32_t *const *peer_flags, uint32_t rank, uint32_t world, uint32_t step)
{
    if (threadIdx.x != 0) return;
    for (uint32_t q = 0; q < world; q++)
        if (q != rank) __hip_atomic_store(peer_flags[q] + rank, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_flow_pull(const P2PArgs &a, uint32_t *arrived, hipStream_t st)
{
    if (a.world < 2) return hipSuccess;
    hipLaunchKernelGGL(flow_pull_kernel, dim3(a.world - 1), dim3(512), 0, st, a, arrived);
    return hipGetLastError();
}

hipError_t launch_flow_publish(uint32_t *const *peer_flags, uint32_t rank, uint32_t world, uint32_t step, hipStream_t st)
{
    hipLaunchKernelGGL(flow_publish_kernel, dim3(1), dim3(64), 0, st, peer_flags, rank, world, step);
    return hipGetLastError();
}

hipError_t launch_p2p_gather(const P2PArgs &a, hipStream_t st)
{
    if (a.world < 2) return hipSuccess;
    hipLaunchKernelGGL(p2p_gather_kernel, dim3(a.world - 1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The consumer's fence as memory words (the render adapter's shared fence, Compute.cpp:1012
// `m_commandQueue->Wait(m_sharedRenderFence, in_sharedFenceValue - 1)`): a GPU-side wait that can be
// queued BEFORE the consumer has signalled -- the compute stream parks in this one-lane kernel
// until either word reaches `need`.  host_word: pinned host memory written by
// mapn_consumer_signal(); dev_word: uncached device memory written by fence_signal_kernel (an
// event-ordered signal, possibly from another process through hipIpc).  Bounded: after
// timeout_ticks (100 MHz) it sets *status = 1 and lets the stream continue.
__global__ __launch_bounds__(64) void fence_wait_kernel(const uint32_t *host_word, const uint32_t *dev_word,
                                                        uint32_t need, uint64_t timeout_ticks, uint32_t *status)
{
    if (threadIdx.x != 0) return;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const uint32_t h = __hip_atomic_load(host_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t d = __hip_atomic_load(dev_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(h - need) >= 0 || (int32_t)(d - need) >= 0) break;
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// Signal(fence, value) as a stream operation: dev_word = max(dev_word, value), system scope.
__global__ __launch_bounds__(64) void fence_signal_kernel(uint32_t *dev_word, uint32_t value)
{
    if (threadIdx.x != 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_fetch_max(dev_word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Status block of an exported context (mapn_ipc_export): after a step, on the compute stream,
// block[17] = index of the buffer holding the latest positions, then block[16] = the fence value
// that step signalled -- what a consumer in ANOTHER process reads to learn where the results are.
__global__ __launch_bounds__(64) void status_publish_kernel(uint32_t *block, uint32_t fence_value, uint32_t latest_index)
{
    if (threadIdx.x != 0) return;
    __hip_atomic_store(block + 17, latest_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(block + 16, fence_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_status_publish(uint32_t *block, uint32_t fence_value, uint32_t latest_index, hipStream_t st)
{
    hipLaunchKernelGGL(status_publish_kernel, dim3(1), dim3(64), 0, st, block, fence_value, latest_index);
    return hipGetLastError();
}

hipError_t launch_fence_wait(const uint32_t *host_word, const uint32_t *dev_word, uint32_t need, uint64_t timeout_ticks,
                             uint32_t *status, hipStream_t st)
{
    hipLaunchKernelGGL(fence_wait_kernel, dim3(1), dim3(64), 0, st, host_word, dev_word, need, timeout_ticks, status);
    return hipGetLastError();
}

hipError_t launch_fence_signal(uint32_t *dev_word, uint32_t value, hipStream_t st)
{
    hipLaunchKernelGGL(fence_signal_kernel, dim3(1), dim3(64), 0, st, dev_word, value);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host-side launch table

template <int K2, int WAVES>
static hipError_t launch_force_variant(int kind, int epi, dim3 grid, const StepArgs &a, hipStream_t st)
{
    const dim3 block(64 * WAVES);
    if (kind == KERNEL_LDS) {
        if (epi == EPI_FUSED)       hipLaunchKernelGGL((force_lds_kernel<K2, WAVES, EPI_FUSED>), grid, block, 0, st, a);
        else if (epi == EPI_TICKET) hipLaunchKernelGGL((force_lds_kernel<K2, WAVES, EPI_TICKET>), grid, block, 0, st, a);
        else                        hipLaunchKernelGGL((force_lds_kernel<K2, WAVES, EPI_ROWS>), grid, block, 0, st, a);
    } else {
        if (epi == EPI_FUSED)       hipLaunchKernelGGL((force_sgpr_kernel<K2, WAVES, EPI_FUSED>), grid, block, 0, st, a, a.pos_old);
        else if (epi == EPI_TICKET) hipLaunchKernelGGL((force_sgpr_kernel<K2, WAVES, EPI_TICKET>), grid, block, 0, st, a, a.pos_old);
        else                        hipLaunchKernelGGL((force_sgpr_kernel<K2, WAVES, EPI_ROWS>), grid, block, 0, st, a, a.pos_old);
    }
    return hipGetLastError();
}

bool force_plan_supported(const ForcePlan &plan)
{
    if (plan.kind != KERNEL_LDS && plan.kind != KERNEL_SGPR) return false;
    if (plan.k != 2 && plan.k != 4 && plan.k != 8) return false;
    if (plan.waves != 1 && plan.waves != 2 && plan.waves != 4 && plan.waves != 8 && plan.waves != 16) return false;
    if (plan.k == 8 && plan.waves == 16) return false;
    if (plan.epi != EPI_FUSED && plan.epi != EPI_ROWS && plan.epi != EPI_TICKET) return false;
    if (plan.epi == EPI_FUSED && (plan.sb != 1 || plan.nseg != 1)) return false;
    return plan.sb >= 1 && plan.nseg >= 1 && plan.nseg <= MAX_SEGMENTS;
}

hipError_t launch_force(const ForcePlan &plan, const StepArgs &a, hipStream_t st)
{
    if (!force_plan_supported(plan)) return hipErrorInvalidConfiguration;
    const uint32_t per_block = 64u * plan.k;
    const dim3 grid((a.i_count + per_block - 1) / per_block, plan.sb, plan.nseg);
#define MAPN_CASE(KK, WW) \
    if (plan.k == 2 * KK && plan.waves == WW) return launch_force_variant<KK, WW>(plan.kind, plan.epi, grid, a, st);
    MAPN_CASE(1, 1) MAPN_CASE(1, 2) MAPN_CASE(1, 4) MAPN_CASE(1, 8) MAPN_CASE(1, 16)
    MAPN_CASE(2, 1) MAPN_CASE(2, 2) MAPN_CASE(2, 4) MAPN_CASE(2, 8) MAPN_CASE(2, 16)
    MAPN_CASE(4, 1) MAPN_CASE(4, 2) MAPN_CASE(4, 4) MAPN_CASE(4, 8)
#undef MAPN_CASE
    return hipErrorInvalidConfiguration;
}

hipError_t launch_reduce_integrate(const StepArgs &a, uint32_t slots, hipStream_t st)
{
    // a latency-bound pass: small slices (a shard of a multi-GPU job) use 64-thread workgroups so
    // that 8192 bodies still spread over 128 CUs instead of 32
    const uint32_t block = a.i_count <= 32768u ? 64u : 256u;
    hipLaunchKernelGGL(reduce_integrate_kernel, dim3((a.i_count + block - 1u) / block), dim3(block), 0, st, a, slots);
    return hipGetLastError();
}

hipError_t launch_central_well(const StepArgs &a, hipStream_t st)
{
    // (the new positions of flow mode go write-through at system scope: never the streaming form there)
    const bool stream = !a.flow_arrived && (uint64_t)a.i_count * 56u > (320ull << 20);
    if (stream) hipLaunchKernelGGL(central_well_kernel<true>, dim3((a.i_count + 255u) / 256u), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(central_well_kernel<false>, dim3((a.i_count + 255u) / 256u), dim3(256), 0, st, a);
    return hipGetLastError();
}

const char *force_kernel_name(const ForcePlan &plan)
{
    return plan.kind == KERNEL_LDS ? "force_lds_kernel" : "force_sgpr_kernel";
}

}  // namespace mapn
