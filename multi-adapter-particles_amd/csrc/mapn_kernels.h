// mapn_kernels.h -- launch interface between the host context (mapn_context.cpp) and the
// gfx950 kernels (mapn_kernels.hip).  Internal; the public boundary is include/mapn.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mapn_sym_plan.h"   // sym_runs_half: which block of a half-ring pair runs its meetings

namespace mapn {

enum { KERNEL_LDS = 1, KERNEL_SGPR = 2, KERNEL_SYM = 3 };
enum { MAX_SEGMENTS = 3 };
// where the kick-drift integrator runs (see finish<> in mapn_kernels.hip)
enum { EPI_ROWS = 0, EPI_FUSED = 1, EPI_TICKET = 2 };

// One launch's view of the state.  Passed by value (kernarg segment, read through s_load).
struct StepArgs {
    const float4 *pos_old;    // float4[N]  read buffer  (oldPosition, hlsl:77 u1)
    const float  *vel_old;    // float3[N]  packed, 12 B stride (oldVelocity, u4)
    float4       *pos_new;    // write buffer (newPosition, u0)
    float        *vel_new;    // (newVelocity, u3)
    float4       *partial;    // [slots][partial_stride] chunk sums of the non-fused path
    uint32_t      partial_stride;
    uint32_t     *ticket;     // EPI_TICKET: one arrival counter per i-tile, zero between launches
    uint32_t      ticket_total;             // arrivals that complete an i-tile = partial rows of the step
    uint32_t      i_first;    // bodies [i_first, i_first + i_count) advance in this launch
    uint32_t      i_count;
    uint32_t      seg_first[MAX_SEGMENTS];  // j-segments of pos_old this launch sums over
    uint32_t      seg_count[MAX_SEGMENTS];
    uint32_t      seg_slot[MAX_SEGMENTS];   // first partial slot of each segment
    uint32_t      seg_tiles_base[MAX_SEGMENTS];  // 64-body tiles per chunk ...
    uint32_t      seg_tiles_rem[MAX_SEGMENTS];   // ... the first `rem` chunks take one more
    float         mass, soft2, dt, damping; // hlsl:37-38, Compute.cpp:545-546
    uint32_t      xcd_remap;                // 1: XCD-aware block remap (default), 0: plain blockIdx (A/B)
    // sharded "flow" mode (gather algorithm 3): the exchange runs BESIDE this launch (flow_pull_kernel on
    // the comm stream); waves of a remote j-chunk wait for that peer's slice, the last tile publishes.
    // flow_arrived == null: not in flow mode.
    const uint32_t *flow_arrived;           // [world] local: exchange number of the latest slice pulled from each peer
    uint32_t     *flow_tiles_done;          // local counter of integrated i-tiles, zero between launches
    uint32_t *const *flow_peer_flags;       // device table [world]: every rank's flag array as mapped here
    uint32_t     *flow_status;              // host-visible words: [0] = 1 + peer whose slice never arrived
    uint64_t      flow_timeout_ticks;
    uint32_t      flow_need;                // remote chunks need flow_arrived[q] >= flow_need (the previous exchange)
    uint32_t      flow_publish;             // exchange number this launch publishes to the peers
    uint32_t      flow_rank, flow_world, flow_count;   // flow_count = bodies per rank
    uint32_t      flow_row_rot;             // physical block row y handles logical row (y + rot) % rows: own slice first
    unsigned long long *stamps;             // diagnostic launches only (mapn_measure_clock): per wave {d s_memtime, d s_memrealtime}; else null
};

// How the j-range is cut: S = sb * waves chunks per segment.
struct ForcePlan {
    int      kind;     // KERNEL_*
    uint32_t k;        // bodies per lane (2, 4 or 8: packed fp32 handles them in pairs)
    uint32_t waves;    // waves per workgroup (j-split inside the workgroup)
    uint32_t sb;       // j-split across workgroups (gridDim.y)
    uint32_t nseg;     // gridDim.z
    int      epi;      // EPI_*: EPI_FUSED requires sb == 1 and nseg == 1
};

// The symmetric (Newton's third law) all-pairs step, mapn_sym.hip.  The launch plan (which steps every wave runs, in how
// many launches a step is made) is built on the host: mapn_sym_plan.h.
enum { SYM_K2 = 8,                     // packed pairs of bodies i per lane
       SYM_BLOCK = 128 * SYM_K2,       // bodies per I-block (one wave)
       SYM_JPI = SYM_BLOCK / 64 };     // 64-body J-blocks per I-block
// One element of a force row: x, y, z -- 12 bytes, packed like the velocities (round 4; until then a float4 whose .w was never
// read: a quarter of the N^2/128 bytes a step writes and reads back were padding).
struct SymRow { float x, y, z; };
static_assert(sizeof(SymRow) == 12, "rows are packed");

struct SymArgs {
    const float4 *pos_old;
    const float  *vel_old;
    float4       *pos_new;
    float        *vel_new;
    SymRow       *arow;       // [blocks of the launch][parts][SYM_BLOCK]  force on the bodies of an I-block, one row per workgroup
    SymRow       *brow;       // [n / 64][brows][64] reaction on the bodies of a J-block, one row per meeting (sharded: [n / 64][shard_nbl][64])
    SymRow       *brow1;      // [blocks of the launch][parts][64] head rows: the last steps of a meeting that was cut between two workgroups
    const float4 *acc_in;     // reduce launch: forces summed by the earlier windows of this step (null: none)
    float4       *acc_out;    // reduce launch: where this window's running sum goes (null: last window -- integrate)
    const float4 *extra;      // reduce launch of a PARTIALLY ACTIVE step's first window: [extra_rows][extra_stride] partial force rows of the one-sided
    uint32_t      extra_rows, extra_stride;   // launch over the frozen bodies (force_*_kernel, EPI_ROWS), added in ascending row order before this launch's own rows (null: none)
    const uint32_t *tab;      // this window's tables: bounds[2][nwaves + 1], split[2][max_meetings] (SymPlanHost)
    uint32_t      n, nb;      // bodies, I-blocks of SYM_BLOCK (the last may be padded)
    uint32_t      n_integrate;// bodies [0, n_integrate) are advanced by the reduce launch (roundup64(num_active)); the rest only exert force
    uint32_t      parts;      // workgroups per I-block (gridDim.y)
    uint32_t      nwaves;     // parts * waves per workgroup
    uint32_t      max_meetings;
    uint32_t      sets;       // table sets: 2, or 16 = XCD-weighted parts ("spread"): workgroup (x, y) runs part y of block (x + y) mod blocks, set = class + 2 * (block mod 8)
    uint32_t      la_flip;    // 1: workgroup (x, y) runs block x ^ 1 (sharded launches without XCD weights: puts the blocks with the half-ring group on the odd dispatch slots)
    const uint32_t *wgmap;    // XCD-weighted parts, class-aware form: workgroup (x, y) runs (block << 16 | part) = wgmap[y * gridDim.x + x] (null: the mappings above)
    uint32_t      g0, g1;     // meeting groups of this launch: 0 the block itself, 1 .. D partner a + g, D + 1 the half-ring partner
    uint32_t      brows;      // rows allocated per J-block (symmetric groups of the widest window)
    uint32_t      half_d;     // nb / 2 when nb is even (the half-ring partner), else 0
    uint32_t      a0;         // sharded form: first I-block of this rank (gridDim.x = its shard_nbl blocks); 0 otherwise
    uint32_t      shard_nbl;  // sharded form: I-blocks of this rank -- brow is one row per (J-block, local I-block); 0 = unsharded
    // sharded form, positions PUSHED by the peers (gather algorithm 5): before a wave reads another rank's bodies it waits
    // (bounded) until every peer's publication counter has reached wait_need; those bodies are then read past this GPU's caches
    const uint32_t *wait_counters;   // this rank's counter array [world] (null: nothing to wait for)
    uint32_t     *wait_status;       // host-visible word: 1 + peer whose slice never arrived
    uint64_t      wait_timeout_ticks;
    uint32_t      wait_need, wait_world, wait_rank, wait_self;   // wait_self: loopback timing only -- the "peers" are this rank
    uint32_t     *wait_dead;         // this rank's SYM_DEAD_WORD
    // ... and CHECKS what they pushed (verify_sums != null: the pushes of publication `verify_epoch` have not been checked yet): the
    // pusher stored one checksum word per 32 bodies (sym_push_checksum) behind its data; every wave of this launch re-computes the
    // checksums of a few groups from what it reads past the caches and reports a mismatch (status 0x200 + sender) instead of
    // integrating positions that are stale, torn or misplaced (the reaction rows carry a tag, the positions cannot)
    const uint32_t *verify_sums;     // this rank's checksum rows [publication parity][sender][count / 32] (uncached region)
    uint32_t      verify_epoch, verify_count;    // publication number the checksums must carry; bodies per rank
    uint32_t      verify_active;                 // bodies [0, verify_active) of the whole job were pushed by that publication (N: all; fewer: a partially active step)
    uint32_t      stage_iblock;   // the workgroup's waves share the I-block's global loads through LDS
    uint32_t      row_wt;     // rows are stored write-through (sc1) as they are produced instead of waiting in L2 for the end-of-kernel write-back
    float         mass, soft2, dt, damping;
    unsigned long long *stamps;   // diagnostic launches only (mapn_measure_clock), else null
    unsigned long long *timeline; // diagnostic launches only (MAPN_STAMP_DUMP): per wave {entry, loop start, loop end, exit (100 MHz), hw id, steps}
};
hipError_t launch_force_sym(const SymArgs &a, uint32_t waves, hipStream_t st);
hipError_t launch_sym_reduce(const SymArgs &a, hipStream_t st);

// The symmetric step SHARDED over ranks (gather algorithm 4): a rank runs the meetings of its own I-blocks
// (force_sym_kernel with a0 / shard_nbl), adds the reactions it produced for every rank's bodies over its
// I-blocks in ascending order and stores ONE row per destination rank straight into that rank's receive
// region (remote stores through the hipIpc mapping, then the flag), waits (bounded) for the flags of the
// ranks that owe it rows, integrates its own bodies from its a-rows plus the rows received, publishes its
// new slice and pulls the peers' slices (sym_shard_exchange_kernel: one launch).
enum { P2P_MAX_RANKS = 16 };          // ranks of a direct peer-to-peer job (one process per GPU, buffers mapped through hipIpc)
// word SYM_DEAD_WORD of a rank's own flag array: set (never cleared: the context stays failed) when a bounded device-side wait of this rank
// gave up or a row / position check failed.  Every LATER bounded wait of the rank reads it once it actually has to wait and gives up at
// once: the launches still queued behind a failure drain in microseconds instead of one time-out each (bench.py's fall-back after a
// failure in the timed run waited 30 launches x 3 s before this).
enum { SYM_DEAD_WORD = 48 };
enum { SYM_FLAG_BASE = 16,            // reaction-arrival counters follow the P2P_MAX_RANKS publication counters of the one-sided exchange
       SYM_POS_BASE = 32,             // position counters of the sharded symmetric step (algorithms 4 / 5)
       SYM_RECV_OFFSET = 4096 };      // byte offset of the receive region [world][count] float4 inside the flags allocation
// The position counters ADVANCE by this much per launch: every workgroup of the exchange launch adds its share once its stores are
// acknowledged (the shares of a launch sum to exactly this, whatever its grid), a waiter needs launch number x this
constexpr uint32_t SYM_COUNT_PER_LAUNCH = 1u << 16;
// Checksum of a pushed position as the pusher holds it in registers: the four words, each in its own rotation, and the body's
// index (a row that lands in the wrong place must not pass); the 32 bodies of a group are XOR-ed, then the publication number is
// mixed in (a group AND its checksum both left over from two steps ago must not pass either).
__host__ __device__ inline uint32_t sym_push_checksum(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t body)
{
    return x ^ ((y << 8) | (y >> 24)) ^ ((z << 16) | (z >> 16)) ^ ((w << 24) | (w >> 8)) ^ (body * 0x9E3779B1u);
}
__host__ __device__ inline uint32_t sym_push_epoch_mix(uint32_t epoch) { return epoch * 0x85EBCA6Bu + 0x1234567u; }
// The tag a reaction row carries in .w when the receiver polls the rows themselves (SymShardArgs::poll_rows): a hash of the row's
// three words and the exchange number, never 0 for the rows a peer-to-peer exchange sends (the receive region starts zeroed).
__host__ __device__ inline uint32_t sym_row_tag(uint32_t x, uint32_t y, uint32_t z, uint32_t step)
{
    const uint32_t h = (x ^ ((y << 11) | (y >> 21)) ^ ((z << 22) | (z >> 10))) * 0x9E3779B1u + step * 0x85EBCA6Bu;
    return h | 1u;
}
// layout of a rank's uncached exchange region (one allocation, one hipIpc handle), in 32-bit words from its start:
//   [0, 1024)  counters;  SYM_RECV_OFFSET: receive rows float4[world][count];  then arrival flags [world][count / 256];
//   then the checksums of pushed positions [2][world][count / 32] -- TWO sets, by the parity of the publication number, like the
//   position buffers themselves: a peer that is one step ahead has already stored the NEXT publication's checksums while this
//   rank has not yet checked the last one (found by the 8-process test: with one set the check raced with the next push)
inline size_t sym_region_chunk_flags_word(uint32_t world, uint32_t count) { return (SYM_RECV_OFFSET + (size_t)world * count * sizeof(float4)) / sizeof(uint32_t); }
inline size_t sym_region_pos_sums_word(uint32_t world, uint32_t count) { return sym_region_chunk_flags_word(world, count) + (size_t)world * ((count + 255u) / 256u); }
inline size_t sym_region_bytes(uint32_t world, uint32_t count) { return (sym_region_pos_sums_word(world, count) + 2u * (size_t)world * ((count + 31u) / 32u)) * sizeof(uint32_t); }

struct SymShardArgs {
    const float4 *pos_old;
    const float  *vel_old;
    float4       *pos_new;
    float        *vel_new;
    const SymRow *arow;                       // [nbl][parts][SYM_BLOCK]
    const SymRow *brow;                       // [n / 64][nbl][64]
    const SymRow *brow1;                      // [nbl][parts][64]
    const uint32_t *tab;                      // the (single) window's tables
    float4       *recv_peer[P2P_MAX_RANKS];   // rank q's receive region as mapped here: row [sender][body of q]
    uint32_t     *flags_peer[P2P_MAX_RANKS];  // rank q's flag array as mapped here
    float4       *pos_peer[P2P_MAX_RANKS];    // rank q's WRITTEN position buffer as mapped here (null: positions travel in another launch)
    uint32_t      phase;                      // 0: the whole exchange in this launch (peer-to-peer); 1: PACK only -- (1), rows stored where recv_peer[]
                                              // points (a local send buffer), no counters; 2: REDUCE only -- (2) and (4) on rows that a
                                              // collective library has delivered into recv_mine (gather algorithm 6)
    uint32_t      release;                    // 1: a system-scope release fence (buffer_wbl2 + wait) in front of the flags / counters
    uint32_t      send_row;                   // row of the destination's region this rank's reactions go to (its rank; 0 when packing)
    uint32_t      push;                       // 1: the new positions are stored into every peer's buffer here (and the peers' NEXT force launch waits for
                                              // the counter); 0: this launch waits for the peers' counters and pulls their slices
    const float4 *recv_mine;
    uint32_t     *flags_mine;
    uint32_t     *ticket;                     // workgroups of this launch whose sends are acknowledged (zero between launches)
    uint32_t      chunk_flags;                // != 0: arrival flags per (sender, 256-body chunk) at this word offset of the flag arrays instead of
                                              // the ticket + one flag per sender
    uint32_t      poll_rows;                  // 1 (round 4): NO arrival flags at all -- every reaction row validates ITSELF: its .w is a hash of its
                                              // x, y, z and the exchange number (sym_row_tag), the receiver re-reads a body's rows (bounded) until
                                              // every one of them carries the tag its contents demand.  One trip through memory (row store -> row
                                              // load) instead of three (row store -> acknowledgement -> flag store -> flag load -> row load); a torn
                                              // or stale row cannot pass, whatever the order in which its bytes arrive
    uint32_t     *status;                     // host-visible word: non-zero = a wait timed out
    uint32_t      rank, world, count;         // count = bodies per rank (a multiple of SYM_BLOCK)
    uint32_t      active, count_active;       // the bodies [0, active) of the whole job advance (N: all of them), count_active of them are this rank's (its first ones)
    const float4 *extra;                      // a PARTIALLY ACTIVE step: [extra_rows][extra_stride] partial force rows of the one-sided launch over this rank's FROZEN bodies
    uint32_t      extra_rows, extra_stride;   // (force_*_kernel, EPI_ROWS; indexed by the body's number in the whole job), added per destination body in front of the reactions (null: none)
    uint32_t      nb, nbl, a0, half_d, parts, nwaves, max_meetings, sets;   // nb: blocks of the (active) job's ring; nbl / a0: this rank's blocks in it (nbl 0: it runs no meetings)
    uint32_t      send_mask, recv_mask;       // bit q: this rank produces reactions for / receives reactions from rank q
    uint32_t      step;                       // monotonically increasing (>= 1): number of the reaction exchange
    uint32_t      pos_step;                   // number of this publication of new positions by a sharded symmetric step (0: they travel in another launch)
    uint32_t      pull_self;                  // loopback timing only: the "peers" are this rank, pull from every slot
    uint32_t      wait_tail;                  // pushed positions, partially active step: this launch also WAITS (bounded) for the peers' pushes of this
                                              // publication and checks them -- the rank's next launch (one-sided, over its frozen bodies) cannot
    uint32_t      pos_sums;                   // push form: word offset of the checksum rows [publication parity][sender][count / 32] in the flag arrays (0: none)
    uint32_t      corrupt_row;                // TEST HOOK (MAPN_TEST_HOOKS=1 MAPN_TEST_CORRUPT_ROW=<exchange>): one bit of ONE reaction row this launch sends
                                              // is flipped after its tag was formed -- the receiver must never accept it (bounded wait, then reported)
    uint32_t      corrupt_push;               // TEST HOOK (MAPN_TEST_HOOKS=1 MAPN_TEST_CORRUPT_PUSH=<publication>): this launch flips one bit of
                                              // ONE pushed position after its checksum was formed -- the peers must report it
    uint64_t      timeout_ticks;
    float         mass, dt, damping;
    unsigned long long *timeline;             // diagnostic launches only (MAPN_STAMP_DUMP): 8 wall-clock stamps per workgroup
};
hipError_t launch_sym_shard_exchange(const SymShardArgs &a, uint32_t max_workgroups, hipStream_t st);
uint32_t sym_shard_exchange_resident_workgroups(uint32_t count, int cus);   // how many of its workgroups `cus` compute units hold at once (with headroom)
int probe_active_compute_units(hipStream_t st);                             // compute units that really take this process's workgroups (0: probe failed)
// stream operation: wait (bounded) until every peer's publication counter has reached `need` (positions pushed by the peers)
// ... and, when verify_sums is given, check the pushed slices of `replica` against the pushers' checksums (as the force launch does)
hipError_t launch_p2p_wait(const uint32_t *counters, uint32_t need, uint32_t world, uint32_t rank, uint32_t self, uint64_t timeout_ticks,
                           uint32_t *status, uint32_t *dead, const float4 *replica, const uint32_t *verify_sums, uint32_t verify_epoch, uint32_t count,
                           uint32_t verify_active, hipStream_t st);

bool force_plan_supported(const ForcePlan &plan);
hipError_t launch_force(const ForcePlan &plan, const StepArgs &a, hipStream_t st);
hipError_t launch_reduce_integrate(const StepArgs &a, uint32_t slots, hipStream_t st);
hipError_t launch_central_well(const StepArgs &a, hipStream_t st);

// Direct peer-to-peer exchange of the new position slices (one process per GPU, peers' buffers
// mapped through hipIpc).  See p2p_gather_kernel.
struct P2PArgs {
    float4       *local;                    // this rank's written position buffer (full N)
    const float4 *peer[P2P_MAX_RANKS];      // every rank's written position buffer, as mapped here
    uint32_t     *peer_flags[P2P_MAX_RANKS];// every rank's flag array [world], as mapped here
    uint32_t     *my_flags;                 // this rank's flag array: my_flags[q] = last step rank q published
    uint32_t     *status;                   // host-visible word: non-zero = a wait timed out
    uint32_t      rank, world, count;       // count = bodies per rank
    uint32_t      step;                     // monotonically increasing publication number (>= 1)
    uint64_t      timeout_ticks;            // s_memrealtime ticks (100 MHz) before a wait gives up
};
hipError_t launch_p2p_gather(const P2PArgs &a, hipStream_t st);
// flow mode: the pull half alone (wait for the peers' flags, pull, mark arrived) and the publish half
// alone (ranks whose slice is not advanced by a force launch in this step)
hipError_t launch_flow_pull(const P2PArgs &a, uint32_t *arrived, hipStream_t st);
hipError_t launch_flow_publish(uint32_t *const *peer_flags, uint32_t rank, uint32_t world, uint32_t step, hipStream_t st);

// the consumer's fence as memory words: a queued GPU-side wait / an event-ordered signal
hipError_t launch_fence_wait(const uint32_t *host_word, const uint32_t *dev_word, uint32_t need, uint64_t timeout_ticks,
                             uint32_t *status, hipStream_t st);
hipError_t launch_fence_signal(uint32_t *dev_word, uint32_t value, hipStream_t st);
hipError_t launch_status_publish(uint32_t *block, uint32_t fence_value, uint32_t latest_index, hipStream_t st);
const char *force_kernel_name(const ForcePlan &plan);

}  // namespace mapn
