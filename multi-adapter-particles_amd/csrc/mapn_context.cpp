// mapn_context.cpp -- host side of libmapn.so: the C ABI of include/mapn.h.
//
// Mirrors the public surface of the reference's `class Compute`
// (reference/Particles/Compute.h:33-78, Compute.cpp) on HIP: device, two streams (compute +
// comm), ping-pong position / velocity buffers, a monotonically increasing fence value backed
// by hipEvents, event timers with the reference's EMA, and the sharded multi-GPU step with an
// RCCL all-gather.  No CPU fallback: without a gfx950 device mapn_create() fails.
// (The symmetric kernel's host side, the sharded exchanges and the state hand-off live in mapn_sym_host.cpp, mapn_shard.cpp and
// mapn_state.cpp; mapn_internal.h is what they share.)
#include "mapn_internal.h"

using namespace mapn::host;

namespace mapn {
namespace host {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

const char *test_hook(const char *name)
{
    const char *on = getenv("MAPN_TEST_HOOKS");
    return (on && on[0] == '1') ? getenv(name) : nullptr;
}

}  // namespace host
}  // namespace mapn

namespace mapn {
namespace host {

int resolve_timers(mapn_ctx *c, bool block)
{
    for (int k = 0; k < kTimerRing; k++) {
        StepTimer &t = c->timers[(c->timer_head + k) % kTimerRing];
        if (!t.pending) continue;
        hipError_t q = hipEventQuery(t.stop);
        if (q == hipErrorNotReady) {
            if (!block) continue;
            HIP_TRY(hipEventSynchronize(t.stop));
        } else if (q != hipSuccess) {
            return fail(MAPN_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
        }
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t.start, t.stop));
        // D3D12GpuTimer.h:151-153: t = t*(averageOver-1); t = (t + delta)/averageOver
        c->ema_seconds = (c->ema_seconds * (kAverageOver - 1) + ms * 1e-3f) / kAverageOver;
        float fms = 0.f;
        if (t.has_force || t.force_is_step) {
            fms = ms;
            if (t.has_force) HIP_TRY(hipEventElapsedTime(&fms, t.start, t.force_done));
            c->force_seconds_sum += fms * 1e-3;
            c->force_launches++;
        }
        if (c->samples.size() < 4096) c->samples.push_back({(uint32_t)t.step_index, ms, fms});
        t.pending = false;
    }
    return MAPN_OK;
}

int update_completed(mapn_ctx *c)
{
    if (c->exported_value > c->completed && hipEventQuery(c->exported_done) == hipSuccess) c->completed = c->exported_value;
    for (int k = 0; k < kTimerRing; k++) {
        const uint64_t v = c->fence_event_value[k];
        if (v > c->completed && c->fence_events[k] && hipEventQuery(c->fence_events[k]) == hipSuccess)
            c->completed = v;
    }
    return MAPN_OK;
}

// Signal(fence, value) on the compute stream
int signal_fence(mapn_ctx *c, uint64_t value)
{
    const int slot = (int)(value % kTimerRing);
    HIP_TRY(hipEventRecord(c->fence_events[slot], c->compute));
    c->fence_event_value[slot] = value;
    c->step_done = c->fence_events[slot];
    return MAPN_OK;
}

// Device-side waits are bounded; one that gave up leaves a word in pinned host memory.  Every
// entry point that hands results to the caller checks it, so a timed-out exchange or consumer
// wait is never a silent MAPN_OK.
int check_async_errors(mapn_ctx *c)
{
    if (!c->async_status) return MAPN_OK;
    const uint32_t p2p = reinterpret_cast<volatile uint32_t *>(c->async_status)[0];
    const uint32_t cons = reinterpret_cast<volatile uint32_t *>(c->async_status)[1];
    if (p2p >= 0x200u)
        return fail(MAPN_ERR_COMM, "sharded symmetric step: the positions rank %u PUSHED into this rank's replica do not match the checksums it stored "
                    "behind them (stale, torn or misplaced data: its counter overtook its stores, or it went on after a timed-out wait of its own); "
                    "this rank's position replica is not to be trusted from that step on", p2p - 0x200u);
    if (p2p >= 0x100u)
        return fail(MAPN_ERR_COMM, "sharded symmetric step: a reaction row of this exchange never arrived whole within %.0f ms, or (flag forms) a row read "
                    "after its sender's flag did not carry this exchange's number (sender %u places behind this rank on the ring): that sender is "
                    "late or gone, went on after a timed-out wait of its own (it reports the timeout) and overwrote the row, or its flag overtook its data",
                    c->p2p_timeout_ticks / 1e5, p2p - 0x100u);
    if (p2p)
        return fail(MAPN_ERR_COMM, "peer-to-peer exchange: the wait for rank %u's slice timed out (%.0f ms); "
                    "this rank's position replica is stale from that step on", p2p - 1u, c->p2p_timeout_ticks / 1e5);
    if (cons)
        return fail(MAPN_ERR_STATE, "the queued wait on the consumer's fence timed out (%.0f ms): the consumer never "
                    "signalled the value Simulate was told to wait for", c->consumer_timeout_ticks / 1e5);
    return MAPN_OK;
}

uint32_t active_bodies(int num_active, uint32_t n)
{
    if (num_active <= 0) return 0;
    const uint64_t groups = ((uint64_t)num_active + kBlock - 1) / kBlock;   // Compute.cpp:1041
    return (uint32_t)std::min<uint64_t>(groups * kBlock, n);
}

// Where the integrator runs: inside the workgroup when it sees all chunks of its bodies, else by
// the last workgroup to arrive at the i-tile's ticket (one launch per step either way).
// MAPN_EPILOGUE=rows selects the two-kernel form (partial rows + reduce_integrate_kernel) for A/B.
int choose_epilogue(const mapn_ctx *c, const mapn::ForcePlan &p, bool allow_fused)
{
    int want = c->plan_forced ? c->forced_epilogue : 1;
    if (c->p2p_ready && c->gather_algo == 3) return mapn::EPI_TICKET;   // flow mode publishes from the ticket epilogue
    const char *e = test_hook("MAPN_EPILOGUE");
    if (!c->plan_forced && e && strcmp(e, "rows") == 0) want = 0;
    if (want == 0) return mapn::EPI_ROWS;
    if (want == 1 && allow_fused && p.sb == 1 && p.nseg == 1) return mapn::EPI_FUSED;
    return mapn::EPI_TICKET;
}

// Plan of the all-pairs launch.  Measured on MI355X (profiles/r01_sweep*.txt): the scalar-cache
// kernel beats the LDS-tiled one by ~5 % at every size; 2 bodies per lane win up to ~128 Ki
// bodies and 4 above; 8 waves per workgroup; and the j-range wants to be cut S = 64..128 ways
// (several rounds of workgroups per CU balance the tail better than one fully resident round:
// at 65 536 bodies S = 64 runs the force kernel in 0.884 ms against 1.018 ms for S = 8).
mapn::ForcePlan choose_plan(const mapn_ctx *c, uint32_t i_count, uint32_t j_total, uint32_t nseg, bool allow_fused)
{
    mapn::ForcePlan p{};
    if (c->plan_forced) {
        p = c->forced_plan;
        p.nseg = nseg;
        p.epi = choose_epilogue(c, p, allow_fused);
        return p;
    }
    p.kind = c->cfg.kernel == MAPN_KERNEL_LDS ? mapn::KERNEL_LDS : mapn::KERNEL_SGPR;      // (SYMMETRIC falls back to the scalar-cache kernel)
    p.k = i_count >= 196608u ? 4 : 2;
    p.nseg = nseg;
    const uint64_t i_waves = (i_count + 64ull * p.k - 1) / (64ull * p.k);
    const uint64_t target = (uint64_t)c->cus * 4 * 32;     // 32768 waves on 256 CUs
    uint64_t S = std::min<uint64_t>(128, std::max<uint64_t>(64, (target + i_waves - 1) / i_waves));
    const uint64_t tiles = std::max<uint64_t>(1, (j_total / std::max(1u, nseg) + 63) / 64);
    S = std::max<uint64_t>(1, std::min<uint64_t>(S, tiles));
    if (S <= 16) {
        uint32_t w = 1;
        while (w < S) w <<= 1;
        p.waves = std::max(w, 4u);
        p.sb = 1;
    } else {
        p.waves = i_waves <= 128 ? 16 : 8;                 // few i-tiles (a shard): 16-wave workgroups
        if (i_waves <= 128) S = std::min<uint64_t>(256, tiles);
        p.sb = (uint32_t)((S + p.waves - 1) / p.waves);
        if (p.sb > 8) p.sb = (p.sb + 7) / 8 * 8;          // multiples of 8 rows: XCD-aware remap
    }
    p.epi = choose_epilogue(c, p, allow_fused);
    return p;
}

// MAPN_OWN_PLAN / MAPN_REM_PLAN = "k,waves,sb": tuning override of the two sharded launches
bool env_plan(const char *name, mapn::ForcePlan &p)
{
    const char *e = test_hook(name);
    unsigned k = 0, w = 0, sb = 0;
    if (!e || sscanf(e, "%u,%u,%u", &k, &w, &sb) != 3) return false;
    mapn::ForcePlan q = p;
    q.k = k; q.waves = w; q.sb = sb;
    if (q.epi == mapn::EPI_FUSED) q.epi = mapn::EPI_TICKET;
    if (!mapn::force_plan_supported(q)) return false;
    p = q;
    return true;
}

int ensure_partial(mapn_ctx *c, size_t slots, size_t stride)
{
    const size_t need = slots * stride * sizeof(float4);
    if (need <= c->partial_bytes) return MAPN_OK;
    // (the buffer moves: steps still queued -- possibly parked behind the consumer's fence -- read the old one, and captured steps hold its
    //  address.  Nothing is waited for here (mapn_simulate only enqueues, Compute.cpp:1009-1055): the old buffer and the graphs are RETIRED and
    //  freed once the stream has run dry, collect_retired)
    float4 *fresh = nullptr;
    HIP_TRY(hipMalloc(&fresh, need));
    retire(c, c->partial, nullptr, nullptr);
    drop_graphs(c);
    c->partial = fresh;
    c->partial_bytes = need;
    return MAPN_OK;
}

void fill_segment(mapn::StepArgs &a, int s, uint32_t first, uint32_t count, uint32_t slot, uint32_t S)
{
    a.seg_first[s] = first;
    a.seg_count[s] = count;
    a.seg_slot[s] = slot;
    const uint32_t tiles = (count + 63u) / 64u;
    a.seg_tiles_base[s] = tiles / S;
    a.seg_tiles_rem[s] = tiles % S;
}

mapn::StepArgs base_args(const mapn_ctx *c, uint32_t w, uint32_t r)
{
    mapn::StepArgs a{};
    a.pos_old = c->pos[r];
    a.vel_old = c->vel[r];
    a.pos_new = c->pos[w];
    a.vel_new = c->vel[w];
    a.mass = c->cfg.mass;
    a.soft2 = c->cfg.softening_squared;
    a.dt = c->cfg.dt;
    a.damping = c->cfg.damping;
    const char *nr = test_hook("MAPN_NO_XCD_REMAP");          // A/B switch for the XCD-aware mapping
    a.xcd_remap = (nr && nr[0] == '1') ? 0u : 1u;
    return a;
}

int wait_for_consumer(mapn_ctx *c, uint64_t wait_value)
{
    if (!c->consumer_enabled || wait_value == 0) return MAPN_OK;
    const uint64_t need = wait_value - 1;                  // Compute.cpp:1012
    if (c->consumer_value >= need) return MAPN_OK;
    // registered events the consumer has already passed are history
    c->consumer_events.erase(std::remove_if(c->consumer_events.begin(), c->consumer_events.end(),
                                            [&](const std::pair<uint64_t, hipEvent_t> &e) { return e.first < need; }),
                             c->consumer_events.end());
    // the consumer's work for `need` is already enqueued on its own stream: wait for its event
    hipEvent_t best = nullptr;
    uint64_t best_v = 0;
    for (auto &e : c->consumer_events)
        if (e.first >= need && (!best || e.first < best_v)) { best = e.second; best_v = e.first; }
    if (best) {
        HIP_TRY(hipStreamWaitEvent(c->compute, best, 0));
        c->consumer_events.erase(std::remove_if(c->consumer_events.begin(), c->consumer_events.end(),
                                                [&](const std::pair<uint64_t, hipEvent_t> &e) { return e.first <= best_v; }),
                                 c->consumer_events.end());
        c->consumer_value = std::max(c->consumer_value, best_v);
        return MAPN_OK;
    }
    if (c->cfg.flags & MAPN_FLAG_STRICT_CONSUMER)
        return fail(MAPN_ERR_STATE, "simulate(wait_value=%llu): consumer has not signalled %llu "
                    "(last %llu); the step would overwrite a buffer still being read",
                    (unsigned long long)wait_value, (unsigned long long)need,
                    (unsigned long long)c->consumer_value);
    // Not signalled and nothing registered yet: queue the wait anyway, exactly like
    // m_commandQueue->Wait(m_sharedRenderFence, v - 1) (Compute.cpp:1012) -- the compute stream
    // parks in a one-lane kernel until the consumer's fence words reach `need`.
    HIP_TRY(mapn::launch_fence_wait(c->fence_host_word, c->fence_dev_block, (uint32_t)need, c->consumer_timeout_ticks,
                                    c->async_status + 1, c->compute));
    c->deferred_need = std::max(c->deferred_need, need);
    return MAPN_OK;
}

// exported context: tell a consumer in another process which buffer holds the results of the step
// that signals `fence_value` (ordered on the compute stream, before the exported event)
int publish_ipc_status(mapn_ctx *c, uint64_t fence_value, uint32_t latest_index)
{
    HIP_TRY(mapn::launch_status_publish(c->fence_dev_block, (uint32_t)fence_value, latest_index, c->compute));
    return MAPN_OK;
}

// Enqueue the kernels of one step on the compute stream (no fence/flip bookkeeping).
int enqueue_step(mapn_ctx *c, uint32_t active, StepTimer *timer)
{
    const uint32_t w = c->buffer_index, r = 1 - c->buffer_index;
    // this shard's active bodies: [first, min(first+count, active))
    const uint32_t lo = c->first, i_count = shard_active_count(c->first, c->count, active);
    mapn::StepArgs a = base_args(c, w, r);
    a.i_first = lo;
    a.i_count = i_count;
    const bool flow = c->p2p_ready && c->gather_algo == 3;
    const bool sharded_native = c->comm != nullptr || (c->p2p_ready && (c->gather_algo == 2 || c->gather_algo == 4 || c->gather_algo == 5));
    const bool overlap = c->comm != nullptr && c->gather_algo < 2 && (c->cfg.flags & MAPN_FLAG_SHARD_OVERLAP);
    if (flow) {
        // the pull half of THIS step's exchange runs beside the launch on the comm stream (it only
        // depends on the peers' flags); the launch's remote chunks need the PREVIOUS exchange
        a.flow_arrived = c->flow_block;
        a.flow_tiles_done = c->flow_block + 16;
        a.flow_peer_flags = c->p2p_flag_table;
        a.flow_status = c->async_status;
        a.flow_timeout_ticks = c->p2p_timeout_ticks;
        a.flow_need = c->p2p_step;
        a.flow_publish = c->p2p_step + 1;
        a.flow_rank = (uint32_t)c->cfg.rank;
        a.flow_world = (uint32_t)c->cfg.world_size;
        a.flow_count = c->count;
        a.xcd_remap = 0;                                   // dispatch order = row order: own rows first
    }

    // a PARTIALLY ACTIVE step of a sharded job in its split form: every rank takes part, also one whose slice is frozen (i_count == 0)
    const bool shard_split = c->cfg.force_mode == MAPN_FORCE_ALL_PAIRS && !flow && sym_shard_split_eligible(c, active);
    if (shard_split) { if (int rc = prepare_sym_active(c, active)) return rc; }   // (this rank's plan for the count: a lookup after the first step with it; an error if it cannot be had)
    if (c->push_pending && !shard_split && !(i_count > 0 && c->cfg.force_mode == MAPN_FORCE_ALL_PAIRS && sym_shard_eligible(c, active) && c->gather_algo == 5))
        if (int rc = settle_push(c)) return rc;            // this step's launch does not wait for the peers' pushes itself

    // an unsharded all-pairs step: full symmetric, split (active x active symmetric + active x frozen one-sided) or one-sided
    StepForm form = FORM_ONE_SIDED;
    if (i_count > 0 && c->cfg.force_mode == MAPN_FORCE_ALL_PAIRS && !flow) {
        form = sym_step_form(c, active);
        if (form == FORM_SYM_SPLIT) {
            if (int rc = prepare_sym_active(c, active)) return rc;    // (a lookup among the cached plans; a new count makes its plan -- without waiting for the device)
            if (!act_ready(c, active)) form = sym_form_without_split(c, active);   // it could not be made (act_failed remembers): another form, whatever the A/B hook says
        }
    }
    c->last_split_active = 0;

    if (timer) HIP_TRY(hipEventRecord(timer->start, c->compute));

    if (shard_split) {
        if (int rc = enqueue_sym_shard_split(c, a, timer)) return rc;
    } else if (i_count > 0 && c->cfg.force_mode == MAPN_FORCE_CENTRAL_WELL) {
        mapn::StepArgs w0 = a;
        w0.flow_arrived = nullptr;                         // plain stores: the flag goes out behind the kernel boundary
        HIP_TRY(mapn::launch_central_well(w0, c->compute));
        if (flow) HIP_TRY(mapn::launch_flow_publish(c->p2p_flag_table, a.flow_rank, a.flow_world, a.flow_publish, c->compute));
    } else if (i_count == 0 && flow) {
        // nothing of this rank's slice advances in this step: it still owes its peers the flag
        HIP_TRY(mapn::launch_flow_publish(c->p2p_flag_table, a.flow_rank, a.flow_world, a.flow_publish, c->compute));
    } else if (i_count > 0 && form == FORM_SYM_FULL) {
        if (int rc = enqueue_sym(c, a, timer)) return rc;
    } else if (i_count > 0 && form == FORM_SYM_SPLIT) {
        if (int rc = enqueue_sym_split(c, a, timer)) return rc;
    } else if (i_count > 0 && sym_shard_eligible(c, active)) {
        if (int rc = c->gather_algo == 6 ? enqueue_sym_shard_rccl(c, a, timer) : enqueue_sym_shard(c, a, timer)) return rc;
    } else if (i_count > 0 && !overlap) {
        // one force launch over all j.  Sharded: the read buffer is complete once the all-gather
        // that filled it has finished (event recorded on the comm stream).
        if (sharded_native && c->gather_recorded[r]) HIP_TRY(hipStreamWaitEvent(c->compute, c->gather_done[r], 0));
        mapn::ForcePlan plan = choose_plan(c, i_count, c->n, 1, true);
        const uint32_t S = plan.sb * plan.waves;
        fill_segment(a, 0, 0, c->n, 0, S);
        if (flow) {
            // rotate the block rows so that the rows holding this rank's own slice are dispatched first
            const uint32_t t_own = c->first / 64u, base = a.seg_tiles_base[0], rem = a.seg_tiles_rem[0];
            const uint32_t c_own = t_own < rem * (base + 1u) ? t_own / (base + 1u) : (base ? rem + (t_own - rem * (base + 1u)) / base : 0u);
            a.flow_row_rot = std::min(c_own / plan.waves, plan.sb - 1u);
        }
        if (plan.epi != mapn::EPI_FUSED) {
            a.partial_stride = (i_count + 63u) & ~63u;
            if (int rc = ensure_partial(c, plan.sb, a.partial_stride)) return rc;   // one row per block row
            a.partial = c->partial;
            a.ticket = c->ticket;
            a.ticket_total = plan.sb;
        }
        if (c->stamp_next && plan.kind == mapn::KERNEL_SGPR) {
            const size_t waves = (size_t)((i_count + 64 * plan.k - 1) / (64 * plan.k)) * plan.sb * plan.waves;
            if (waves > c->stamp_waves) {
                if (c->stamp_buf) HIP_TRY(hipFree(c->stamp_buf));
                c->stamp_buf = nullptr; c->stamp_waves = 0;
                HIP_TRY(hipMalloc(&c->stamp_buf, waves * 16));
                c->stamp_waves = waves;
            }
            HIP_TRY(hipMemsetAsync(c->stamp_buf, 0, c->stamp_waves * 16, c->compute));
            a.stamps = c->stamp_buf;
        }
        HIP_TRY(mapn::launch_force(plan, a, c->compute));
        a.stamps = nullptr;
        if (plan.epi == mapn::EPI_ROWS) {
            if (timer) { HIP_TRY(hipEventRecord(timer->force_done, c->compute)); timer->has_force = true; }
            HIP_TRY(mapn::launch_reduce_integrate(a, plan.sb, c->compute));
        } else if (timer) {
            timer->force_is_step = true;                   // one launch: [start, stop] brackets the force kernel
        }
        c->last_plan = plan; c->last_i_count = i_count; c->last_launches = 1;
    } else if (i_count > 0) {
        // sharded with MAPN_FLAG_SHARD_OVERLAP: own slice first (needs only data this rank wrote),
        // then the remote segments once the all-gather that filled the read buffer has finished.
        // Two launches share the i-tiles' tickets: the last arriver of the second one integrates.
        const uint32_t own_first = c->first, own_count = c->count;
        mapn::ForcePlan own = choose_plan(c, i_count, own_count, 1, false);
        mapn::ForcePlan rem = choose_plan(c, i_count, c->n - own_count, 2, false);
        env_plan("MAPN_OWN_PLAN", own);
        env_plan("MAPN_REM_PLAN", rem);
        rem.epi = own.epi;                                 // both row-producing launches use one hand-off form
        const uint32_t S_own = own.sb * own.waves, S_rem = rem.sb * rem.waves;
        const uint32_t slots = own.sb + 2 * rem.sb;        // partial rows: one per block row per segment
        a.partial_stride = (i_count + 63u) & ~63u;
        if (int rc = ensure_partial(c, slots, a.partial_stride)) return rc;
        a.partial = c->partial;
        a.ticket = c->ticket;
        a.ticket_total = slots;
        fill_segment(a, 0, own_first, own_count, 0, S_own);
        HIP_TRY(mapn::launch_force(own, a, c->compute));
        if (c->gather_recorded[r]) HIP_TRY(hipStreamWaitEvent(c->compute, c->gather_done[r], 0));
        mapn::StepArgs b = a;
        fill_segment(b, 0, 0, own_first, own.sb, S_rem);
        fill_segment(b, 1, own_first + own_count, c->n - own_first - own_count, own.sb + rem.sb, S_rem);
        HIP_TRY(mapn::launch_force(rem, b, c->compute));
        if (own.epi == mapn::EPI_ROWS) {
            if (timer) { HIP_TRY(hipEventRecord(timer->force_done, c->compute)); timer->has_force = true; }
            HIP_TRY(mapn::launch_reduce_integrate(a, slots, c->compute));
        } else if (timer) {
            timer->force_is_step = true;
        }
        c->last_plan = rem; c->last_i_count = i_count; c->last_launches = 2;
    }
    return MAPN_OK;
}

// (deferred: a replay may still be queued -- the executables are destroyed once the stream has run dry, collect_retired)
void drop_graphs(mapn_ctx *c)
{
    for (int b = 0; b < 2; b++)
        for (mapn_ctx::StepGraph &g : c->graphs[b]) {
            retire(c, nullptr, nullptr, g.exec);
            g = mapn_ctx::StepGraph{};
        }
}

// MAPN_FLAG_USE_GRAPH: the step's launches (force [+ reduce/integrate]) are captured once per ping-pong parity and KEY -- the count,
// the form the step takes for it (the A/B hook or a plan that could not be made change the form of the same count) and, split form,
// which plan (a slot of the plan cache is reused) -- and replayed with one hipGraphLaunch; the last kGraphs keys per parity are kept, so
// a slider moving between a few values replays.  Scratch memory is sized before the capture (no allocation inside it; the symmetric
// kernel's was made when the context was created).  A step that carries timer events runs eagerly.
int enqueue_step_graph(mapn_ctx *c, uint32_t active)
{
    const uint32_t w = c->buffer_index;
    int form = -1;
    uint64_t generation = 0;
    const uint32_t lo = c->first, hi = std::min(c->first + c->count, active);
    if (hi > lo && c->cfg.force_mode == MAPN_FORCE_ALL_PAIRS) {
        StepForm f = sym_step_form(c, active);
        if (f == FORM_SYM_SPLIT) {                          // its plan and scratch, outside the capture
            if (int rc = prepare_sym_active(c, active)) return rc;
            if (act_ready(c, active)) generation = c->act_plans[c->act_cur].generation;
            else f = sym_form_without_split(c, active);
        }
        if (f == FORM_ONE_SIDED) {
            mapn::ForcePlan plan = choose_plan(c, hi - lo, c->n, 1, true);
            if (plan.epi != mapn::EPI_FUSED)
                if (int rc = ensure_partial(c, plan.sb, ((hi - lo) + 63u) & ~63u)) return rc;
        }
        form = (int)f;
    }
    mapn_ctx::StepGraph *g = nullptr, *victim = nullptr;   // victim: a free slot, else the least recently replayed one
    for (mapn_ctx::StepGraph &e : c->graphs[w]) {
        if (e.exec && e.active == active && e.form == form && e.generation == generation) { g = &e; break; }
        if (!victim || (victim->exec && (!e.exec || e.used < victim->used))) victim = &e;
    }
    if (!g) {
        retire(c, nullptr, nullptr, victim->exec);
        *victim = mapn_ctx::StepGraph{};
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(c->compute, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_step(c, active, nullptr);
        hipError_t e = hipStreamEndCapture(c->compute, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return fail(MAPN_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&victim->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { victim->exec = nullptr; return fail(MAPN_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
        victim->active = active; victim->form = form; victim->generation = generation;
        victim->plan = c->last_plan; victim->i_count = c->last_i_count; victim->launches = c->last_launches; victim->split_active = c->last_split_active;
        victim->act_slot = c->act_cur;
        g = victim;
    }
    c->last_plan = g->plan; c->last_i_count = g->i_count; c->last_launches = g->launches; c->last_split_active = g->split_active;
    if (g->split_active) c->act_cur = g->act_slot;        // (the plan mapn_get_split_plan describes: the replayed step's)
    g->used = ++c->graph_clock;
    HIP_TRY(hipGraphLaunch(g->exec, c->compute));
    return MAPN_OK;
}

int alloc_state(mapn_ctx *c)
{
    const uint64_t data = (uint64_t)c->n * sizeof(float4);
    c->aligned_data_size = (data + kHeapAlign - 1) / kHeapAlign * kHeapAlign;   // Compute.cpp:185-194
    // A SHARDED context's position heap is read and written by the peers' GPUs through hipIpc mappings (gather algorithms 2 - 5)
    // with system-scope loads and stores: FINE-GRAINED memory is what the HIP memory model promises coherence for at that scope
    // (coarse-grained memory is only guaranteed coherent between agents at kernel boundaries).  MAPN_TEST_HOOKS=1
    // MAPN_SHARD_HEAP=coarse: the plain allocation of rounds 1 - 3, for the A/B.
    const char *hk = test_hook("MAPN_SHARD_HEAP");
    const bool fine = c->cfg.world_size > 1 && !(hk && hk[0] == 'c');
    if (fine) HIP_TRY(hipExtMallocWithFlags(reinterpret_cast<void **>(&c->pos_heap), 2 * c->aligned_data_size, hipDeviceMallocFinegrained));
    else HIP_TRY(hipMalloc(&c->pos_heap, 2 * c->aligned_data_size));
    HIP_TRY(hipMemset(c->pos_heap, 0, 2 * c->aligned_data_size));
    c->pos_own[0] = c->pos_heap;
    c->pos_own[1] = reinterpret_cast<float4 *>(reinterpret_cast<char *>(c->pos_heap) + c->aligned_data_size);
    c->pos[0] = c->pos_own[0];
    c->pos[1] = c->pos_own[1];
    for (int b = 0; b < 2; b++) {
        HIP_TRY(hipMalloc(&c->vel[b], (size_t)c->n * 12));
        HIP_TRY(hipMemset(c->vel[b], 0, (size_t)c->n * 12));
    }
    const size_t tiles = ((size_t)c->n + 127) / 128 + 1;   // i-tiles of the smallest tile (2 bodies per lane)
    HIP_TRY(hipMalloc(&c->ticket, tiles * sizeof(uint32_t)));
    HIP_TRY(hipMemset(c->ticket, 0, tiles * sizeof(uint32_t)));
    return MAPN_OK;
}

int create_common(const mapn_config *cfg, mapn_ctx **out)
{
    if (!cfg || !out) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(mapn_config))
        return fail(MAPN_ERR_INVALID_ARGUMENT, "mapn_config.struct_size %u != %zu", cfg->struct_size, sizeof(mapn_config));
    if (cfg->num_particles == 0) return fail(MAPN_ERR_INVALID_ARGUMENT, "num_particles must be > 0");
    if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "rank %d / world_size %d", cfg->rank, cfg->world_size);
    if (cfg->num_particles % (uint32_t)cfg->world_size)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "world_size %d must divide num_particles %u", cfg->world_size, cfg->num_particles);
    if (cfg->force_mode != MAPN_FORCE_ALL_PAIRS && cfg->force_mode != MAPN_FORCE_CENTRAL_WELL)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "force_mode %d", cfg->force_mode);
    if (cfg->kernel < MAPN_KERNEL_AUTO || cfg->kernel > MAPN_KERNEL_SYMMETRIC)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "kernel %d", cfg->kernel);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MAPN_ERR_NO_DEVICE, "no HIP device visible: libmapn has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", cfg->device, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MAPN_ERR_NO_DEVICE, "device %d is %s; libmapn ships gfx950 code only", cfg->device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(cfg->device));

    mapn_ctx *c = new mapn_ctx();
    c->cfg = *cfg;
    c->n = cfg->num_particles;
    c->device = cfg->device;
    c->cus = prop.multiProcessorCount;
    shard_slice(c->n, (uint32_t)cfg->rank, (uint32_t)cfg->world_size, c->first, c->count);
    c->timers_enabled = true;
    *out = c;

    HIP_TRY(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
    // the exchange stream outranks the compute stream: its few workgroups (RCCL, the pull kernels) must get a slot as soon
    // as one frees up, also while a force launch keeps the device full
    int prio_low = 0, prio_high = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    HIP_TRY(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_high));
    for (int k = 0; k < kTimerRing; k++) {
        HIP_TRY(hipEventCreateWithFlags(&c->fence_events[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&c->timers[k].start));
        HIP_TRY(hipEventCreate(&c->timers[k].force_done));
        HIP_TRY(hipEventCreate(&c->timers[k].stop));
    }
    for (int b = 0; b < 2; b++) HIP_TRY(hipEventCreateWithFlags(&c->gather_done[b], hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->exported_done, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->fence_host_word), 64, hipHostMallocMapped));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->async_status), 64, hipHostMallocMapped));
    memset(c->fence_host_word, 0, 64);
    memset(c->async_status, 0, 64);
    HIP_TRY(hipExtMallocWithFlags(reinterpret_cast<void **>(&c->fence_dev_block), 256, hipDeviceMallocUncached));
    HIP_TRY(hipMemset(c->fence_dev_block, 0, 256));
    // Compute.cpp:434-436: fence created with value 0, m_fenceValue++ -> 1
    c->fence_value = 1;
    if (int rc = alloc_state(c)) return rc;
    // the symmetric kernel's plan and scratch (unsharded contexts; a sharded one prepares when it is wired for exchange
    // algorithm 4): made here so that mapn_simulate never allocates
    if (int rc = prepare_sym(c, false)) return rc;
    // Compute.cpp:563: Initialize ends with WaitForGpu
    return mapn_wait_idle(c);
}

int observe_steps(mapn_ctx *c)
{
    if (c->fence_every_step) return MAPN_OK;
    // from now on every step re-records the exported event; make it cover everything enqueued so far
    c->fence_every_step = true;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->exported_done, c->compute));
    c->exported_value = c->fence_value - 1;
    return MAPN_OK;
}

}  // namespace host
}  // namespace mapn

extern "C" {

int mapn_abi_version(void) { return MAPN_ABI_VERSION; }

int mapn_tuning_abi_version(void) { return MAPN_TUNING_ABI_VERSION; }

const char *mapn_last_error(void) { return g_last_error.c_str(); }

int mapn_config_default(mapn_config *cfg)
{
    if (!cfg) return fail(MAPN_ERR_INVALID_ARGUMENT, "null config");
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->num_particles = 4u * 1024u * 1024u;   // defines.h:45 MAX_NUM_PARTICLES (Particles.cpp default)
    cfg->device = 0;
    cfg->force_mode = MAPN_FORCE_ALL_PAIRS;
    cfg->mass = 70000.0f;                      // nBodyGravityCS.hlsl:38
    cfg->softening_squared = 25.0f;            // nBodyGravityCS.hlsl:37
    cfg->dt = 0.1f;                            // Compute.cpp:545
    cfg->damping = 1.0f;                       // Compute.cpp:546
    cfg->seed = 1;
    cfg->spread = 400.0f;                      // defines.h:42
    cfg->initial_speed = 15.0f;                // defines.h:39
    cfg->flags = 0;
    cfg->kernel = MAPN_KERNEL_AUTO;
    cfg->rank = 0;
    cfg->world_size = 1;
    return MAPN_OK;
}

int mapn_create(const mapn_config *cfg, mapn_ctx **out_ctx)
{
    if (!out_ctx) return fail(MAPN_ERR_INVALID_ARGUMENT, "mapn_create: out_ctx is null");
    *out_ctx = nullptr;
    mapn_ctx *c = nullptr;
    int rc = create_common(cfg, &c);
    if (rc) { std::string keep = g_last_error; if (c) mapn_destroy(c); g_last_error = keep; return rc; }
    if (!(cfg->flags & MAPN_FLAG_NO_INIT)) {
        // Compute.cpp:820-923 InitializeParticles: generate, upload to both buffers, WaitForGpu
        std::vector<float> pos((size_t)c->n * 4), vel((size_t)c->n * 3);
        rc = mapn_generate_initial_state_ex(cfg->init_variant, cfg->seed, c->n, cfg->spread, cfg->initial_speed, pos.data(), vel.data());
        if (rc) fail(rc, "init_variant %d", cfg->init_variant);
        if (!rc) rc = mapn_upload_state(c, pos.data(), vel.data());
    }
    if (!rc) rc = calibrate_at_creation(c);    // (MAPN_FLAG_XCD_CALIBRATE only; state and fence value come back unchanged)
    if (!rc) rc = mapn_wait_idle(c);           // Compute.cpp:922
    if (!rc) rc = mapn_wait_idle(c);           // Compute.cpp:97
    if (rc) { std::string keep = g_last_error; mapn_destroy(c); g_last_error = keep; *out_ctx = nullptr; return rc; }
    *out_ctx = c;
    return MAPN_OK;
}

int mapn_create_from(const mapn_config *cfg, mapn_ctx *old, mapn_ctx **out_ctx)
{
    if (!out_ctx) return fail(MAPN_ERR_INVALID_ARGUMENT, "mapn_create_from: out_ctx is null");
    *out_ctx = nullptr;
    if (!old) return fail(MAPN_ERR_INVALID_ARGUMENT, "null source context");
    if (!cfg || cfg->num_particles != old->n)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "create_from: num_particles must match the source context");
    // Compute.cpp:305: the source leaves async mode first; Particles.cpp:467-471 drains it
    int rc = mapn_reset_from_async(old);
    if (!rc) rc = mapn_wait_idle(old);
    if (rc) return rc;
    mapn_ctx *c = nullptr;
    rc = create_common(cfg, &c);
    if (rc) { std::string keep = g_last_error; if (c) mapn_destroy(c); g_last_error = keep; return rc; }
    // Compute.cpp:303-410 CopyState: both position buffers, both velocity buffers, buffer index
    for (int b = 0; b < 2 && !rc; b++) {
        if (hipMemcpyPeer(c->pos[b], c->device, old->pos[b], old->device, (size_t)c->n * 16) != hipSuccess ||
            hipMemcpyPeer(c->vel[b], c->device, old->vel[b], old->device, (size_t)c->n * 12) != hipSuccess)
            rc = fail(MAPN_ERR_HIP, "CopyState: device-to-device copy failed: %s", hipGetErrorString(hipGetLastError()));
    }
    c->buffer_index = old->buffer_index;
    (void)hipSetDevice(c->device);
    if (!rc) rc = calibrate_at_creation(c);    // (MAPN_FLAG_XCD_CALIBRATE only; the copied state comes back unchanged)
    if (!rc) rc = mapn_wait_idle(c);           // Compute.cpp:354 / :409
    if (!rc) rc = mapn_wait_idle(c);           // Compute.cpp:97
    if (rc) { std::string keep = g_last_error; mapn_destroy(c); g_last_error = keep; *out_ctx = nullptr; return rc; }
    *out_ctx = c;
    return MAPN_OK;
}

int mapn_destroy(mapn_ctx *c)
{
    if (!c) return MAPN_OK;
    (void)hipSetDevice(c->device);
    if (c->compute) (void)hipStreamSynchronize(c->compute);          // Compute.cpp:104 WaitForGpu first
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm) mapn::comm_destroy(c->comm);
    for (int q = 0; q < mapn::P2P_MAX_RANKS; q++) {
        if (q == c->cfg.rank || c->p2p_loopback) continue;
        if (c->p2p_peer_heap[q]) (void)hipIpcCloseMemHandle(c->p2p_peer_heap[q]);
        if (c->p2p_peer_flags[q]) (void)hipIpcCloseMemHandle(c->p2p_peer_flags[q]);
    }
    if (c->p2p_flags) (void)hipFree(c->p2p_flags);
    if (c->p2p_flag_table) (void)hipFree(c->p2p_flag_table);
    if (c->flow_block) (void)hipFree(c->flow_block);
    if (c->sym_shard_ticket) (void)hipFree(c->sym_shard_ticket);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->exported_done) (void)hipEventDestroy(c->exported_done);
    if (c->fence_host_word) (void)hipHostFree(c->fence_host_word);
    if (c->async_status) (void)hipHostFree(c->async_status);
    if (c->fence_dev_block) (void)hipFree(c->fence_dev_block);
    drop_graphs(c);
    collect_retired(c, true);                                          // (the streams have been drained above)
    for (int b = 0; b < 2; b++) {
        if (c->vel[b]) (void)hipFree(c->vel[b]);
        if (c->gather_done[b]) (void)hipEventDestroy(c->gather_done[b]);
    }
    if (c->pos_heap) (void)hipFree(c->pos_heap);
    if (c->partial) (void)hipFree(c->partial);
    if (c->ticket) (void)hipFree(c->ticket);
    if (c->stamp_buf) (void)hipFree(c->stamp_buf);
    if (c->timeline_buf) (void)hipFree(c->timeline_buf);
    if (c->xtimeline_buf) (void)hipFree(c->xtimeline_buf);
    release_sym(c);
    release_sym_active(c);
    for (int k = 0; k < kTimerRing; k++) {
        if (c->fence_events[k]) (void)hipEventDestroy(c->fence_events[k]);
        if (c->timers[k].start) (void)hipEventDestroy(c->timers[k].start);
        if (c->timers[k].force_done) (void)hipEventDestroy(c->timers[k].force_done);
        if (c->timers[k].stop) (void)hipEventDestroy(c->timers[k].stop);
    }
    if (c->compute) (void)hipStreamDestroy(c->compute);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    delete c;
    return MAPN_OK;
}

int mapn_simulate(mapn_ctx *c, int num_active, uint64_t wait_value)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (c->cfg.world_size > 1 && !c->comm && !c->external_gather && !(c->p2p_ready && c->gather_algo >= 2))
        return fail(MAPN_ERR_STATE, "sharded context (world_size %d): call mapn_comm_init or "
                    "mapn_set_external_gather before simulate", c->cfg.world_size);
    if (int rc = check_async_errors(c)) return rc;                     // a device-side wait of an earlier step gave up
    collect_retired(c, false);                                         // (what a growth or a re-plan replaced, once nothing queued can still use it)
    if (int rc = wait_for_consumer(c, wait_value)) return rc;          // Compute.cpp:1012
    const uint32_t active = active_bodies(num_active, c->n);

    StepTimer *timer = nullptr;
    if (c->timers_enabled && (c->steps_enqueued++ % c->timer_interval) == 0) {
        timer = &c->timers[c->timer_head];
        if (timer->pending) { if (int rc = resolve_timers(c, true)) return rc; }
        timer->has_force = false;
        timer->force_is_step = false;
        timer->step_index = c->steps_since_reset;
    }
    c->steps_since_reset++;
    const bool use_graph = (c->cfg.flags & MAPN_FLAG_USE_GRAPH) && !c->comm && !c->p2p_ready && !timer && active > 0 &&
                           !c->stamp_next;                 // a stamped diagnostic step is never a replay
    if (int rc = use_graph ? enqueue_step_graph(c, active) : enqueue_step(c, active, timer)) return rc;
    if (timer) {
        HIP_TRY(hipEventRecord(timer->stop, c->compute));              // Compute.cpp:1046-1047
        timer->pending = true;
        c->timer_head = (c->timer_head + 1) % kTimerRing;
    }
    // MoveToNextFrame, Compute.cpp:993-1004: Signal(fence, v); v++; index = 1 - index
    const bool exchanging = c->comm != nullptr || (c->p2p_ready && c->gather_algo >= 2);
    const bool gather_first = exchanging && (!(c->cfg.flags & MAPN_FLAG_SHARD_OVERLAP) || c->gather_algo >= 2);   // (the overlap structure exists for algorithms 0 / 1 only)
    if (gather_first) { if (int rc = enqueue_gather(c)) return rc; }   // same stream: the fence then covers the gather
    // The fence value always advances.  While somebody can observe completion (an attached consumer,
    // exported handles) the ONE exported event is re-recorded after every step; the ring event behind
    // mapn_completed_value() (a few us of queue time each) is recorded when the overlap structure
    // needs it and otherwise on every 16th step, which only makes mapn_completed_value()
    // conservative.  mapn_wait_idle() always records.
    if (c->fence_every_step || c->consumer_enabled) {
        if (c->ipc_exported) { if (int rc = publish_ipc_status(c, c->fence_value, c->buffer_index)) return rc; }
        HIP_TRY(hipEventRecord(c->exported_done, c->compute));
        c->exported_value = c->fence_value;
    }
    const bool record = (exchanging && !gather_first) || (c->fence_value % 16) == 0;
    if (record) { if (int rc = signal_fence(c, c->fence_value)) return rc; }
    if (exchanging && !gather_first) { if (int rc = enqueue_gather(c)) return rc; }   // overlap: behind the fence event
    c->fence_value++;
    c->buffer_index = 1 - c->buffer_index;
    if (timer && (c->timer_head % 16) == 0) (void)resolve_timers(c, false);
    return MAPN_OK;
}

uint64_t mapn_fence_value(const mapn_ctx *c) { return c ? c->fence_value : 0; }

uint64_t mapn_completed_value(mapn_ctx *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    update_completed(c);
    return c->completed;
}

int mapn_wait_idle(mapn_ctx *c)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = settle_push(c)) return rc;                           // (gather algorithm 5: the peers' latest pushes belong to "all enqueued work")
    // Compute.cpp:928-940: Signal(fence, v); v++; wait
    if (int rc = signal_fence(c, c->fence_value)) return rc;
    const uint64_t v = c->fence_value++;
    HIP_TRY(hipStreamSynchronize(c->compute));
    HIP_TRY(hipStreamSynchronize(c->comm_stream));
    collect_retired(c, true);
    c->completed = std::max(c->completed, v);
    if (int rc = resolve_timers(c, true)) return rc;
    return check_async_errors(c);
}

uint32_t mapn_buffer_index(const mapn_ctx *c) { return c ? c->buffer_index : 0; }

uint32_t mapn_num_particles(const mapn_ctx *c) { return c ? c->n : 0; }

int mapn_get_shared_handles(mapn_ctx *c, mapn_shared_handles *out)
{
    if (!c || !out) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    if (int rc = observe_steps(c)) return rc;
    out->positions[0] = c->pos[0];
    out->positions[1] = c->pos[1];
    out->step_done_event = c->exported_done;                           // one handle, valid until mapn_destroy
    out->aligned_data_size = c->aligned_data_size;
    out->buffer_index = c->buffer_index;                               // Compute.cpp:948
    out->reserved = 0;
    return MAPN_OK;
}

int mapn_set_consumer(mapn_ctx *c, int enabled)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    c->consumer_enabled = enabled != 0;
    return MAPN_OK;
}

int mapn_consumer_signal(mapn_ctx *c, uint64_t value)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    c->consumer_value = std::max(c->consumer_value, value);
    // also where a wait that Simulate has ALREADY queued on the device can see it
    __atomic_store_n(c->fence_host_word, (uint32_t)c->consumer_value, __ATOMIC_RELEASE);
    return MAPN_OK;
}

int mapn_consumer_signal_event(mapn_ctx *c, uint64_t value, void *hip_event)
{
    if (!c || !hip_event) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    c->consumer_events.emplace_back(value, static_cast<hipEvent_t>(hip_event));
    if (c->deferred_need > c->consumer_value) {
        // a Simulate is already parked on the device waiting for this value: release it when the event fires
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamWaitEvent(c->aux_stream, static_cast<hipEvent_t>(hip_event), 0));
        HIP_TRY(mapn::launch_fence_signal(c->fence_dev_block, (uint32_t)value, c->aux_stream));
    }
    return MAPN_OK;
}

int mapn_set_timeouts(mapn_ctx *c, uint32_t p2p_ms, uint32_t consumer_ms)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    if (p2p_ms) c->p2p_timeout_ticks = (uint64_t)p2p_ms * 100000ull;         // s_memrealtime runs at 100 MHz
    if (consumer_ms) c->consumer_timeout_ticks = (uint64_t)consumer_ms * 100000ull;
    return MAPN_OK;
}

int mapn_adopt_position_buffers(mapn_ctx *c, void *buffers[2], uint32_t buffer_index)
{
    if (!c || !buffers || !buffers[0] || !buffers[1] || buffer_index > 1)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "adopt_position_buffers: bad argument");
    // sharded (any transport): peers pull this rank's slice from the context's OWN heap and the
    // exchange writes into pos[], so computing into foreign buffers would freeze every replica
    if (c->cfg.world_size > 1 || c->comm || c->p2p_ready || c->external_gather)
        return fail(MAPN_ERR_STATE, "adopt_position_buffers is not available in sharded mode (world_size %d)", c->cfg.world_size);
    if (int rc = mapn_wait_idle(c)) return rc;
    // Compute.cpp:956-987 SetAsync: take the consumer's two buffers; next write = 1 - its index
    c->pos[0] = static_cast<float4 *>(buffers[0]);
    c->pos[1] = static_cast<float4 *>(buffers[1]);
    c->adopted = true;
    c->buffer_index = 1 - buffer_index;
    drop_graphs(c);
    return MAPN_OK;
}

int mapn_reset_from_async(mapn_ctx *c)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    if (!c->adopted) return MAPN_OK;                                   // Compute.cpp:262-265
    if (int rc = mapn_wait_idle(c)) return rc;
    // Compute.cpp:260-298: copy the state back into our own buffers and use those again
    for (int b = 0; b < 2; b++)
        HIP_TRY(hipMemcpy(c->pos_own[b], c->pos[b], (size_t)c->n * 16, hipMemcpyDeviceToDevice));
    c->pos[0] = c->pos_own[0];
    c->pos[1] = c->pos_own[1];
    c->adopted = false;
    drop_graphs(c);
    return MAPN_OK;
}

float mapn_last_step_seconds(mapn_ctx *c)
{
    if (!c) return 0.f;
    (void)hipSetDevice(c->device);
    resolve_timers(c, false);
    return c->ema_seconds;
}

const char *mapn_timer_name(void) { return "simulate ms"; }              // Compute.cpp:446

int mapn_set_use_intel_command_queue_extension(mapn_ctx *, int) { return MAPN_OK; }

int mapn_get_using_intel_command_queue_extension(const mapn_ctx *) { return 0; }

int mapn_get_is_uma(const mapn_ctx *) { return 0; }

void *mapn_compute_stream(mapn_ctx *c) { return c ? c->compute : nullptr; }

int mapn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mapn_get_device_info(int device, mapn_device_info *out)
{
    if (!out) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", p.name);
    snprintf(out->arch, sizeof out->arch, "%s", p.gcnArchName);
    out->compute_units = p.multiProcessorCount;
    out->clock_khz = p.clockRate;
    out->wavefront_size = p.warpSize;
    out->peak_fp32_flops = (double)p.multiProcessorCount * (double)p.clockRate * 1e3 * 256.0;
    out->total_memory_bytes = p.totalGlobalMem;
    return MAPN_OK;
}

int mapn_set_force_plan(mapn_ctx *c, int kernel, uint32_t bodies_per_lane, uint32_t waves, uint32_t sb, int fused)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    if (kernel == MAPN_KERNEL_AUTO) { c->plan_forced = false; drop_graphs(c); return MAPN_OK; }
    mapn::ForcePlan p{};
    p.kind = kernel == MAPN_KERNEL_SCALAR ? mapn::KERNEL_SGPR : mapn::KERNEL_LDS;
    if (fused < 0 || fused > 2) return fail(MAPN_ERR_INVALID_ARGUMENT, "set_force_plan: fused must be 0 (two kernels), 1 (one launch) or 2 (ticket form even when one workgroup sees all chunks)");
    p.k = bodies_per_lane; p.waves = waves; p.sb = sb; p.nseg = 1;
    p.epi = fused == 0 ? mapn::EPI_ROWS : (fused == 1 && sb == 1 ? mapn::EPI_FUSED : mapn::EPI_TICKET);
    if (!mapn::force_plan_supported(p))
        return fail(MAPN_ERR_INVALID_ARGUMENT, "unsupported force plan kernel=%d k=%u waves=%u sb=%u", kernel, bodies_per_lane, waves, sb);
    c->forced_plan = p;
    c->forced_epilogue = fused;
    c->plan_forced = true;
    drop_graphs(c);
    return MAPN_OK;
}

int mapn_measure_clock(mapn_ctx *c, int steps, mapn_clock_info *out)
{
    if (!c || !out || steps < 1) return fail(MAPN_ERR_INVALID_ARGUMENT, "measure_clock: bad argument");
    memset(out, 0, sizeof *out);
    if (c->cfg.force_mode != MAPN_FORCE_ALL_PAIRS) return fail(MAPN_ERR_STATE, "measure_clock: all-pairs mode only");
    {
        // refuse BEFORE any step is taken: only the scalar-cache and the symmetric force kernels carry the stamps
        const bool sym = sym_eligible(c, c->n) || sym_shard_eligible(c, c->n);
        const int kind = c->plan_forced ? c->forced_plan.kind : (c->cfg.kernel == MAPN_KERNEL_LDS ? mapn::KERNEL_LDS : mapn::KERNEL_SGPR);
        if (!sym && kind != mapn::KERNEL_SGPR)
            return fail(MAPN_ERR_STATE, "measure_clock: the stamped diagnostic exists for the scalar-cache and the symmetric force kernels only");
    }
    HIP_TRY(hipSetDevice(c->device));
    c->stamp_next = true;
    int rc = MAPN_OK;
    for (int s = 0; s < steps && !rc; s++) rc = mapn_simulate(c, (int)c->n, 0);
    c->stamp_next = false;
    if (!rc) rc = mapn_wait_idle(c);
    if (rc) return rc;
    if (!c->stamp_buf || (c->last_plan.kind != mapn::KERNEL_SGPR && c->last_plan.kind != mapn::KERNEL_SYM))
        return fail(MAPN_ERR_STATE, "measure_clock: the stamped diagnostic exists for the scalar-cache and the symmetric force kernels only");
    const size_t waves = c->last_plan.kind == mapn::KERNEL_SYM
        ? (size_t)((c->last_i_count + mapn::SYM_BLOCK - 1) / mapn::SYM_BLOCK) * c->sym_parts * c->sym_waves   // (sharded: this rank's blocks)
        : (size_t)((c->last_i_count + 64 * c->last_plan.k - 1) / (64 * c->last_plan.k)) * c->last_plan.sb * c->last_plan.waves;
    std::vector<unsigned long long> h(2 * waves);
    HIP_TRY(hipMemcpy(h.data(), c->stamp_buf, waves * 16, hipMemcpyDeviceToHost));
    if (const char *dump = test_hook("MAPN_STAMP_DUMP")) {
        if (c->timeline_buf && c->timeline_last && c->last_plan.kind == mapn::KERNEL_SYM) {
            std::vector<unsigned long long> tl(6 * c->timeline_last);
            HIP_TRY(hipMemcpy(tl.data(), c->timeline_buf, c->timeline_last * 48, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(dump, "wb")) {
                const unsigned long long hdr[4] = {c->timeline_last, c->sym_parts, c->sym_waves, (unsigned long long)c->cfg.rank};
                fwrite(hdr, 8, 4, f);
                fwrite(tl.data(), 8, tl.size(), f);
                if (c->xtimeline_buf) {
                    std::vector<unsigned long long> xt(4096 * 8);
                    if (hipMemcpy(xt.data(), c->xtimeline_buf, xt.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) fwrite(xt.data(), 8, xt.size(), f);
                }
                fclose(f);
            }
        }
    }
    std::vector<double> ghz, cyc;
    for (size_t w = 0; w < waves; w++)
        if (h[2 * w + 1] > 1000) { ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1); cyc.push_back((double)h[2 * w]); }
    if (ghz.empty()) return fail(MAPN_ERR_STATE, "measure_clock: no wave ran long enough to stamp");
    std::sort(ghz.begin(), ghz.end());
    std::sort(cyc.begin(), cyc.end());
    out->shader_clock_ghz = ghz[ghz.size() / 2];
    out->shader_clock_ghz_p10 = ghz[ghz.size() / 10];
    out->shader_clock_ghz_p90 = ghz[ghz.size() * 9 / 10];
    out->median_wave_cycles = cyc[cyc.size() / 2];
    out->waves_stamped = (uint32_t)ghz.size();
    out->steps = (uint32_t)steps;
    return MAPN_OK;
}

int mapn_set_timers(mapn_ctx *c, int interval)
{
    if (!c || interval < 0) return fail(MAPN_ERR_INVALID_ARGUMENT, "set_timers: bad argument");
    if (int rc = mapn_wait_idle(c)) return rc;
    c->timers_enabled = interval != 0;
    c->timer_interval = interval > 0 ? (uint32_t)interval : 1;
    c->steps_enqueued = 0;
    return MAPN_OK;
}

int mapn_get_kernel_stats(mapn_ctx *c, int reset, mapn_kernel_stats *out)
{
    if (!c || !out) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = resolve_timers(c, true)) return rc;
    memset(out, 0, sizeof *out);
    // the plan enqueue_step launched last (sharded overlap: the remote-segments launch), or, before
    // any step, the plan the next full step would use
    mapn::ForcePlan p = c->last_plan;
    uint32_t i_count = c->last_i_count;
    if (c->last_launches == 0) { i_count = c->count; p = choose_plan(c, i_count, c->n, 1, true); }
    snprintf(out->kernel_name, sizeof out->kernel_name, "%s", p.kind == mapn::KERNEL_SYM ? "force_sym_kernel" : mapn::force_kernel_name(p));
    out->launches = c->force_launches;
    out->avg_seconds = c->force_launches ? c->force_seconds_sum / (double)c->force_launches : 0.0;
    out->grid_x = (i_count + 64 * p.k - 1) / (64 * p.k);
    out->grid_y = p.sb;
    out->grid_z = p.nseg;
    out->block_x = 64 * p.waves;
    out->bodies_per_lane = p.k;
    out->j_splits = p.sb * p.waves;
    out->fused = p.epi != mapn::EPI_ROWS ? 1u : 0u;
    out->epilogue = (uint32_t)p.epi;
    out->force_launches_per_step = c->last_launches ? c->last_launches : 1u;
    if (p.kind == mapn::KERNEL_SYM) {
        out->force_launches_per_step = std::max(1u, c->last_launches / 2u);   // every force launch (one per window of partner distance) is followed by a reduce launch (fused = 0); sharded: by the exchange launch
        out->grid_x = (i_count + mapn::SYM_BLOCK - 1) / mapn::SYM_BLOCK; out->grid_y = p.sb; out->j_splits = p.sb * p.waves; out->fused = 0; out->epilogue = 3;   // grid (I-blocks, parts)
    }
    out->split_active = c->last_split_active;
    out->split_plans_built = c->split_plans_built;
    if (reset) { c->force_launches = 0; c->force_seconds_sum = 0.0; c->steps_since_reset = 0; c->samples.clear(); }
    return MAPN_OK;
}

int mapn_get_step_samples(mapn_ctx *c, uint32_t *step_index, float *step_ms, float *force_ms, uint32_t capacity, uint32_t *count)
{
    if (!c || !count) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = resolve_timers(c, true)) return rc;
    std::sort(c->samples.begin(), c->samples.end(), [](const mapn_ctx::StepSample &a, const mapn_ctx::StepSample &b) { return a.step < b.step; });
    *count = (uint32_t)c->samples.size();
    for (uint32_t k = 0; k < *count && k < capacity; k++) {
        if (step_index) step_index[k] = c->samples[k].step;
        if (step_ms) step_ms[k] = c->samples[k].step_ms;
        if (force_ms) force_ms[k] = c->samples[k].force_ms;
    }
    return MAPN_OK;
}

}  // extern "C"
