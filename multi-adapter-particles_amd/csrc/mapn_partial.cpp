// mapn_partial.cpp -- PARTIALLY ACTIVE steps (num_active < N: Particles.cpp:391-394's slider -> Compute.cpp:1041: bodies [0, roundup64(num_active))
// advance, the rest stay frozen in both buffers and still exert force): which of the three forms an unsharded step takes, the SPLIT form
// (active x active under the symmetric kernel with a plan of the active blocks, active x frozen one-sided), its sharded counterpart
// (gather algorithms 4 / 5: the active ring run by its owners, every owner of frozen bodies computing what those do to all active ones),
// the cache of per-count plans and the deferred frees that keep mapn_simulate from ever waiting for the device.
#include "mapn_internal.h"

using namespace mapn::host;

namespace mapn {
namespace host {

// The cost model behind sym_step_form, a pure function of (N, active) -- also behind the device-less mapn_step_form_describe, which the
// CPU tests hold against the measured sweep (profiles/r05_partial_active_sweep.txt).  Costs in pair evaluations at the symmetric
// kernel's rate (measured, flat over 65 536 .. 4 194 304 bodies: 7.1e12 ordered pairs / s against 4.9e12 for the one-sided kernel):
// full N^2; one-sided 1.45 A N; split roundup1024(A)^2 + 1.45 A (N - A) + the extra launch and the frozen rows' pass through the
// reduce launch (about 17 us).
StepForm sym_form_by_cost(uint32_t n, uint32_t active)
{
    if (active >= n) return FORM_SYM_FULL;
    // (the full form stays the faster one against the one-sided kernel down to about 0.7 N active bodies)
    const bool full_ok = (uint64_t)active * 4u >= (uint64_t)n * 3u;
    const double ratio = 7.1 / 4.9, N = (double)n, A = (double)active;
    const double Ap = (double)(((uint64_t)active + mapn::SYM_BLOCK - 1u) / mapn::SYM_BLOCK * mapn::SYM_BLOCK);
    const double one = ratio * A * N, full = N * N, split = Ap * Ap + ratio * A * (N - A) + 1.2e8;
    const double other = full_ok ? full : one;
    // (3 % in hand: the symmetric kernel over an awkward block count -- one that fills no whole rounds of the compute units -- runs up to
    //  7 % behind the model, e.g. 61 440 of 65 536 bodies active: split 0.644 ms against 0.621 for the full form; profiles/r05_partial_active_sweep.txt;
    //  fewer than 8 blocks do not fill the device under the symmetric kernel at all)
    if (active >= 8u * mapn::SYM_BLOCK && split < 0.97 * other) return FORM_SYM_SPLIT;
    return full_ok ? FORM_SYM_FULL : FORM_ONE_SIDED;
}

// Which form an unsharded all-pairs step with `active` = roundup64(num_active) bodies runs (a pure function of the context's shape and
// `active`: a given (N, num_active) always runs the same form, i.e. the same summation order):
//   FORM_SYM_FULL   the symmetric kernel over all N bodies; the reduce launch stops at `active` (the frozen bodies still exert force)
//   FORM_SYM_SPLIT  active x active under the symmetric kernel with a plan of the ACTIVE blocks only, active x frozen one-sided
//                   (enqueue_sym_split) -- every evaluation that feeds only frozen bodies is dropped
//   FORM_ONE_SIDED  active x N through the scalar-cache kernel
// The choice is sym_form_by_cost above; here: whether the symmetric kernel runs in this context at all, the A/B hook, and a count
// whose split plan could not be made.
StepForm sym_step_form(const mapn_ctx *c, uint32_t active)
{
    if (!c->sym_ready || c->sym_sharded || c->plan_forced || active == 0) return FORM_ONE_SIDED;
    if (c->comm || c->external_gather || c->p2p_ready) return FORM_ONE_SIDED;   // a context wired for an exchange runs the sharded step
    if (active >= c->n) return FORM_SYM_FULL;
    // (the full form stays the faster one against the one-sided kernel down to about 0.7 N active bodies)
    const bool full_ok = (uint64_t)active * 4u >= (uint64_t)c->n * 3u;
    if (const char *f = test_hook("MAPN_PARTIAL_FORM")) {                       // A/B: "one", "full", "split"
        if (f[0] == 'o') return FORM_ONE_SIDED;
        if (f[0] == 'f') return FORM_SYM_FULL;
        if (f[0] == 's' && active >= 2u * mapn::SYM_BLOCK) return FORM_SYM_SPLIT;
    }
    const StepForm form = sym_form_by_cost(c->n, active);
    if (form == FORM_SYM_SPLIT && std::find(c->act_failed.begin(), c->act_failed.end(), active) != c->act_failed.end())
        return full_ok ? FORM_SYM_FULL : FORM_ONE_SIDED;   // (its plan or scratch could not be had for this count)
    return form;
}

// the form a partially active step takes when its split plan could not be made (also when the A/B hook asked for the split form)
StepForm sym_form_without_split(const mapn_ctx *c, uint32_t active) { return (uint64_t)active * 4u >= (uint64_t)c->n * 3u ? FORM_SYM_FULL : FORM_ONE_SIDED; }

// this STEP runs the symmetric kernel over the whole job (all bodies active, or so many that the frozen ones are not worth a split)
bool sym_eligible(const mapn_ctx *c, uint32_t active) { return sym_step_form(c, active) == FORM_SYM_FULL; }

// ---- the PARTIALLY ACTIVE step, split form ----------------------------------------------------------
// The reference lets the user simulate any count of the bodies (Particles.cpp:391-394 -> Compute.cpp:1041: bodies
// [0, roundup64(num_active)) advance, the rest stay frozen but still exert force).  Until round 4 every such step with fewer than
// 0.75 N active bodies ran the one-sided kernel over active x N ordered pairs.  Of those only active x FROZEN has to be one-sided:
// the active bodies meet EACH OTHER under the symmetric kernel -- a plan of the active blocks alone (the bodies past A are the
// kernel's far-away stand-ins), its own scratch -- and the one-sided launch over the frozen j-segment [A, N) leaves partial force
// rows (EPI_ROWS) that the first window's reduce launch adds, in ascending row order, in front of its own rows.  Summation order
// (what the order-matched checker restates): frozen rows as in mapn_set_force_plan's comment with the j-range [A, N), then the
// symmetric plan's order over the bodies [0, A); the mass multiplies the total.
// (destruction: the streams have been drained)
void release_sym_active(mapn_ctx *c)
{
    for (mapn_ctx::SymActive &s : c->act_plans) {
        if (s.tab) (void)hipFree(s.tab);
        if (s.stage) (void)hipHostFree(s.stage);
        if (s.uploaded) (void)hipEventDestroy(s.uploaded);
        s = mapn_ctx::SymActive{};
    }
    mapn_ctx::SymActiveRows &r = c->act_rows;
    if (r.arow) (void)hipFree(r.arow);
    if (r.brow) (void)hipFree(r.brow);
    if (r.brow1) (void)hipFree(r.brow1);
    if (r.acc) (void)hipFree(r.acc);
    r = mapn_ctx::SymActiveRows{};
    c->act_cur = -1;
}

// the context's own plan has changed (weights, shape): the cached split plans were made under the old one -- their buffers stay
void forget_sym_active(mapn_ctx *c)
{
    for (mapn_ctx::SymActive &s : c->act_plans) s.active = 0;
    c->act_cur = -1;
    c->act_failed.clear();
}

bool act_ready(const mapn_ctx *c, uint32_t active) { return c->act_cur >= 0 && c->act_plans[c->act_cur].active == active; }

void retire(mapn_ctx *c, void *dev, void *host, hipGraphExec_t graph)
{
    if (!dev && !host && !graph) return;
    mapn_ctx::Retired r;
    r.dev = dev; r.host = host; r.graph = graph;
    c->retired.push_back(r);
}

void collect_retired(mapn_ctx *c, bool drained)
{
    if (c->retired.empty()) return;
    if (!drained && hipStreamQuery(c->compute) != hipSuccess) { (void)hipGetLastError(); return; }   // steps that may use them are still queued
    for (mapn_ctx::Retired &r : c->retired) {
        if (r.graph) (void)hipGraphExecDestroy(r.graph);
        if (r.dev) (void)hipFree(r.dev);
        if (r.host) (void)hipHostFree(r.host);
    }
    c->retired.clear();
}

// Who does what in a PARTIALLY ACTIVE step of a SHARDED job (gather algorithms 4 / 5): the bodies [0, active) of the whole job advance.
// They form a ring of nba = ceil(active / 1024) blocks of their own; rank `rank` runs the meetings of ITS blocks in that ring (nbl of
// them from block a0: the blocks of its slice that hold active bodies -- none on a rank whose slice is frozen) and integrates its `ac`
// active bodies; the FROZEN bodies it owns ([fz_first, fz_first + fz_count)) still exert force, and their OWNER computes it: one
// one-sided launch over active x (its frozen bodies), whose sums travel to the active bodies' owners in the same rows as the reactions.
// So a rank whose whole slice is frozen is not idle: at N / 2 of 65 536 / 8 the four active ranks run 8 blocks of a 32-block ring
// (138e6 pair evaluations each), the four frozen ranks 32 768 x 8192 one-sided pairs each -- side by side.
ShardSplit shard_split_describe(uint32_t n, uint32_t world, uint32_t rank, uint32_t active)
{
    ShardSplit r{};
    const uint32_t count = n / world, cblk = count / mapn::SYM_BLOCK, first = count * rank;
    auto ac_of = [&](uint32_t q) { return shard_active_count(count * q, count, active); };
    r.nba = (active + mapn::SYM_BLOCK - 1u) / mapn::SYM_BLOCK;
    r.ac = ac_of(rank);
    r.nbl = (r.ac + mapn::SYM_BLOCK - 1u) / mapn::SYM_BLOCK;
    r.a0 = rank * cblk;
    r.fz_first = std::max(first, std::min(active, first + count));
    r.fz_count = first + count - r.fz_first;
    // the ring of the ACTIVE blocks (sym_shard_masks over nba blocks, cblk of them per rank) ...
    const uint32_t D = (r.nba - 1u) / 2u, half = (r.nba & 1u) ? 0u : r.nba / 2u;
    for (uint32_t a = 0; a < r.nba && cblk; a++)
        for (uint32_t d = 1; d <= D + (half ? 1u : 0u); d++) {
            if (d > D && !(d == half && mapn::sym_runs_half(a, half))) continue;
            const uint32_t b = (a + d) % r.nba, ra = a / cblk, rb = b / cblk;
            if (ra == rank) r.send_mask |= 1u << rb;
            if (rb == rank) r.recv_mask |= 1u << ra;
        }
    // ... and the frozen bodies' forces: from every rank that owns frozen bodies to every rank that owns active ones
    for (uint32_t q = 0; q < world; q++) {
        if (r.fz_count && ac_of(q)) r.send_mask |= 1u << q;
        if (r.ac && ac_of(q) < count) r.recv_mask |= 1u << q;
    }
    return r;
}

// Plan and scratch for `active` bodies: a LOOKUP among the cached plans (the last kActPlans counts), else a host-side plan into the
// least recently used slot, uploaded STREAM-ORDERED into the slot's own table buffer.  Never waits for the device (Compute.cpp:1009-1055
// only enqueues; the compute stream may be parked behind the consumer's fence, and the caller may be the thread that signals it): rows
// that have to grow are allocated anew and the old ones retired, nothing is freed or synchronised here.
// Unsharded (the split form of enqueue_sym_split): failure is not an error -- act_cur stays -1, act_failed remembers the count, and
// the caller takes another form.  SHARDED (enqueue_sym_shard_split): this rank's part of the job-wide step -- the plan of its blocks in
// the active ring, the one-sided launch over its frozen bodies, the masks; there a failure IS an error: the ranks chose the form
// together (a pure function of N, P and the count) and one of them falling back alone would leave the others waiting.
int prepare_sym_active(mapn_ctx *c, uint32_t active)
{
    const bool sharded = c->sym_sharded;
    if (act_ready(c, active)) { c->act_plans[c->act_cur].used = ++c->act_clock; return MAPN_OK; }
    c->act_cur = -1;
    ShardSplit role{};
    if (sharded) role = shard_split_describe(c->n, (uint32_t)c->cfg.world_size, (uint32_t)c->cfg.rank, active);
    else { role.nba = (active + mapn::SYM_BLOCK - 1) / mapn::SYM_BLOCK; role.nbl = role.nba; role.ac = active; role.fz_first = active; role.fz_count = c->n - active; }
    // the one-sided launch over the frozen bodies (sharded: the ones THIS rank owns): the default plan of an active x frozen launch, partial rows instead of the integrator
    mapn::ForcePlan frozen{};
    if (role.fz_count) {
        frozen = choose_plan(c, active, role.fz_count, 1, false);
        // (a rank's frozen bodies are FEW -- 8192 at 65 536 / 8: 128 tiles -- and the default plan would give every wave ONE tile and the step
        //  sixteen rows to add up: at least two tiles per wave, half the rows -- 67.4 against 69.8 and 80.7 against 84.0 us per step of a frozen
        //  rank at N / 2 and 5 N / 8 active, tools/shard_frozen_plan_sweep.py; an unsharded step's frozen range is mostly long enough not to be touched)
        while (frozen.sb >= 2u && (uint64_t)frozen.sb * frozen.waves * 128u > role.fz_count) frozen.sb /= 2u;
        env_plan("MAPN_FROZEN_PLAN", frozen);              // (hook: "k,waves,sb" -- the sweep behind the default)
        frozen.epi = mapn::EPI_ROWS;
    }
    int slot = -1;
    int slots = mapn_ctx::kActPlans;
    if (const char *hk = test_hook("MAPN_ACT_PLANS")) slots = std::max(1, std::min(slots, atoi(hk)));   // (A/B: 1 = round 5's one remembered count)
    // The symmetric plan is a function of the BLOCKS, not of the count: every count with the same ring (ceil(active / 1024) blocks; sharded:
    // the same blocks of it on this rank) runs the cached plan -- only the count itself (the kernel's bound for the far-away stand-ins), the
    // frozen launch and the roles are per step.  A slider dragged through 64-body steps re-plans once per 1024 bodies, not per frame.
    for (int k = 0; k < slots; k++) {
        mapn_ctx::SymActive &h = c->act_plans[k];
        if (h.active == 0 || h.role.nba != role.nba || h.role.nbl != role.nbl || h.role.a0 != role.a0) continue;
        if (role.fz_count && ensure_partial(c, frozen.sb, ((size_t)active + 63u) & ~(size_t)63u) != MAPN_OK) break;   // (cannot be had: the slow path below reports it)
        h.active = active; h.role = role; h.frozen = frozen; h.used = ++c->act_clock;
        c->act_cur = k;
        return MAPN_OK;
    }
    for (int k = 0; k < slots; k++) {
        if (c->act_plans[k].active == 0) { slot = k; break; }
        if (slot < 0 || c->act_plans[k].used < c->act_plans[slot].used) slot = k;
    }
    auto give_up = [&](const std::string &why) {
        if (sharded) return fail(MAPN_ERR_HIP, "sharded partially active step (%u of %u bodies): %s", active, c->n, why.c_str());
        if (std::find(c->act_failed.begin(), c->act_failed.end(), active) == c->act_failed.end()) {
            if (c->act_failed.size() >= 16) c->act_failed.erase(c->act_failed.begin());
            c->act_failed.push_back(active);
        }
        g_last_error = why;
        return (int)MAPN_OK;
    };
    const uint32_t nb = role.nba, nbl = role.nbl;
    const uint32_t gsym = (nb - 1u) / 2u + ((nb & 1u) ? 0u : 1u);
    const char *e = getenv("MAPN_SYM_MAX_MB");
    const uint64_t cap = (e ? strtoull(e, nullptr, 10) : 1024ull) << 20;
    uint32_t gpw = 0;                                      // (sharded: one window -- the rows are [J-block][this rank's blocks])
    if (!sharded) {
        const uint64_t per_group = (uint64_t)nb * mapn::SYM_BLOCK * sizeof(mapn::SymRow), fit = std::max<uint64_t>(1, cap / per_group);
        if (fit < gsym) gpw = (uint32_t)fit;
    }
    std::string err;
    mapn::SymPlanHost pl;
    if (nbl) {                                             // (a rank whose slice is frozen runs no meetings: no plan)
        bool built = false;
        for (const Shape &sh : candidate_shapes(c, sharded, nb, nbl, gsym, gpw, false))
            if ((built = mapn::build_sym_plan(nb, gpw, sh.parts, sh.t1, sh.t2, sh.waves, sh.hi, sh.lo, c->sym_xcd_weighted ? c->sym_xcd_w : nullptr, nbl, role.a0, 0u, pl, err))) break;
        if (!built) return give_up("partially active step: " + err + (sharded ? "" : "; another form runs"));
    }
    c->split_plans_built++;
    bool moved = false;                                    // a buffer captured graphs hold the address of was replaced
    auto grow = [&](void **p, size_t &have, size_t need) -> hipError_t {
        if (need <= have) return hipSuccess;
        void *fresh = nullptr;
        const hipError_t e2 = hipMalloc(&fresh, need);
        if (e2 != hipSuccess) return e2;
        retire(c, *p, nullptr, nullptr);                   // (steps still queued read the old one)
        *p = fresh; have = need; moved = true;
        return hipSuccess;
    };
    mapn_ctx::SymActiveRows &r = c->act_rows;
    mapn_ctx::SymActive &s = c->act_plans[slot];
    const size_t ab = (size_t)nbl * pl.parts * mapn::SYM_BLOCK * sizeof(mapn::SymRow);
    const size_t bb = sharded ? (size_t)nb * mapn::SYM_BLOCK * nbl * sizeof(mapn::SymRow) : (size_t)nb * mapn::SYM_BLOCK * pl.brows * sizeof(mapn::SymRow);
    const size_t hb = (size_t)nbl * pl.parts * 64 * sizeof(mapn::SymRow);
    const size_t cb = pl.windows.size() > 1 ? (size_t)nb * mapn::SYM_BLOCK * sizeof(float4) : 0;
    const size_t tb = pl.tables.size() * sizeof(uint32_t);
    hipError_t he = test_hook("MAPN_SYM_FAIL_ALLOC") ? hipErrorOutOfMemory : hipSuccess;
    if (he == hipSuccess) he = grow(reinterpret_cast<void **>(&r.arow), r.cap_arow, ab);
    if (he == hipSuccess) he = grow(reinterpret_cast<void **>(&r.brow), r.cap_brow, bb);
    if (he == hipSuccess) he = grow(reinterpret_cast<void **>(&r.brow1), r.cap_brow1, hb);
    if (he == hipSuccess && cb) he = grow(reinterpret_cast<void **>(&r.acc), r.cap_acc, cb);
    s.active = 0;                                          // (from here on the slot's old plan is gone)
    if (he == hipSuccess && tb) he = grow(reinterpret_cast<void **>(&s.tab), s.cap_tab, tb);
    if (he == hipSuccess && tb && !s.uploaded) he = hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming);
    if (he == hipSuccess && tb && (tb > s.cap_stage || (s.stage && hipEventQuery(s.uploaded) != hipSuccess))) {
        // the slot's pinned copy is too small, or an upload out of it is still queued (a parked stream): a fresh one, the old one retired
        (void)hipGetLastError();
        void *fresh = nullptr;
        he = hipHostMalloc(&fresh, std::max(tb, s.cap_stage), hipHostMallocDefault);
        if (he == hipSuccess) { retire(c, nullptr, s.stage, nullptr); s.stage = static_cast<uint32_t *>(fresh); s.cap_stage = std::max(tb, s.cap_stage); }
    }
    if (he == hipSuccess && tb) {
        memcpy(s.stage, pl.tables.data(), tb);
        he = hipMemcpyAsync(s.tab, s.stage, tb, hipMemcpyHostToDevice, c->compute);   // behind the steps that still read the slot's old tables
        if (he == hipSuccess) he = hipEventRecord(s.uploaded, c->compute);
    }
    if (he == hipSuccess && role.fz_count && ensure_partial(c, frozen.sb, ((size_t)active + 63u) & ~(size_t)63u) != MAPN_OK) he = hipErrorOutOfMemory;
    if (moved) drop_graphs(c);
    if (he != hipSuccess) {
        (void)hipGetLastError();
        char msg[256];
        snprintf(msg, sizeof msg, "partially active step: %.1f MiB of scratch for %u active bodies could not be allocated (%s)%s",
                 (double)(ab + bb + hb + cb + tb) / 1048576.0, active, hipGetErrorString(he), sharded ? "" : "; another form runs");
        return give_up(msg);
    }
    s.plan = std::move(pl);
    s.frozen = frozen;
    s.role = role;
    s.active = active;
    s.used = ++c->act_clock;
    s.generation = ++c->act_generation;
    c->act_cur = slot;
    return MAPN_OK;
}

int enqueue_sym_split(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer)
{
    const mapn_ctx::SymActive &s = c->act_plans[c->act_cur];   // (enqueue_step has made sure of it: act_ready)
    const mapn_ctx::SymActiveRows &rows = c->act_rows;
    const mapn::SymPlanHost &pl = s.plan;
    const uint32_t A = s.active;
    // (1) what the frozen bodies [A, N) do to the active ones: partial rows, one per block row of the launch
    mapn::StepArgs f = base;
    f.i_first = 0; f.i_count = A;
    fill_segment(f, 0, A, c->n - A, 0, s.frozen.sb * s.frozen.waves);
    f.partial_stride = (A + 63u) & ~63u;
    if (int rc = ensure_partial(c, s.frozen.sb, f.partial_stride)) return rc;     // (sized by prepare_sym_active: a no-op here)
    f.partial = c->partial; f.ticket = c->ticket; f.ticket_total = s.frozen.sb;
    HIP_TRY(mapn::launch_force(s.frozen, f, c->compute));
    // (2) the active bodies among themselves, window by window; the first reduce launch takes the frozen rows in
    const size_t nwin = pl.windows.size();
    for (size_t k = 0; k < nwin; k++) {
        mapn::SymArgs a = sym_args_of(pl, rows.arow, rows.brow, rows.brow1, s.tab, A, base, k);
        a.acc_in = k ? rows.acc : nullptr;
        a.acc_out = k + 1 < nwin ? rows.acc : nullptr;
        if (k == 0) { a.extra = c->partial; a.extra_rows = s.frozen.sb; a.extra_stride = f.partial_stride; }
        HIP_TRY(mapn::launch_force_sym(a, pl.waves, c->compute));
        if (timer && nwin == 1) { HIP_TRY(hipEventRecord(timer->force_done, c->compute)); timer->has_force = true; }   // (the frozen launch and the symmetric one)
        HIP_TRY(mapn::launch_sym_reduce(a, c->compute));
    }
    mapn::ForcePlan p{};
    p.kind = mapn::KERNEL_SYM; p.k = 2 * mapn::SYM_K2; p.waves = pl.waves; p.sb = pl.parts; p.nseg = 1; p.epi = mapn::EPI_ROWS;
    c->last_plan = p; c->last_i_count = A; c->last_launches = 2 * (uint32_t)nwin + 1u;
    c->last_split_active = A;
    return MAPN_OK;
}

// A PARTIALLY ACTIVE step of a sharded job in its split form (enqueue_sym_shard_split): a pure function of the context's shape and the
// count -- every rank of the job takes the same decision (a rank falling back alone would leave the others waiting for its rows).
bool sym_shard_split_eligible(const mapn_ctx *c, uint32_t active)
{
    if (!c->sym_ready || !c->sym_sharded || c->plan_forced || !c->p2p_ready || (c->gather_algo != 4 && c->gather_algo != 5)) return false;
    if (active >= c->n || active < 2u * mapn::SYM_BLOCK) return false;       // (all bodies: the sharded symmetric step proper; a handful: the one-sided step is as good)
    if (test_hook("MAPN_SYM_SHARD_CHUNK_FLAGS")) return false;               // (the flag forms of the reaction exchange, kept for the A/B, do not carry it)
    if (const char *f = test_hook("MAPN_SHARD_PARTIAL_FORM")) if (f[0] == 'o') return false;   // A/B: "one" = the one-sided step + pull of rounds 1 - 5
    return true;
}

// The PARTIALLY ACTIVE step of a sharded job (num_active < N on P ranks: Particles.cpp:391-394's slider, Compute.cpp:1041; VERDICT r5 #3 --
// until round 5 such a step ran the one-sided kernel over (this rank's active bodies) x N and pulled: 116 instead of 90 us at 65 536 / 8).
// Roles: shard_split_describe.  On the compute stream, in order:
//   (1) a rank that owns FROZEN bodies: one one-sided launch over active x (its frozen bodies), partial rows (EPI_ROWS) indexed by
//       the active body's number in the whole job;
//   (2) a rank that owns ACTIVE bodies: force_sym_kernel over its blocks of the ACTIVE ring (nb = nba, n = active: what lies past it
//       are the kernel's far-away stand-ins);
//   (3) every rank: the exchange launch -- per destination body the frozen rows (ascending, from zero), then the reactions of its
//       blocks, ONE row into the owner's receive region; its own active bodies integrated from its a-rows + the rows received; the new
//       positions of the ACTIVE bodies pushed / pulled, the counters advanced (by every rank: also one that moved nothing).
// The frozen bodies stay as they are in every replica's written buffer, bit for bit, like in the unsharded step.
int enqueue_sym_shard_split(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer)
{
    const uint32_t world = (uint32_t)c->cfg.world_size, rank = (uint32_t)c->cfg.rank;
    const mapn_ctx::SymActive &s = c->act_plans[c->act_cur];
    const mapn_ctx::SymActiveRows &rows = c->act_rows;
    const ShardSplit &ro = s.role;
    const uint32_t A = s.active;
    const bool push = c->gather_algo == 5;
    bool wait_in_launch = false;
    if (push) {
        // the replica these launches read was completed by the peers' pushes of the previous step.  A rank without frozen bodies lets its
        // symmetric launch wait for the counters itself, like enqueue_sym_shard; the one-sided launch cannot: one stream operation in front
        if (ro.fz_count || !ro.nbl || (c->p2p_shared_device && !c->p2p_loopback)) { if (int rc = settle_push(c)) return rc; }
        else wait_in_launch = true;
    }
    mapn::StepArgs f = base;
    if (ro.fz_count) {
        f.i_first = 0; f.i_count = A;
        fill_segment(f, 0, ro.fz_first, ro.fz_count, 0, s.frozen.sb * s.frozen.waves);
        f.partial_stride = (A + 63u) & ~63u;
        if (int rc = ensure_partial(c, s.frozen.sb, f.partial_stride)) return rc;     // (sized by prepare_sym_active: a no-op here)
        f.partial = c->partial; f.ticket = c->ticket; f.ticket_total = s.frozen.sb;
        HIP_TRY(mapn::launch_force(s.frozen, f, c->compute));
    }
    mapn::SymArgs a{};
    if (ro.nbl) {
        a = sym_args_of(s.plan, rows.arow, rows.brow, rows.brow1, s.tab, A, base, 0);
        a.shard_nbl = ro.nbl; a.a0 = ro.a0;
        if (wait_in_launch) {
            a.wait_counters = c->p2p_flags + mapn::SYM_POS_BASE; a.wait_status = c->async_status; a.wait_timeout_ticks = c->p2p_timeout_ticks;
            a.wait_need = c->sym_pos_epoch * mapn::SYM_COUNT_PER_LAUNCH; a.wait_world = world; a.wait_rank = rank; a.wait_self = c->p2p_loopback ? 1u : 0u;
            a.wait_dead = c->p2p_flags + mapn::SYM_DEAD_WORD;
            if (c->push_pending && sym_push_check()) {
                a.verify_sums = c->p2p_flags + mapn::sym_region_pos_sums_word(world, c->count); a.verify_epoch = c->sym_pos_epoch; a.verify_count = c->count; a.verify_active = c->push_active;
            }
            c->push_pending = false;
        }
        if (int rc = stamps_prepare(c, (size_t)ro.nbl * s.plan.nwaves, a)) return rc;
        HIP_TRY(mapn::launch_force_sym(a, s.plan.waves, c->compute));
    }
    if (timer) { HIP_TRY(hipEventRecord(timer->force_done, c->compute)); timer->has_force = true; }

    mapn::SymShardArgs h{};
    h.pos_old = base.pos_old; h.vel_old = base.vel_old; h.pos_new = base.pos_new; h.vel_new = base.vel_new;
    h.arow = rows.arow; h.brow = rows.brow; h.brow1 = rows.brow1; h.tab = s.tab;
    for (uint32_t q = 0; q < world; q++) {
        h.flags_peer[q] = c->p2p_peer_flags[q];
        h.recv_peer[q] = reinterpret_cast<float4 *>(reinterpret_cast<char *>(c->p2p_peer_flags[q]) + mapn::SYM_RECV_OFFSET);
        h.pos_peer[q] = reinterpret_cast<float4 *>(static_cast<char *>(c->p2p_peer_heap[q]) + (size_t)c->buffer_index * c->aligned_data_size);
    }
    h.push = push ? 1u : 0u;
    h.send_row = rank;
    h.flags_mine = c->p2p_flags;
    h.recv_mine = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(c->p2p_flags) + mapn::SYM_RECV_OFFSET);
    h.ticket = c->sym_shard_ticket;
    h.poll_rows = 1u;                                      // (self-validating rows: the only form that carries this step, sym_shard_split_eligible)
    h.pos_sums = push && sym_push_check() ? (uint32_t)mapn::sym_region_pos_sums_word(world, c->count) : 0u;
    h.status = c->async_status;
    h.rank = rank; h.world = world; h.count = c->count;
    h.active = A; h.count_active = ro.ac;
    if (ro.fz_count) { h.extra = c->partial; h.extra_rows = s.frozen.sb; h.extra_stride = f.partial_stride; }
    h.nb = ro.nba; h.nbl = ro.nbl; h.a0 = ro.a0; h.half_d = (ro.nba & 1u) ? 0u : ro.nba / 2u;
    h.parts = s.plan.parts; h.nwaves = s.plan.nwaves; h.max_meetings = s.plan.max_meetings; h.sets = ro.nbl ? s.plan.sets : 2u;
    h.send_mask = ro.send_mask; h.recv_mask = ro.recv_mask;
    if (c->p2p_loopback) {                                 // (timing on one GPU: every peer is this rank -- only its own row is waited for)
        const char *loop = test_hook("MAPN_P2P_LOOPBACK");
        h.recv_mask = ro.recv_mask & (1u << rank);
        if (loop && loop[0] == '2') h.send_mask &= 1u << rank;
    }
    h.step = ++c->sym_shard_step;
    h.pos_step = ++c->sym_pos_epoch;
    c->step_pulled = true;
    // a rank that owns frozen bodies starts its next partially active step with the one-sided launch, which cannot wait for the peers'
    // pushes itself: this exchange launch waits for them (and checks them) at its tail -- nothing is pending behind it
    h.wait_tail = push && (ro.fz_count || !ro.nbl) ? 1u : 0u;
    c->push_pending = push && !h.wait_tail;
    c->push_active = A;
    h.pull_self = c->p2p_loopback ? 1u : 0u;
    h.timeout_ticks = c->p2p_timeout_ticks;
    h.mass = base.mass; h.dt = base.dt; h.damping = base.damping;
    HIP_TRY(mapn::launch_sym_shard_exchange(h, std::min(c->sym_exchange_cap, 4096u), c->compute));
    mapn::ForcePlan p{};
    p.kind = mapn::KERNEL_SYM; p.k = 2 * mapn::SYM_K2; p.waves = ro.nbl ? s.plan.waves : 4u; p.sb = ro.nbl ? s.plan.parts : 1u; p.nseg = 1; p.epi = mapn::EPI_ROWS;
    c->last_plan = p; c->last_i_count = ro.nbl * mapn::SYM_BLOCK; c->last_launches = 2;
    c->last_split_active = A;
    return MAPN_OK;
}

}  // namespace host
}  // namespace mapn

extern "C" {

int mapn_step_form_describe(uint32_t num_particles, int32_t num_active)
{
    if (num_particles == 0) return fail(MAPN_ERR_INVALID_ARGUMENT, "step_form_describe: num_particles must be > 0");
    const uint32_t active = active_bodies(num_active, num_particles);
    if (active == 0) return (int)FORM_ONE_SIDED;                 // (nothing advances: the step only flips)
    if (num_particles < mapn::SYM_BLOCK) return (int)FORM_ONE_SIDED;      // (less than one block: the symmetric kernel does not apply)
    return (int)sym_form_by_cost(num_particles, active);
}

int mapn_get_split_plan(mapn_ctx *c, mapn_split_info *split, mapn_sym_plan_info *info, uint32_t *windows, uint64_t windows_capacity, uint32_t *tables, uint64_t tables_capacity)
{
    if (!c || !split || !info) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    memset(info, 0, sizeof *info);
    memset(split, 0, sizeof *split);
    if (c->act_cur < 0 || !c->act_plans[c->act_cur].active) {
        snprintf(info->error, sizeof info->error, "no partially active step has run in its split form yet");
        return fail(MAPN_ERR_STATE, "get_split_plan: %s", info->error);
    }
    const mapn_ctx::SymActive &s = c->act_plans[c->act_cur];
    split->active = s.active; split->frozen = s.role.fz_count;      // (sharded: the frozen bodies THIS rank owns, the j-range of its one-sided launch)
    split->frozen_first = s.role.fz_first; split->has_plan = s.role.nbl ? 1u : 0u;
    info->a0 = c->sym_sharded ? s.role.a0 : 0u; info->nbl = c->sym_sharded ? s.role.nbl : 0u;
    split->frozen_kernel = s.frozen.kind == mapn::KERNEL_LDS ? MAPN_KERNEL_LDS : MAPN_KERNEL_SCALAR;
    split->frozen_bodies_per_lane = s.frozen.k; split->frozen_waves = s.frozen.waves; split->frozen_sb = s.frozen.sb;
    info->scratch_bytes = c->act_rows.cap_arow + c->act_rows.cap_brow + c->act_rows.cap_brow1 + c->act_rows.cap_acc + s.cap_tab;
    if (!s.role.nbl) return MAPN_OK;                       // (a rank whose slice is frozen runs no meetings: no plan, windows = 0)
    return export_plan(s.plan, "get_split_plan", info, windows, windows_capacity, tables, tables_capacity);
}

}  // extern "C"
