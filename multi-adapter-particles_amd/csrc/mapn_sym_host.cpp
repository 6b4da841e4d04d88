// mapn_sym_host.cpp -- host side of the SYMMETRIC kernel (mapn_sym.hip): launch plan and scratch (prepare_sym), the unsharded step,
// the step sharded over ranks through mapped peer memory (gather algorithms 4 / 5) and over RCCL alone (6), the checks of pushed
// positions, XCD calibration (also at creation: MAPN_FLAG_XCD_CALIBRATE) and the plan's tuning / introspection entry points.
#include "mapn_internal.h"

using namespace mapn::host;

namespace mapn {
namespace host {

// set around the creation of calibrate_for_shard's bounded stand-in context: take its reading without the creation-time A/B
static thread_local bool g_calibration_without_ab = false;

void release_sym(mapn_ctx *c)
{
    if (c->sym_arow) (void)hipFree(c->sym_arow);
    if (c->sym_brow) (void)hipFree(c->sym_brow);
    if (c->sym_brow1) (void)hipFree(c->sym_brow1);
    if (c->sym_acc) (void)hipFree(c->sym_acc);
    if (c->sym_tab) (void)hipFree(c->sym_tab);
    if (c->sym_send) (void)hipFree(c->sym_send);
    if (c->sym_recv) (void)hipFree(c->sym_recv);
    c->sym_send = c->sym_recv = nullptr;
    c->sym_arow = c->sym_brow = c->sym_brow1 = nullptr; c->sym_acc = nullptr;
    c->sym_tab = nullptr;
    c->sym_scratch_bytes = 0;
    c->sym_ready = false;
}

// which ranks this rank produces reactions for / receives reactions from: the meeting schedule of
// force_sym_kernel (I-block a meets a+1 .. a+D, and a+nb/2 when nb is even and a is that pair's runner: sym_runs_half), block -> owner
void sym_shard_masks(uint32_t nb, uint32_t world, uint32_t rank, uint32_t &send, uint32_t &recv)
{
    const uint32_t nbl = nb / world, D = (nb - 1u) / 2u, half = (nb & 1u) ? 0u : nb / 2u;
    send = recv = 0;
    for (uint32_t a = 0; a < nb; a++) {
        for (uint32_t d = 1; d <= D + (half ? 1u : 0u); d++) {
            if (d > D && !(d == half && mapn::sym_runs_half(a, half))) continue;
            const uint32_t b = (a + d) % nb, ra = a / nbl, rb = b / nbl;
            if (ra == rank) send |= 1u << rb;
            if (rb == rank) recv |= 1u << ra;
        }
    }
}

// Does the symmetric kernel apply to this context at all (independent of a step's num_active)?  Unsharded: the
// whole-N all-pairs step with N >= 1024 (the last block is padded inside the kernel).  Sharded (gather algorithm
// 4): every rank's slice is whole 1024-body blocks.
bool sym_applies(const mapn_ctx *c, bool sharded)
{
    if (c->cfg.kernel != MAPN_KERNEL_SYMMETRIC && c->cfg.kernel != MAPN_KERNEL_AUTO) return false;
    if (c->cfg.force_mode != MAPN_FORCE_ALL_PAIRS) return false;
    const char *off = test_hook("MAPN_NO_SYM");                       // (a hook: MAPN_TEST_HOOKS=1 as well)
    if (off && off[0] == '1' && c->cfg.kernel == MAPN_KERNEL_AUTO) return false;
    if (sharded) return c->cfg.world_size >= 2 && c->count % mapn::SYM_BLOCK == 0 && c->count * (uint32_t)c->cfg.world_size == c->n;
    return c->cfg.world_size == 1 && c->n >= mapn::SYM_BLOCK;   // (a smaller job does not fill one block: one-sided)
}


// The launch shapes prepare_sym tries, in order, for a job of nb blocks (nbl of them in one launch; gsym symmetric groups; gpw: groups
// per window, may be narrowed by a hook / the user's plan).  tunable: honour mapn_set_sym_plan and the MAPN_SYM_* hooks (the
// context's own plan); false: the default shapes only (the plan of the ACTIVE bodies of a partially active step).
std::vector<Shape> candidate_shapes(const mapn_ctx *c, bool sharded, uint32_t nb, uint32_t nbl, uint32_t gsym, uint32_t &gpw, bool tunable)
{
    auto hook = [&](const char *name) -> const char * { return tunable ? test_hook(name) : nullptr; };
    const bool user_plan = tunable && c->sym_user_plan;
    // shape: 4-wave workgroups (2 waves per SIMD are resident: 248 VGPRs).  Unsharded: about 8192 workgroups per launch
    // but at most 32 per I-block (65 536 bodies: parts 24 / 32 / 48 / 64 -> 0.663 / 0.647 / 0.669 / 0.675 ms; 262 144: 8 / 32 /
    // 64 / 128 -> 9.95 / 9.60 / 9.60 / 9.89 ms; 1 048 576: 2 / 8 / 32 / 64 equal within 1 %); few rounds of workgroups
    // (65 536 .. 131 072 bodies, one window): parts that TAPER 4 : 2 : 1 so that the workgroups dispatched last are a
    // quarter of the first ones' size (+1.1 % at 65 536, +3.6 % at 100 000).  Sharded: one resident round -- about 512
    // workgroups, at least 32 per block.
    // A launch of ONE resident round (sharded, up to 256 workgroups of 8 waves): 8-wave workgroups whose first four waves -- the
    // older wave of every SIMD, which the SIMD issues first -- carry 3 (2) times the steps of the last four, so that the two
    // waves of a SIMD finish together (build_sym_plan; rank 0 of 65 536 / 8: force launch 83.7 against 87.9 us).
    uint32_t waves = 4, parts = sharded ? std::max(32u, (512u + nbl - 1u) / nbl) : std::min(32u, std::max(1u, (8192u + nb - 1u) / nb));
    std::vector<Shape> tries;
    {
        unsigned ew = 0, ep = 0, tp = 0, t1 = 0, t2 = 0, eg = 0, bh = 1, bl = 1;
        const char *pl = hook(sharded ? "MAPN_SYM_SHARD_PLAN" : "MAPN_SYM_PLAN");     // "waves,parts" tuning override
        if (pl && sscanf(pl, "%u,%u", &ew, &ep) == 2 && (ew == 4 || ew == 8) && ep >= 1) { waves = ew; parts = ep; }
        const char *wb = hook(sharded ? "MAPN_SYM_SHARD_WAVE_BIAS" : "MAPN_SYM_WAVE_BIAS");   // "hi,lo": first half : second half of a workgroup's waves
        const bool bias_env = wb && sscanf(wb, "%u,%u", &bh, &bl) == 2 && bh >= 1 && bl >= 1;
        if (!bias_env) bh = bl = 1;
        const char *tw = hook("MAPN_SYM_WINDOW");                                    // groups per window (unsharded)
        if (tw && !sharded && sscanf(tw, "%u", &eg) == 1 && eg >= 1) gpw = eg >= gsym ? 0u : eg;
        const char *t = hook(sharded ? "MAPN_SYM_SHARD_TAPER" : "MAPN_SYM_TAPER");     // "parts,taper1,taper2"; "0" = equal parts
        if (user_plan) {
            waves = c->sym_user[0]; parts = c->sym_user[1];
            tries.push_back({parts, c->sym_user[2] + c->sym_user[3] ? c->sym_user[2] : parts, c->sym_user[3], waves, c->sym_user[5], c->sym_user[6]});
            if (!sharded && c->sym_user[4]) gpw = c->sym_user[4] >= gsym ? 0u : c->sym_user[4];
        } else if (t && sscanf(t, "%u,%u,%u", &tp, &t1, &t2) == 3 && tp >= 1 && t1 + t2 <= tp) {
            tries.push_back({tp, t1, t2, waves, bh, bl});
        } else if (!(t && t[0] == '0') && !pl && !bias_env && !sharded) {
            // biased 8-wave workgroups (one per compute unit) where the launch's workgroups fill whole rounds of the device
            const uint32_t cus = c->cus > 0 ? (uint32_t)c->cus : 256u;
            for (uint32_t q : {4u, 5u, 6u, 7u, 8u, 9u, 10u, 11u, 12u, 13u, 14u, 15u, 16u, 3u, 2u}) {
                const uint64_t wg = (uint64_t)nb * q, rounds = (wg + cus - 1u) / cus;
                // (the last round at least 97 % full: 69 632 bodies, 68 blocks x 11 = 748 of 768: +2 % over the tapered 4-wave shape;
                //  90 112 bodies, 88 x 14 = 1232 of 1280 = 96 %: -0.9 %)
                if (wg < cus || (rounds < 16u && wg * 100u < rounds * cus * 97u)) continue;
                tries.push_back({q, q, 0, 8, 10, 3}); tries.push_back({q, q, 0, 8, 3, 1});
                break;
            }
            if (gpw == 0 && nb >= 64u && nb <= 128u) {
                tries.push_back({40, 28, 4, 4, 1, 1}); tries.push_back({38, 28, 4, 4, 1, 1}); tries.push_back({36, 28, 4, 4, 1, 1});
                tries.push_back({36, 28, 8, 4, 1, 1});     // (XCD-weighted parts shrink the slow dies' shares: no part of one unit then)
            }
        } else if (sharded && !pl && !bias_env && !c->p2p_shared_device) {
            // (not when several ranks share this GPU: an 8-wave workgroup needs BOTH wave slots of all four SIMDs of a compute
            //  unit, and cannot be placed on one where a peer's exchange workgroup sits waiting -- for this very launch's rows)
            // the fewest parts per block that fill whole rounds of the device (as above; 196 608 bodies over 8 ranks, 24 blocks:
            // 21 parts = 504 of 512 workgroups: 703 us per step against 862 with 16 parts = one and a half rounds)
            const uint32_t cus = c->cus > 0 ? (uint32_t)c->cus : 256u;
            uint32_t p8 = std::max(16u, (cus + nbl - 1u) / nbl);
            for (uint32_t q = 4u; q <= 64u; q++) {
                const uint64_t wg = (uint64_t)nbl * q, rounds = (wg + cus - 1u) / cus;
                if (wg < cus || (rounds < 16u && wg * 100u < rounds * cus * 97u)) continue;
                p8 = q;
                break;
            }
            tries.push_back({p8, p8, 0, 8, 10, 3}); tries.push_back({p8, p8, 0, 8, 3, 1}); tries.push_back({p8, p8, 0, 8, 2, 1});
        }
        if (!user_plan)
            for (uint32_t q = parts; q >= 1u; q = q > 1u ? q / 2u : 0u) tries.push_back({q, q, 0, waves, bh, bl});   // equal parts, halved until every wave has 64 steps
    }
    return tries;
}

// Build the launch plan and allocate ALL of the symmetric step's scratch: a-rows [blocks][parts][1024], b-rows
// (unsharded: [N/64][groups of the widest window][64], bounded by MAPN_SYM_MAX_MB -- a step is made in as many windows
// of partner distance as that takes, so the scratch is O(N); sharded: [N/64][blocks of this rank][64]), head rows,
// the running sum between windows, the plan tables.  Returns MAPN_OK with sym_ready false (and the reason in sym_note)
// when the kernel does not apply or -- MAPN_KERNEL_AUTO only -- the memory is not to be had: the one-sided kernel
// then runs every step.  An explicit MAPN_KERNEL_SYMMETRIC / mapn_set_sym_plan that cannot be honoured is an error.
int prepare_sym(mapn_ctx *c, bool sharded)
{
    release_sym(c);
    c->sym_note.clear();
    forget_sym_active(c);                                  // (the split form's plans follow this one's weights: made again by the next partially active step)
    if (!sym_applies(c, sharded)) return MAPN_OK;
    const bool must = c->cfg.kernel == MAPN_KERNEL_SYMMETRIC || c->sym_user_plan;
    const uint32_t nb = (c->n + mapn::SYM_BLOCK - 1) / mapn::SYM_BLOCK, nbl = sharded ? c->count / mapn::SYM_BLOCK : nb;
    const uint32_t gsym = (nb - 1u) / 2u + ((nb & 1u) ? 0u : 1u);
    const char *e = getenv("MAPN_SYM_MAX_MB");
    const bool simulate_failure = test_hook("MAPN_SYM_FAIL_ALLOC") != nullptr;     // tests: behave as if hipMalloc had failed
    // unsharded: 1 GiB of b-rows by default (12 B per body and group: 1 048 576 bodies 7 windows, 4 194 304 bodies 97); sharded: one window, up to 16 GiB
    const uint64_t cap = (e ? strtoull(e, nullptr, 10) : (sharded ? 16384ull : 1024ull)) << 20;
    uint32_t gpw = 0;                                                           // symmetric groups per window (0: all in one)
    if (sharded) {
        if ((uint64_t)c->n * nbl * sizeof(mapn::SymRow) > cap) {
            c->sym_note = "symmetric kernel (sharded): reaction rows exceed MAPN_SYM_MAX_MB";
            return must ? fail(MAPN_ERR_INVALID_ARGUMENT, "%s", c->sym_note.c_str()) : MAPN_OK;
        }
    } else {
        const uint64_t per_group = (uint64_t)nb * mapn::SYM_BLOCK * sizeof(mapn::SymRow);
        const uint64_t fit = std::max<uint64_t>(1, cap / per_group);
        if (fit < gsym) gpw = (uint32_t)fit;
    }
    const std::vector<Shape> tries = candidate_shapes(c, sharded, nb, nbl, gsym, gpw, true);
    std::string err;
    bool built = false;
    // XCD weights: class-aware where it applies (heavy blocks on the fast dies), else spread; MAPN_SYM_XCD_MODE=spread: the A/B of the earlier form
    static const uint32_t xcd_mode = [] { const char *e = test_hook("MAPN_SYM_XCD_MODE"); return (e && e[0] == 's') ? 1u : 0u; }();
    for (const Shape &sh : tries)
        if ((built = mapn::build_sym_plan(nb, gpw, sh.parts, sh.t1, sh.t2, sh.waves, sh.hi, sh.lo, c->sym_xcd_weighted ? c->sym_xcd_w : nullptr, nbl,
                                          sharded ? (uint32_t)c->cfg.rank * nbl : 0u, xcd_mode, c->sym_plan, err))) break;
    if (!built) {
        c->sym_note = err;
        return must ? fail(MAPN_ERR_INVALID_ARGUMENT, "%s", err.c_str()) : MAPN_OK;
    }
    const mapn::SymPlanHost &pl = c->sym_plan;
    const size_t ab = (size_t)nbl * pl.parts * mapn::SYM_BLOCK * sizeof(mapn::SymRow);
    const size_t bb = sharded ? (size_t)c->n * nbl * sizeof(mapn::SymRow) : (size_t)nb * mapn::SYM_BLOCK * pl.brows * sizeof(mapn::SymRow);
    const size_t hb = (size_t)nbl * pl.parts * 64 * sizeof(mapn::SymRow);
    const size_t cb = pl.windows.size() > 1 ? (size_t)nb * mapn::SYM_BLOCK * sizeof(float4) : 0;
    const size_t tb = pl.tables.size() * sizeof(uint32_t);
    hipError_t he = simulate_failure ? hipErrorOutOfMemory : hipSuccess;
    if (he == hipSuccess) he = hipMalloc(&c->sym_arow, ab);
    if (he == hipSuccess) he = hipMalloc(&c->sym_brow, bb);
    if (he == hipSuccess) he = hipMalloc(&c->sym_brow1, hb);
    if (he == hipSuccess && cb) he = hipMalloc(&c->sym_acc, cb);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&c->sym_tab), tb);
    if (he == hipSuccess) he = hipMemcpy(c->sym_tab, pl.tables.data(), tb, hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        (void)hipGetLastError();
        release_sym(c);
        char msg[256];
        snprintf(msg, sizeof msg, "symmetric kernel: %.1f MiB of scratch could not be allocated (%s); the one-sided kernel runs instead",
                 (double)(ab + bb + hb + cb + tb) / 1048576.0, hipGetErrorString(he));
        c->sym_note = msg;
        if (must) return fail(MAPN_ERR_HIP, "%s", msg);
        g_last_error = msg;                                // a warning: the call that got here still succeeds
        return MAPN_OK;
    }
    c->sym_scratch_bytes = ab + bb + hb + cb + tb;
    c->sym_parts = pl.parts; c->sym_waves = pl.waves;
    c->sym_sharded = sharded;
    c->sym_ready = true;
    if (sharded) {
        // the exchange launch's workgroups must all be resident at once: size it for the compute units this process really gets
        // (HSA_CU_MASK / a partition leave fewer than the device properties say), not for the nominal count
        if (!c->cus_active) { c->cus_active = mapn::probe_active_compute_units(c->compute); if (c->cus_active <= 0 || c->cus_active > c->cus) c->cus_active = c->cus; }
        c->sym_exchange_cap = mapn::sym_shard_exchange_resident_workgroups(c->count, c->cus_active);
        // ranks that share this GPU run their exchange launches side by side (every process has its own hardware queue, nothing
        // time-slices them): together they must fit, or the device fills with workgroups waiting for peers whose own cannot start
        if (c->p2p_ranks_on_device > 1u) c->sym_exchange_cap = std::max(1u, c->sym_exchange_cap / c->p2p_ranks_on_device);
    }
    return MAPN_OK;
}

// MAPN_STAMP_DUMP=<file> (development tool): a stamped diagnostic launch of the symmetric kernel also records, per wave,
// its entry / loop start / loop end / exit times (100 MHz) and where it ran; mapn_measure_clock writes them to the file.
int timeline_prepare(mapn_ctx *c, size_t nw, mapn::SymArgs &a)
{
    if (!c->stamp_next || !(c->calibrating || test_hook("MAPN_STAMP_DUMP"))) return MAPN_OK;
    if (nw > c->timeline_waves) {
        if (c->timeline_buf) HIP_TRY(hipFree(c->timeline_buf));
        c->timeline_buf = nullptr; c->timeline_waves = 0;
        HIP_TRY(hipMalloc(&c->timeline_buf, nw * 48));
        c->timeline_waves = nw;
    }
    HIP_TRY(hipMemsetAsync(c->timeline_buf, 0, nw * 48, c->compute));
    a.timeline = c->timeline_buf;
    c->timeline_last = nw;
    return MAPN_OK;
}

// the stamp buffer of a diagnostic launch (mapn_measure_clock; never in an ordinary step)
int stamps_prepare(mapn_ctx *c, size_t nw, mapn::SymArgs &a)
{
    if (!c->stamp_next) return MAPN_OK;
    if (nw > c->stamp_waves) {
        if (c->stamp_buf) HIP_TRY(hipFree(c->stamp_buf));
        c->stamp_buf = nullptr; c->stamp_waves = 0;
        HIP_TRY(hipMalloc(&c->stamp_buf, nw * 16));
        c->stamp_waves = nw;
    }
    HIP_TRY(hipMemsetAsync(c->stamp_buf, 0, c->stamp_waves * 16, c->compute));
    a.stamps = c->stamp_buf;
    return timeline_prepare(c, nw, a);
}

mapn::SymArgs sym_args_of(const mapn::SymPlanHost &pl, mapn::SymRow *arow, mapn::SymRow *brow, mapn::SymRow *brow1, const uint32_t *tab, uint32_t n,
                          const mapn::StepArgs &base, size_t window)
{
    mapn::SymArgs a{};
    a.pos_old = base.pos_old; a.vel_old = base.vel_old; a.pos_new = base.pos_new; a.vel_new = base.vel_new;
    a.arow = arow; a.brow = brow; a.brow1 = brow1;
    a.tab = tab + window * pl.table_stride;
    a.wgmap = pl.wgmap_entries ? tab + pl.wgmap_offset : nullptr;
    a.la_flip = pl.la_flip;
    a.n = n; a.n_integrate = n; a.nb = pl.nb; a.parts = pl.parts; a.nwaves = pl.nwaves; a.max_meetings = pl.max_meetings; a.sets = pl.sets;
    a.g0 = pl.windows[window].g0; a.g1 = pl.windows[window].g1;
    a.brows = pl.brows; a.half_d = pl.half;
    a.mass = base.mass; a.soft2 = base.soft2; a.dt = base.dt; a.damping = base.damping;
    // rows leave the XCD as they are produced (write-through) instead of waiting in its L2 for the end-of-kernel write-back:
    // same box, rank 0 of 65 536 / 8: force launch 92.7 against 95.9 us; 65 536 unsharded 0.3 % faster (MAPN_SYM_ROW_WT=0: A/B)
    static const uint32_t wt = [] { const char *e = test_hook("MAPN_SYM_ROW_WT"); return e ? (uint32_t)atoi(e) : 1u; }();
    a.row_wt = wt;
    // the I-block reaches the workgroup's waves through LDS (a quarter of the global loads at launch start): same box, rank 0 of
    // 65 536 / 8: prologue 2.9 against 5.0 us, force launch 89.9 against 92.4 us; 65 536 unsharded 0.45 % faster (MAPN_SYM_STAGE=0: A/B)
    static const uint32_t stage = [] { const char *e = test_hook("MAPN_SYM_STAGE"); return e ? (uint32_t)atoi(e) : 1u; }();
    a.stage_iblock = stage;
    return a;
}

mapn::SymArgs sym_args(const mapn_ctx *c, const mapn::StepArgs &base, size_t window)
{
    return sym_args_of(c->sym_plan, c->sym_arow, c->sym_brow, c->sym_brow1, c->sym_tab, c->n, base, window);
}

// One step = one force launch + one reduce launch per window of partner distance; the reduce launches carry the
// running sum from window to window (in a fixed order: bit-reproducible), the last one integrates.
int enqueue_sym(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer)
{
    const mapn::SymPlanHost &pl = c->sym_plan;
    const size_t nwin = pl.windows.size();
    for (size_t k = 0; k < nwin; k++) {
        mapn::SymArgs a = sym_args(c, base, k);
        a.n_integrate = base.i_count;                      // (unsharded: the active bodies are [0, i_count))
        a.acc_in = k ? c->sym_acc : nullptr;
        a.acc_out = k + 1 < nwin ? c->sym_acc : nullptr;
        if (k == 0) { if (int rc = stamps_prepare(c, (size_t)a.nb * pl.nwaves, a)) return rc; }
        HIP_TRY(mapn::launch_force_sym(a, pl.waves, c->compute));
        if (timer && nwin == 1) { HIP_TRY(hipEventRecord(timer->force_done, c->compute)); timer->has_force = true; }
        HIP_TRY(mapn::launch_sym_reduce(a, c->compute));
    }
    mapn::ForcePlan p{};
    p.kind = mapn::KERNEL_SYM; p.k = 2 * mapn::SYM_K2; p.waves = pl.waves; p.sb = pl.parts; p.nseg = 1; p.epi = mapn::EPI_ROWS;
    c->last_plan = p; c->last_i_count = c->n; c->last_launches = 2 * (uint32_t)nwin;   // (the force launches always cover all N bodies)
    return MAPN_OK;
}

// Gather algorithm 4: the symmetric step sharded over ranks.
bool sym_shard_eligible(const mapn_ctx *c, uint32_t active)
{
    if (!c->sym_ready || !c->sym_sharded || c->plan_forced) return false;
    if (c->gather_algo == 6) return c->comm != nullptr && c->sym_send != nullptr && active == c->n;
    if (!c->p2p_ready || (c->gather_algo != 4 && c->gather_algo != 5)) return false;
    return active == c->n;
}

// Gather algorithm 5: the peers store their new slices into this rank's replica; whoever reads the replica next must first
// wait for their counters.  The sharded symmetric force launch does that itself; every other reader (a one-sided step, a
// download, wait_idle) gets this stream operation in front.
// are the positions of gather algorithm 5 checked against their pushers' checksums (default; MAPN_SYM_PUSH_CHECK=0: the A/B without)
bool sym_push_check()
{
    static const bool on = [] { const char *e = test_hook("MAPN_SYM_PUSH_CHECK"); return !(e && e[0] == '0'); }();   // (a hook: a stray variable cannot remove the only data check of algorithm 5)
    return on;
}

int settle_push(mapn_ctx *c)
{
    if (!c->push_pending) return MAPN_OK;
    c->push_pending = false;
    // (the latest step wrote buffer 1 - index: that is where the peers pushed; their checksums are verified as the force launch would)
    HIP_TRY(mapn::launch_p2p_wait(c->p2p_flags + mapn::SYM_POS_BASE, c->sym_pos_epoch * mapn::SYM_COUNT_PER_LAUNCH, (uint32_t)c->cfg.world_size, (uint32_t)c->cfg.rank, c->p2p_loopback ? 1u : 0u,
                                  c->p2p_timeout_ticks, c->async_status, c->p2p_flags + mapn::SYM_DEAD_WORD, c->pos[1 - c->buffer_index],
                                  sym_push_check() ? c->p2p_flags + mapn::sym_region_pos_sums_word((uint32_t)c->cfg.world_size, c->count) : nullptr, c->sym_pos_epoch, c->count,
                                  c->push_active, c->compute));
    return MAPN_OK;
}

// do the new positions travel inside the exchange launch (default) or in p2p_gather_kernel behind it (MAPN_SYM_SHARD_PULL=0: A/B)
bool sym_shard_pull_folded()
{
    static const bool folded = [] { const char *e = test_hook("MAPN_SYM_SHARD_PULL"); return !(e && e[0] == '0'); }();
    return folded;
}

int enqueue_sym_shard(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer)
{
    const uint32_t world = (uint32_t)c->cfg.world_size, rank = (uint32_t)c->cfg.rank;
    const mapn::SymPlanHost &pl = c->sym_plan;
    mapn::SymArgs a = sym_args(c, base, 0);
    a.shard_nbl = c->count / mapn::SYM_BLOCK;
    a.a0 = rank * a.shard_nbl;
    const bool push = c->gather_algo == 5;
    if (push && c->p2p_shared_device && !c->p2p_loopback) {
        // ranks SHARING this GPU (tests): a force launch that fills the device while it waits for the peers' counters keeps the
        // peers' own launches out -- eight such launches waited for each other until the timeouts.  One small stream operation
        // waits instead, in front of the launch.
        if (int rc = settle_push(c)) return rc;
    } else if (push) {
        // the replica this launch reads was completed by the peers' pushes of the previous step: wait for their counters in the launch
        a.wait_counters = c->p2p_flags + mapn::SYM_POS_BASE; a.wait_status = c->async_status; a.wait_timeout_ticks = c->p2p_timeout_ticks;
        a.wait_need = c->sym_pos_epoch * mapn::SYM_COUNT_PER_LAUNCH; a.wait_world = world; a.wait_rank = rank; a.wait_self = c->p2p_loopback ? 1u : 0u;
        a.wait_dead = c->p2p_flags + mapn::SYM_DEAD_WORD;
        if (c->push_pending && sym_push_check()) {         // pushes nobody has checked yet (not after an upload: that data is not the peers')
            a.verify_sums = c->p2p_flags + mapn::sym_region_pos_sums_word(world, c->count); a.verify_epoch = c->sym_pos_epoch; a.verify_count = c->count; a.verify_active = c->push_active;
        }
        c->push_pending = false;
    }
    if (int rc = stamps_prepare(c, (size_t)a.shard_nbl * pl.nwaves, a)) return rc;
    HIP_TRY(mapn::launch_force_sym(a, pl.waves, c->compute));
    if (timer) { HIP_TRY(hipEventRecord(timer->force_done, c->compute)); timer->has_force = true; }

    mapn::SymShardArgs h{};
    h.pos_old = a.pos_old; h.vel_old = a.vel_old; h.pos_new = a.pos_new; h.vel_new = a.vel_new;
    h.arow = a.arow; h.brow = a.brow; h.brow1 = a.brow1; h.tab = a.tab;
    const bool pull = push || sym_shard_pull_folded();
    for (uint32_t q = 0; q < world; q++) {
        h.flags_peer[q] = c->p2p_peer_flags[q];
        h.recv_peer[q] = reinterpret_cast<float4 *>(reinterpret_cast<char *>(c->p2p_peer_flags[q]) + mapn::SYM_RECV_OFFSET);
        // both position buffers live in one heap allocation: the written buffer sits buffer_index * aligned_data_size in
        h.pos_peer[q] = pull ? reinterpret_cast<float4 *>(static_cast<char *>(c->p2p_peer_heap[q]) + (size_t)c->buffer_index * c->aligned_data_size) : nullptr;
    }
    h.push = push ? 1u : 0u;
    h.send_row = rank;
    static const uint32_t rel = [] { const char *e = test_hook("MAPN_SYM_SHARD_RELEASE"); return e ? (uint32_t)atoi(e) : 0u; }();   // 1 = a release fence (L2 write-back) before each publication: +22 us per step measured, and the acknowledged write-through stores need none
    h.release = rel;
    h.flags_mine = c->p2p_flags;
    h.recv_mine = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(c->p2p_flags) + mapn::SYM_RECV_OFFSET);
    h.ticket = c->sym_shard_ticket;
    // arrival flags per (sender, 256-body chunk) behind the receive region -- with PUSHED positions (same box, rank 0 of 65 536 / 8:
    // 93.2 against 93.5 us per step); where the launch also PULLS the peers' positions the workgroups' spread-out ends delay
    // the position counters and the ticket form stays (95.1 against 96.6).  MAPN_SYM_SHARD_CHUNK_FLAGS=0 / 2: never / always (A/B)
    static const int chunk_mode = [] { const char *e = test_hook("MAPN_SYM_SHARD_CHUNK_FLAGS"); return e ? atoi(e) : -1; }();   // (-1: not set -- read ONCE: the form cannot flip between two steps)
    const bool chunked = chunk_mode == 2 || ((chunk_mode == 1 || chunk_mode == -1) && push);
    h.chunk_flags = chunked ? (uint32_t)mapn::sym_region_chunk_flags_word(world, c->count) : 0u;
    // round 4: no arrival flags at all -- the rows validate themselves (SymShardArgs::poll_rows); MAPN_SYM_SHARD_CHUNK_FLAGS = 0 / 1 / 2 (a
    // hook): the flag forms of round 3, for the A/B
    h.poll_rows = chunk_mode >= 0 ? 0u : 1u;
    h.pos_sums = push && sym_push_check() ? (uint32_t)mapn::sym_region_pos_sums_word(world, c->count) : 0u;
    h.status = c->async_status;
    h.rank = rank; h.world = world; h.count = c->count;
    h.active = c->n; h.count_active = c->count;            // (all bodies advance: the partially active step is enqueue_sym_shard_split)
    h.nb = a.nb; h.nbl = a.shard_nbl; h.a0 = a.a0; h.half_d = a.half_d; h.parts = pl.parts; h.nwaves = pl.nwaves; h.max_meetings = pl.max_meetings; h.sets = pl.sets;
    h.send_mask = c->sym_send_mask; h.recv_mask = c->sym_recv_mask;
    h.step = ++c->sym_shard_step;
    h.pos_step = pull ? ++c->sym_pos_epoch : 0u;
    { const char *cr = test_hook("MAPN_TEST_CORRUPT_ROW"); if (cr && (uint32_t)strtoul(cr, nullptr, 10) == h.step) h.corrupt_row = 1u; }
    if (push) {                                            // TEST HOOK: "<publication number>", or "once" = the next publication of this process, one time
        static bool corrupted_once = false;
        const char *cp = test_hook("MAPN_TEST_CORRUPT_PUSH");
        if (cp && strcmp(cp, "once") == 0) { if (!corrupted_once) { corrupted_once = true; h.corrupt_push = 1u; } }
        else if (cp && (uint32_t)strtoul(cp, nullptr, 10) == h.pos_step) h.corrupt_push = 1u;
    }
    c->step_pulled = pull;
    c->push_pending = push;
    c->push_active = c->n;
    h.pull_self = c->p2p_loopback ? 1u : 0u;
    h.timeout_ticks = c->p2p_timeout_ticks;
    h.mass = a.mass; h.dt = a.dt; h.damping = a.damping;
    if (a.timeline) {                                      // MAPN_STAMP_DUMP: the exchange launch's phase stamps behind the force launch's
        if (!c->xtimeline_buf) HIP_TRY(hipMalloc(&c->xtimeline_buf, 4096 * 64));
        HIP_TRY(hipMemsetAsync(c->xtimeline_buf, 0, 4096 * 64, c->compute));
        h.timeline = c->xtimeline_buf;
    }
    HIP_TRY(mapn::launch_sym_shard_exchange(h, std::min(c->sym_exchange_cap, 4096u), c->compute));
    mapn::ForcePlan p{};
    p.kind = mapn::KERNEL_SYM; p.k = 2 * mapn::SYM_K2; p.waves = pl.waves; p.sb = pl.parts; p.nseg = 1; p.epi = mapn::EPI_ROWS;
    c->last_plan = p; c->last_i_count = c->count; c->last_launches = 2;
    return MAPN_OK;
}

// Gather algorithm 6: the same sharded symmetric step with the reaction rows carried by RCCL instead of remote stores and
// counters -- pack launch, one group of ncclSend / ncclRecv into the same [sender][body] layout, reduce launch; the new positions
// then travel by ncclAllGather as in algorithm 0.  Four launches and two collectives per step: the form for nodes where peers'
// memory cannot be mapped, not the fast one.
int enqueue_sym_shard_rccl(mapn_ctx *c, const mapn::StepArgs &base, StepTimer *timer)
{
    const uint32_t world = (uint32_t)c->cfg.world_size, rank = (uint32_t)c->cfg.rank;
    const mapn::SymPlanHost &pl = c->sym_plan;
    mapn::SymArgs a = sym_args(c, base, 0);
    a.shard_nbl = c->count / mapn::SYM_BLOCK;
    a.a0 = rank * a.shard_nbl;
    if (int rc = stamps_prepare(c, (size_t)a.shard_nbl * pl.nwaves, a)) return rc;
    HIP_TRY(mapn::launch_force_sym(a, pl.waves, c->compute));
    if (timer) { HIP_TRY(hipEventRecord(timer->force_done, c->compute)); timer->has_force = true; }

    mapn::SymShardArgs h{};
    h.pos_old = a.pos_old; h.vel_old = a.vel_old; h.pos_new = a.pos_new; h.vel_new = a.vel_new;
    h.arow = a.arow; h.brow = a.brow; h.brow1 = a.brow1; h.tab = a.tab;
    for (uint32_t q = 0; q < world; q++) h.recv_peer[q] = (q == rank ? c->sym_recv : c->sym_send) + (size_t)q * c->count;   // own reactions: straight to where the reduce launch reads
    h.recv_mine = c->sym_recv;
    h.status = c->async_status;
    h.rank = rank; h.world = world; h.count = c->count;
    h.active = c->n; h.count_active = c->count;
    h.nb = a.nb; h.nbl = a.shard_nbl; h.a0 = a.a0; h.half_d = a.half_d; h.parts = pl.parts; h.nwaves = pl.nwaves; h.max_meetings = pl.max_meetings; h.sets = pl.sets;
    h.send_mask = c->sym_send_mask; h.recv_mask = c->sym_recv_mask;
    h.step = ++c->sym_rccl_step;
    h.send_row = 0;
    h.mass = a.mass; h.dt = a.dt; h.damping = a.damping;
    h.phase = 1;
    HIP_TRY(mapn::launch_sym_shard_exchange(h, c->sym_exchange_cap, c->compute));
    if (mapn::comm_exchange_rows(c->comm, c->sym_send, c->sym_recv, (size_t)c->count * 4, c->sym_send_mask & ~(1u << rank), c->sym_recv_mask & ~(1u << rank), c->compute))
        return fail(MAPN_ERR_COMM, "reaction exchange failed: %s", mapn::comm_last_error());
    h.phase = 2;
    HIP_TRY(mapn::launch_sym_shard_exchange(h, c->sym_exchange_cap, c->compute));
    mapn::ForcePlan p{};
    p.kind = mapn::KERNEL_SYM; p.k = 2 * mapn::SYM_K2; p.waves = pl.waves; p.sb = pl.parts; p.nseg = 1; p.epi = mapn::EPI_ROWS;
    c->last_plan = p; c->last_i_count = c->count; c->last_launches = 2;
    return MAPN_OK;
}

// MAPN_FLAG_XCD_CALIBRATE on a SHARDED context: the dies of this rank's GPU are measured by a temporary UNSHARDED context of the same
// size (its creation-time calibration: no collective anywhere in it, and every die holds blocks of both classes there, so what is
// measured is the dies -- a sharded launch holds ONE block per die), and the weights are kept for when the sharded symmetric step is
// prepared (mapn_set_gather_algorithm 4 / 5 / 6).  Never fatal; done once per context.
int calibrate_for_shard(mapn_ctx *c)
{
    if (!(c->cfg.flags & MAPN_FLAG_XCD_CALIBRATE) || c->sym_xcd_weighted || c->shard_calibrated) return MAPN_OK;
    c->shard_calibrated = true;
    if (!sym_applies(c, true) || (c->count / mapn::SYM_BLOCK) % 8u != 0u) return MAPN_OK;
    mapn_config t = c->cfg;
    t.rank = 0; t.world_size = 1; t.flags = MAPN_FLAG_XCD_CALIBRATE;
    // BOUNDED: beyond 262 144 bodies the temporary context is one of 65 536 -- the same dies under the same kernel, 3.5 MiB of
    // state and about 0.4 s, instead of a second full state, up to 1 GiB of scratch and whole-N steps of 0.15 s (1 Mi) .. 2.5 s (4 Mi) each
    if (c->n > 262144u) t.num_particles = 65536u;
    mapn_ctx *tmp = nullptr;
    const std::string keep = g_last_error;
    std::string note;
    // (the bounded stand-in is ANOTHER job shape -- an unsharded 64-block launch, not this rank's -- so its creation-time A/B
    //  says nothing about this job: before, a job of more than 262 144 bodies had no A/B at all, and it keeps its weights as then)
    g_calibration_without_ab = t.num_particles != c->n;
    const int rc = mapn_create(&t, &tmp);
    g_calibration_without_ab = false;
    if (rc == MAPN_OK && tmp->sym_ready && tmp->sym_plan.xcd_mode != 0u) {
        for (int k = 0; k < 8; k++) c->sym_xcd_w[k] = tmp->sym_plan.xcd_weight[k];
        c->sym_xcd_weighted = true;
    } else {
        note = "MAPN_FLAG_XCD_CALIBRATE (sharded; the dies were measured by a temporary UNSHARDED context of " + std::to_string(t.num_particles) + " bodies): no die weights -- " + (rc != MAPN_OK ? "the temporary calibration context could not be created: " + g_last_error
                                                                                         : (g_last_error.empty() ? std::string("the calibrated plan was not kept") : g_last_error)) + "; the default plan runs";
    }
    if (tmp) (void)mapn_destroy(tmp);
    (void)hipSetDevice(c->device);
    g_last_error = note.empty() ? keep : note;             // (never fatal, but no longer silent)
    return MAPN_OK;
}

// MAPN_FLAG_XCD_CALIBRATE: measure the dies under the context's OWN state and give the plan their weights; the state, the fence
// value and the buffer index come back exactly as they were (nothing has been exported yet at creation, so nobody can have seen
// the steps in between).  Never fatal: where it does not apply the default plan stays and the note is left in mapn_last_error().
int calibrate_at_creation(mapn_ctx *c)
{
    if (!(c->cfg.flags & MAPN_FLAG_XCD_CALIBRATE) || c->cfg.world_size != 1) return MAPN_OK;   // (a sharded context: calibrate_for_shard, when its symmetric step is prepared)
    if (!sym_eligible(c, c->n) || c->sym_plan.nb % 8u != 0u) {
        g_last_error = "MAPN_FLAG_XCD_CALIBRATE: XCD weights do not apply to this context (they need the unsharded symmetric kernel and a block count that is a multiple of 8); the default plan runs";
        return MAPN_OK;
    }
    std::vector<float> pos[2], vel[2];
    for (uint32_t b = 0; b < 2; b++) {
        pos[b].resize((size_t)c->n * 4); vel[b].resize((size_t)c->n * 3);
        if (int rc = mapn_download_buffer(c, b, pos[b].data(), vel[b].data())) return rc;
    }
    const uint64_t fence = c->fence_value, completed = c->completed;
    const uint32_t index = c->buffer_index;
    const bool timers = c->timers_enabled;
    const float ema = c->ema_seconds;
    c->timers_enabled = false;
    // clock ramp: the chip needs a few hundred ms of load before the dies settle at the speeds they hold under this kernel
    int rc = MAPN_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        float ms = 0.f;
        (void)hipEventRecord(e0, c->compute);
        const int per_burst = c->n <= 131072u ? 8 : c->n <= 524288u ? 2 : 1;     // (a step is 0.6 ms at 65 536 bodies, 2.5 s at 4 Mi: never more than one step past the 200 ms)
        for (int burst = 0; burst < 400 && ms < 200.f && !rc; burst++) {
            for (int k = 0; k < per_burst && !rc; k++) rc = mapn_simulate(c, (int)c->n, 0);
            (void)hipEventRecord(e1, c->compute);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) break;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipGetLastError();
    // Calibrate, apply -- and VERIFY: the calibration reads lone stamped launches, and now and then what it reads is a transient (a die
    // measured 6 - 13 % slow for a few launches: weights like that cost 2 - 4 % per step).  The weighted plan stays only if it wins an A/B
    // of plain steps against the default plan -- best of two bursts each, interleaved, 0.2 % margin: the decision bench.py has always made
    // for itself, now the library's.  A reading that loses gets ONE second reading (round 5: of four creations on one box one lost its
    // A/B by 0.5 % where the other three won by 1.1 - 1.2 %; a second calibration is 0.4 s, only in that case).  (The A/B up to 262 144
    // bodies: beyond, four bursts would cost seconds and a launch is long enough to average the transients out.  MAPN_XCD_VERIFY=0 with
    // MAPN_TEST_HOOKS=1: no A/B, for the tests that need the weighted plan.)
    uint32_t w[8];
    std::string note;
    const char *vf = test_hook("MAPN_XCD_VERIFY");
    const bool verify = c->n <= 262144u && !(vf && vf[0] == '0') && !g_calibration_without_ab;
    if (rc) { note = "MAPN_FLAG_XCD_CALIBRATE: " + g_last_error + "; the default plan runs"; }     // (the clock ramp's steps failed: nothing is calibrated)
    for (int attempt = 0; attempt < 2 && !rc; attempt++) {
        rc = mapn_calibrate_sym_xcds(c, c->n <= 131072u ? 8 : c->n <= 262144u ? 4 : 1, w);
        if (!rc) rc = mapn_set_sym_xcd_weights(c, w);
        if (rc) { note = "MAPN_FLAG_XCD_CALIBRATE: " + g_last_error + "; the default plan runs"; (void)mapn_set_sym_xcd_weights(c, nullptr); rc = MAPN_OK; break; }
        if (!verify) break;
        const double est = 0.6e-3 * ((double)c->n / 65536.0) * ((double)c->n / 65536.0);
        const int kk = std::max(2, std::min(64, (int)(0.04 / est)));
        hipEvent_t a0 = nullptr, a1 = nullptr;
        float best_w = 1e30f, best_d = 1e30f;
        if (hipEventCreate(&a0) == hipSuccess && hipEventCreate(&a1) == hipSuccess) {
            for (int leg = 0; leg < 4 && !rc; leg++) {
                const bool weighted = (leg & 1) == 0;
                if (leg) rc = mapn_set_sym_xcd_weights(c, weighted ? w : nullptr);
                for (int k = 0; k < 4 && !rc; k++) rc = mapn_simulate(c, (int)c->n, 0);        // (the re-plan left the queue idle)
                if (rc) break;
                (void)hipEventRecord(a0, c->compute);
                for (int k = 0; k < kk && !rc; k++) rc = mapn_simulate(c, (int)c->n, 0);
                (void)hipEventRecord(a1, c->compute);
                float ms = 0.f;
                if (rc || hipEventSynchronize(a1) != hipSuccess || hipEventElapsedTime(&ms, a0, a1) != hipSuccess) { best_w = best_d = 1e30f; break; }
                (weighted ? best_w : best_d) = std::min(weighted ? best_w : best_d, ms / (float)kk);
            }
        }
        if (a0) (void)hipEventDestroy(a0);
        if (a1) (void)hipEventDestroy(a1);
        (void)hipGetLastError();
        if (rc) { note = "MAPN_FLAG_XCD_CALIBRATE: " + g_last_error + "; the default plan runs"; (void)mapn_set_sym_xcd_weights(c, nullptr); rc = MAPN_OK; break; }
        const bool keep = best_w < 1e29f && best_w < best_d * 0.998f;
        rc = mapn_set_sym_xcd_weights(c, keep ? w : nullptr);
        if (rc) { note = "MAPN_FLAG_XCD_CALIBRATE: " + g_last_error + "; the default plan runs"; (void)mapn_set_sym_xcd_weights(c, nullptr); rc = MAPN_OK; break; }
        if (keep) { note.clear(); break; }
        char msg[320];
        snprintf(msg, sizeof msg, "MAPN_FLAG_XCD_CALIBRATE: the calibrated plan (weights %u %u %u %u %u %u %u %u, reading %d of 2) did not win its A/B against the default plan "
                 "(%.4f against %.4f ms per step); the default plan runs", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], attempt + 1, (double)best_w, (double)best_d);
        note = msg;
    }
    // put everything back
    rc = mapn_wait_idle(c);
    for (uint32_t b = 0; b < 2 && !rc; b++) {
        if (hipMemcpy(c->pos[b], pos[b].data(), (size_t)c->n * 16, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->vel[b], vel[b].data(), (size_t)c->n * 12, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(MAPN_ERR_HIP, "MAPN_FLAG_XCD_CALIBRATE: restoring the state failed: %s", hipGetErrorString(hipGetLastError()));
    }
    c->fence_value = fence; c->completed = completed; c->buffer_index = index;
    for (int k = 0; k < kTimerRing; k++) c->fence_event_value[k] = 0;
    c->exported_value = 0;
    c->timers_enabled = timers; c->ema_seconds = ema;
    c->steps_enqueued = 0; c->steps_since_reset = 0; c->force_seconds_sum = 0.0; c->force_launches = 0; c->samples.clear();
    c->last_launches = 0;
    drop_graphs(c);
    if (!rc && !note.empty()) g_last_error = note;
    return rc;
}

// a plan as the C ABI describes it (info + the two arrays, capacities checked)
int export_plan(const mapn::SymPlanHost &p, const char *who, mapn_sym_plan_info *info, uint32_t *windows, uint64_t windows_capacity, uint32_t *tables, uint64_t tables_capacity)
{
    info->nb = p.nb; info->groups = p.groups; info->windows = (uint32_t)p.windows.size();
    info->parts = p.parts; info->taper1 = p.taper1; info->taper2 = p.taper2; info->waves = p.waves;
    info->wave_bias[0] = p.bias_hi; info->wave_bias[1] = p.bias_lo;
    info->brows = p.brows; info->max_meetings = p.max_meetings; info->table_stride = p.table_stride;
    info->sets = p.sets; for (int k = 0; k < 8; k++) info->xcd_weight[k] = p.xcd_weight[k];
    info->xcd_mode = p.xcd_mode; info->wgmap_offset = p.wgmap_offset; info->wgmap_entries = p.wgmap_entries; info->la_flip = p.la_flip;
    for (int k = 0; k < 8; k++) info->class_die[k] = p.class_die[k / 4][k % 4];
    if (windows && windows_capacity < 4u * p.windows.size())
        return fail(MAPN_ERR_INVALID_ARGUMENT, "%s: windows_capacity %llu < %zu (the plan has changed since the arrays were sized: query again)", who, (unsigned long long)windows_capacity, 4u * p.windows.size());
    if (windows)
        for (size_t k = 0; k < p.windows.size(); k++) {
            windows[4 * k + 0] = p.windows[k].g0; windows[4 * k + 1] = p.windows[k].g1;
            windows[4 * k + 2] = p.windows[k].meetings[0]; windows[4 * k + 3] = p.windows[k].meetings[1];
        }
    if (tables) {
        if (tables_capacity < p.tables.size()) return fail(MAPN_ERR_INVALID_ARGUMENT, "%s: tables_capacity %llu < %zu", who, (unsigned long long)tables_capacity, p.tables.size());
        std::copy(p.tables.begin(), p.tables.end(), tables);
    }
    return MAPN_OK;
}

}  // namespace host
}  // namespace mapn

extern "C" {

int mapn_set_sym_plan(mapn_ctx *c, uint32_t waves, uint32_t parts, uint32_t taper1, uint32_t taper2, uint32_t groups_per_window,
                      uint32_t wave_bias_hi, uint32_t wave_bias_lo)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    if (int rc = mapn_wait_idle(c)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const bool sharded = c->cfg.world_size > 1;
    if (sharded && !((c->p2p_ready && (c->gather_algo == 4 || c->gather_algo == 5)) || (c->comm && c->gather_algo == 6)))
        return fail(MAPN_ERR_STATE, "set_sym_plan: a sharded context runs the symmetric kernel under gather algorithms 4, 5 and 6 only (set one first)");
    if (waves == 0 && parts == 0) c->sym_user_plan = false;            // back to the default shape
    else {
        if (wave_bias_hi == 0u && wave_bias_lo == 0u) wave_bias_hi = wave_bias_lo = 1u;
        if ((waves != 4 && waves != 8) || parts == 0 || taper1 + taper2 > parts || wave_bias_hi == 0u || wave_bias_lo == 0u || wave_bias_hi > 64u || wave_bias_lo > 64u)
            return fail(MAPN_ERR_INVALID_ARGUMENT, "set_sym_plan: waves must be 4 or 8, parts >= 1, taper1 + taper2 <= parts, wave bias 1 .. 64 (or 0, 0 = equal)");
        c->sym_user_plan = true;
        c->sym_user[0] = waves; c->sym_user[1] = parts; c->sym_user[2] = taper1; c->sym_user[3] = taper2; c->sym_user[4] = groups_per_window;
        c->sym_user[5] = wave_bias_hi; c->sym_user[6] = wave_bias_lo;
    }
    drop_graphs(c);
    if (int rc = prepare_sym(c, sharded)) { c->sym_user_plan = false; std::string keep = g_last_error; (void)prepare_sym(c, sharded); g_last_error = keep; return rc; }
    if (!c->sym_ready) return fail(MAPN_ERR_STATE, "set_sym_plan: the symmetric kernel does not run in this context (%s)", c->sym_note.c_str());
    return MAPN_OK;
}

int mapn_set_sym_xcd_weights(mapn_ctx *c, const uint32_t *w)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    if (int rc = mapn_wait_idle(c)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (w)                                                 // (validated BEFORE anything is touched: ADVICE r3)
        for (int k = 0; k < 8; k++)
            if (w[k] == 0u || w[k] > 4096u) return fail(MAPN_ERR_INVALID_ARGUMENT, "set_sym_xcd_weights: weights must be 1 .. 4096 (1024 = the fastest die)");
    c->sym_xcd_weighted = false;
    if (w)
        for (int k = 0; k < 8; k++) { c->sym_xcd_w[k] = w[k]; c->sym_xcd_weighted = c->sym_xcd_weighted || w[k] != w[0]; }
    if (!c->sym_ready) return MAPN_OK;                     // (kept for when the symmetric step is prepared: a sharded context before algorithm 4 / 5 / 6)
    drop_graphs(c);
    const bool sharded = c->sym_sharded;
    if (int rc = prepare_sym(c, sharded)) { c->sym_xcd_weighted = false; std::string keep = g_last_error; (void)prepare_sym(c, sharded); g_last_error = keep; return rc; }
    return MAPN_OK;
}

int mapn_calibrate_sym_xcds(mapn_ctx *c, int steps, uint32_t out[8])
{
    if (!c || !out || steps < 1) return fail(MAPN_ERR_INVALID_ARGUMENT, "calibrate_sym_xcds: bad argument");
    if (!(sym_eligible(c, c->n) || sym_shard_eligible(c, c->n)))
        return fail(MAPN_ERR_STATE, "calibrate_sym_xcds: the symmetric kernel does not run in this context");
    HIP_TRY(hipSetDevice(c->device));
    const mapn::SymPlanHost &pl = c->sym_plan;
    const uint32_t nblk = c->sym_sharded ? c->count / mapn::SYM_BLOCK : pl.nb;
    // (refused BEFORE a step is taken: in a sharded job every rank calls this, and all of them must take the same number of steps)
    if (nblk % 8u) return fail(MAPN_ERR_STATE, "calibrate_sym_xcds: a launch covers %u blocks, not a multiple of 8: XCD weights do not apply", nblk);
    // first ALL the stamped steps (collective in a sharded job: nothing below may cut them short on one rank), then the analysis
    std::vector<std::vector<unsigned long long>> stamps;
    int rc = MAPN_OK;
    for (int s = 0; s < steps && !rc; s++) {
        c->stamp_next = true; c->calibrating = true;
        rc = mapn_simulate(c, (int)c->n, 0);
        c->stamp_next = false; c->calibrating = false;
        if (!rc) rc = mapn_wait_idle(c);
        if (rc || !c->timeline_buf || !c->timeline_last) continue;
        std::vector<unsigned long long> tl(6 * c->timeline_last);
        if (hipMemcpy(tl.data(), c->timeline_buf, c->timeline_last * 48, hipMemcpyDeviceToHost) == hipSuccess) stamps.push_back(std::move(tl));
        else (void)hipGetLastError();
    }
    if (rc) return rc;
    // A die is identified by the DISPATCH SLOT of the workgroups it gets -- workgroup number mod 8, what the plan's
    // weights are indexed by -- not by its XCC_ID register (the two numberings need not agree); the register only has to
    // be the same for all waves of a slot, which is checked: if it is not, workgroups are not dealt to the dies round-robin
    // on this device and the weighting would be meaningless.
    std::vector<uint32_t> wg_x(pl.wgmap_entries);                                             // class-aware plan: (block, part) -> blockIdx.x
    for (uint32_t e = 0; e < pl.wgmap_entries; e++) {
        const uint32_t m = pl.tables[pl.wgmap_offset + e];
        wg_x[(size_t)(m >> 16) * pl.parts + (m & 0xffffu)] = e % nblk;
    }
    std::vector<double> per[8], launch_median[8];
    int slot_xcc[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    // (Every wave's time per step is read, the older and the younger wave of a SIMD alike.  Reading the older waves only -- the SIMD
    //  serves them first, so their time is the die's own -- was tried: same gain unsharded, and a LOSS sharded, 1.001 - 1.005 against
    //  0.987 - 0.990 of the default step on one box: profiles/r04_xcd_class_aware_ab.txt.)
    for (const std::vector<unsigned long long> &tl : stamps) {
        for (int x = 0; x < 8; x++) per[x].clear();
        for (size_t wv = 0; wv < tl.size() / 6; wv++) {
            const unsigned long long *o = &tl[6 * wv];
            const uint32_t wg = (uint32_t)(wv / pl.waves), la = wg / pl.parts, part = wg % pl.parts;
            uint32_t x = pl.sets > 2u ? (la + nblk * pl.parts - part) % nblk : (la ^ pl.la_flip);   // blockIdx.x of the workgroup: its number mod 8 is x mod 8
            if (pl.wgmap_entries) x = wg_x[(size_t)la * pl.parts + part];                     // (class-aware plan: from the workgroup map)
            const unsigned slot = x & 7u, xcc = (unsigned)((o[4] >> 32) & 15u);
            if (o[5] < 64 || o[2] <= o[1]) continue;
            if (slot_xcc[slot] < 0) slot_xcc[slot] = (int)xcc;
            else if (slot_xcc[slot] != (int)xcc)
                return fail(MAPN_ERR_STATE, "calibrate_sym_xcds: workgroups of dispatch slot %u ran on XCC %d and %u: not dealt round-robin to the dies", slot, slot_xcc[slot], xcc);
            per[slot].push_back((double)(o[2] - o[1]) / (double)o[5]);   // 100 MHz ticks per step
        }
        // (one figure per die and LAUNCH -- the median over its waves -- and then the median over the launches: the launches are lone
        //  ones out of an idle queue, and a die that is slow for one or two of them, which happens, does not move the result)
        for (int x = 0; x < 8; x++) {
            if (per[x].empty()) continue;
            std::nth_element(per[x].begin(), per[x].begin() + per[x].size() / 2, per[x].end());
            launch_median[x].push_back(per[x][per[x].size() / 2]);
        }
    }
    double speed[8], best = 0.0;
    for (int x = 0; x < 8; x++) {
        std::vector<double> &lm = launch_median[x];
        if (lm.empty()) return fail(MAPN_ERR_STATE, "calibrate_sym_xcds: no wave was seen in dispatch slot %d (a partitioned or masked device?)", x);
        std::sort(lm.begin(), lm.end());
        speed[x] = 1.0 / (lm.size() & 1u ? lm[lm.size() / 2] : 0.5 * (lm[lm.size() / 2 - 1] + lm[lm.size() / 2]));
        best = std::max(best, speed[x]);
    }
    for (int x = 0; x < 8; x++) out[x] = (uint32_t)(1024.0 * speed[x] / best + 0.5);
    return MAPN_OK;
}


int mapn_get_sym_plan(mapn_ctx *c, mapn_sym_plan_info *info, uint32_t *windows, uint64_t windows_capacity, uint32_t *tables, uint64_t tables_capacity)
{
    if (!c || !info) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    memset(info, 0, sizeof *info);
    if (!c->sym_ready) {
        snprintf(info->error, sizeof info->error, "%s", c->sym_note.empty() ? "the symmetric kernel does not apply to this context" : c->sym_note.c_str());
        return fail(MAPN_ERR_STATE, "get_sym_plan: %s", info->error);
    }
    info->a0 = c->sym_sharded ? (uint32_t)c->cfg.rank * (c->count / mapn::SYM_BLOCK) : 0u;
    info->nbl = c->sym_sharded ? c->count / mapn::SYM_BLOCK : 0u;
    info->scratch_bytes = c->sym_scratch_bytes;
    info->active_compute_units = c->sym_sharded ? (uint32_t)c->cus_active : 0u;
    info->exchange_workgroups = c->sym_sharded ? c->sym_exchange_cap : 0u;
    return export_plan(c->sym_plan, "get_sym_plan", info, windows, windows_capacity, tables, tables_capacity);
}

}  // extern "C"
