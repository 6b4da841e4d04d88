// mapn_sym_plan.cpp -- see mapn_sym_plan.h.  Host only, no HIP.
#include "mapn_sym_plan.h"

#include <algorithm>
#include <cstdio>

#include "mapn_tuning.h"

namespace mapn {

namespace {

constexpr uint32_t JPI = 16;     // 64-body J-blocks per 1024-body I-block

uint32_t units(uint32_t x, uint32_t t1, uint32_t t2)
{
    return x <= t1 ? 4u * x : x <= t1 + t2 ? 4u * t1 + 2u * (x - t1) : 4u * t1 + 2u * t2 + (x - t1 - t2);
}

}  // namespace

bool build_sym_plan(uint32_t nb, uint32_t groups_per_window, uint32_t parts, uint32_t taper1, uint32_t taper2, uint32_t waves,
                    uint32_t bias_hi, uint32_t bias_lo, const uint32_t *xcd_weight, uint32_t launch_blocks, uint32_t launch_a0, uint32_t xcd_mode,
                    SymPlanHost &out, std::string &err)
{
    char msg[256];
    if (bias_hi == 0u && bias_lo == 0u) bias_hi = bias_lo = 1u;
    if (nb == 0 || parts == 0 || waves == 0 || 64u % waves != 0u || taper1 + taper2 > parts || bias_hi == 0u || bias_lo == 0u || bias_hi > 64u || bias_lo > 64u ||
        (bias_hi != bias_lo && (waves & 1u))) {
        snprintf(msg, sizeof msg, "symmetric plan: bad shape nb=%u parts=%u (%u, %u) waves=%u bias %u : %u", nb, parts, taper1, taper2, waves, bias_hi, bias_lo);
        err = msg;
        return false;
    }
    SymPlanHost p;
    p.nb = nb; p.D = (nb - 1u) / 2u; p.half = (nb & 1u) ? 0u : nb / 2u;
    const uint32_t gsym = p.D + (p.half ? 1u : 0u);            // symmetric groups 1 .. gsym
    p.groups = 1u + gsym;
    p.parts = parts; p.taper1 = taper1; p.taper2 = taper2; p.waves = waves; p.nwaves = parts * waves;
    p.bias_hi = bias_hi; p.bias_lo = bias_lo;
    const uint32_t min_steps = 64u;          // (fewer is no gain: the younger wave of a SIMD ends right after the older one whatever its share, measured 66 .. 58 steps)
    bool weighted = false;
    if (xcd_weight && (launch_blocks ? launch_blocks : nb) % 8u == 0u) {
        for (int k = 0; k < 8; k++) weighted = weighted || xcd_weight[k] != xcd_weight[0];
        for (int k = 0; k < 8; k++) if (xcd_weight[k] == 0u || xcd_weight[k] > 4096u) { err = "symmetric plan: XCD weights must be 1 .. 4096 (1024 = the fastest die)"; return false; }   // (bounded so that cost x weight sums stay far inside 64 bits: ADVICE r3)
    }
    // class-aware where it applies: heavy blocks (class 0) on the four fastest dies, the others on the four slowest
    const uint32_t B = launch_blocks ? launch_blocks : nb;
    std::vector<uint32_t> cls_blocks[2];
    if (weighted && xcd_mode != 1u && p.half && parts % 4u == 0u && B <= 65535u && parts <= 65535u)
        for (uint32_t la = 0; la < B; la++) cls_blocks[sym_runs_half(launch_a0 + la, p.half) ? 0 : 1].push_back(la);
    const bool class_aware = !cls_blocks[0].empty() && cls_blocks[0].size() == cls_blocks[1].size();
    p.xcd_mode = !weighted ? 0u : class_aware ? 2u : 1u;
    p.sets = p.xcd_mode == 1u ? 16u : 2u;
    for (int k = 0; k < 8; k++) p.xcd_weight[k] = weighted ? xcd_weight[k] : 0u;
    uint32_t die_class[8] = {0, 0, 0, 0, 0, 0, 0, 0}, die_rank[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // of a dispatch slot: its class and its place among the class's dies
    if (class_aware) {
        // which four dies take the heavy blocks: of the 70 ways to split the eight dies 4 : 4, the one whose speed ratio comes
        // closest to the classes' work ratio -- (groups of a class-0 block) / sum of its dies' speeds against the same for class 1,
        // the larger of the two as small as possible (65 536 bodies: 33 : 32 groups, so the faster four; 262 144: 129 : 128, a
        // nearly even split)
        const uint64_t work[2] = {(uint64_t)gsym + 1u, (uint64_t)gsym};               // groups of a step, the block itself included
        uint32_t best_mask = 0x0f;
        double best = 1e300;
        for (uint32_t mask = 0; mask < 256u; mask++) {
            if (__builtin_popcount(mask) != 4) continue;
            uint64_t sa = 0, sb = 0;
            for (uint32_t d = 0; d < 8u; d++) ((mask >> d) & 1u ? sa : sb) += xcd_weight[d];
            const double cost = std::max((double)work[0] / (double)sa, (double)work[1] / (double)sb);
            if (cost < best) { best = cost; best_mask = mask; }
        }
        for (uint32_t c = 0; c < 2u; c++) {
            uint32_t dies[4], n = 0;
            for (uint32_t d = 0; d < 8u; d++) if ((((best_mask >> d) & 1u) != 0u) == (c == 0u)) dies[n++] = d;
            std::stable_sort(dies, dies + 4, [&](uint32_t a, uint32_t b) { return xcd_weight[a] > xcd_weight[b]; });
            for (uint32_t k = 0; k < 4u; k++) { p.class_die[c][k] = dies[k]; die_class[dies[k]] = c; die_rank[dies[k]] = k; }
        }
    }
    // windows: the symmetric groups in nwin runs of (nearly) equal length; the block itself rides in the first
    const uint32_t cap = groups_per_window ? groups_per_window : std::max(1u, gsym);
    const uint32_t nwin = std::max(1u, (gsym + cap - 1u) / cap);
    for (uint32_t k = 0; k < nwin; k++) {
        SymWindow w{};
        const uint32_t s0 = 1u + (uint32_t)(((uint64_t)gsym * k) / nwin), s1 = 1u + (uint32_t)(((uint64_t)gsym * (k + 1u)) / nwin);
        w.g0 = k == 0 ? 0u : s0;
        w.g1 = s1;
        const uint32_t ng = w.g1 - w.g0;
        const bool has_half = p.half && w.g1 == p.groups;
        w.meetings[0] = JPI * ng;
        w.meetings[1] = JPI * (has_half ? ng - 1u : ng);
        p.brows = std::max(p.brows, w.g1 - std::max(w.g0, 1u));
        p.max_meetings = std::max(p.max_meetings, w.meetings[0]);
        p.windows.push_back(w);
    }
    p.brows = std::max(p.brows, 1u);
    p.table_stride = p.sets * (p.nwaves + 1u) + p.sets * p.max_meetings;
    p.tables.assign((size_t)nwin * p.table_stride, SYM_SPLIT_NONE);

    for (uint32_t k = 0; k < nwin; k++) {
        const SymWindow &w = p.windows[k];
        for (uint32_t set = 0; set < p.sets; set++) {
            const uint32_t cls = set & 1u, r = set >> 1;           // r = block mod 8: which die part s of the block runs on is (r - s) mod 8
            uint32_t *bounds = p.tables.data() + (size_t)k * p.table_stride + set * (p.nwaves + 1u);
            uint32_t *split = p.tables.data() + (size_t)k * p.table_stride + p.sets * (p.nwaves + 1u) + set * p.max_meetings;
            const uint32_t M = w.meetings[cls], L = 64u * M;
            const uint32_t Ls = (w.g0 == 0u && M) ? 64u * JPI : 0u;        // steps of the block against itself (weighted by SYM_COST_SELF)
            const uint64_t ctot = (uint64_t)SYM_COST_SELF * Ls + (uint64_t)SYM_COST_SYM * (L - Ls);
            // weight of part s: its size in the taper (4 : 2 : 1) times the speed of the die it runs on
            auto part_weight = [&](uint32_t s) -> uint64_t {
                const uint64_t size = units(s + 1u, taper1, taper2) - units(s, taper1, taper2);
                if (class_aware) return size * p.xcd_weight[p.class_die[cls][s & 3u]];      // part s runs on the (s mod 4)-th die of its class
                return size * (weighted ? p.xcd_weight[(r + 8u * parts - s) & 7u] : 1u);
            };
            // the first half of a workgroup's waves (the older wave of every SIMD) weighs bias_hi, the second half bias_lo
            auto wave_units = [&](uint32_t ww) -> uint64_t {               // weight of the waves before wave ww of a workgroup
                const uint32_t h = waves / 2u;
                return ww <= h ? (uint64_t)bias_hi * ww : (uint64_t)bias_hi * h + (uint64_t)bias_lo * (ww - h);
            };
            const uint64_t wg_units = wave_units(waves);
            uint64_t wtot = 0;
            for (uint32_t s = 0; s < parts; s++) wtot += part_weight(s) * wg_units;
            uint64_t before = 0;                                           // weight of the parts before part s
            for (uint32_t v = 0; v <= p.nwaves; v++) {
                const uint32_t s = v / waves, ww = v % waves;
                if (v && ww == 0u) before += part_weight(s - 1u) * wg_units;
                const uint64_t num = before + (s < parts ? part_weight(s) * wave_units(ww) : 0u);
                const uint64_t target = ctot * num / wtot;                 // cost that lies before wave v
                bounds[v] = target <= (uint64_t)SYM_COST_SELF * Ls ? (uint32_t)(target / SYM_COST_SELF)
                                                                   : Ls + (uint32_t)((target - (uint64_t)SYM_COST_SELF * Ls) / SYM_COST_SYM);
            }
            if (bounds[0] != 0u || bounds[p.nwaves] != L) { err = "symmetric plan: internal error (bounds do not span the meetings)"; return false; }
            // XCD-weighted parts: a part on a slow die is a few per cent smaller, and where the younger waves of a biased workgroup
            // sit right at the 64-step floor (65 536 / 8: 198 + 66 steps) that pushed them under it and the whole shape was refused
            // -- the 2 : 1 bias that fitted instead lost what the weights gained.  The floor is kept INSIDE the part: waves under 64
            // steps are raised to 64 and the part's other waves -- the older ones, which have steps to spare -- give them up.
            if (weighted && L) {
                for (uint32_t s = 0; s < parts; s++) {
                    std::vector<uint32_t> len(waves);
                    uint32_t lack = 0, spare = 0;
                    for (uint32_t ww = 0; ww < waves; ww++) {
                        len[ww] = bounds[s * waves + ww + 1u] - bounds[s * waves + ww];
                        if (len[ww] < min_steps) lack += min_steps - len[ww]; else spare += len[ww] - min_steps;
                    }
                    if (!lack || spare < lack) continue;                   // nothing to do / not to be had: the check below refuses the shape
                    for (uint32_t ww = 0; ww < waves; ww++) if (len[ww] < min_steps) len[ww] = min_steps;
                    while (lack) {                                         // one step at a time from the longest wave (the sums involved are a few dozen)
                        uint32_t big = 0;
                        for (uint32_t ww = 1; ww < waves; ww++) if (len[ww] > len[big]) big = ww;
                        len[big]--; lack--;
                    }
                    for (uint32_t ww = 0; ww + 1u < waves; ww++) bounds[s * waves + ww + 1u] = bounds[s * waves + ww] + len[ww];
                }
            }
            for (uint32_t v = 0; v < p.nwaves && L; v++) {
                if (bounds[v + 1u] - bounds[v] < min_steps) {
                    snprintf(msg, sizeof msg, "symmetric plan: wave %u of window %u would run %u steps (< %u): %u meetings are too few for %u x %u waves (taper %u, %u)",
                             v, k, bounds[v + 1u] - bounds[v], min_steps, M, parts, waves, taper1, taper2);
                    err = msg;
                    return false;
                }
            }
            // cut meetings: the wave holding the first step and the wave holding the last
            uint32_t v = 0;
            for (uint32_t m = 0; m < M; m++) {
                split[m] = SYM_SPLIT_NONE;
                while (bounds[v + 1u] <= 64u * m) v++;                      // wave holding step 64 m
                uint32_t vb = v;
                while (bounds[vb + 1u] <= 64u * m + 63u) vb++;              // wave holding step 64 m + 63
                if (vb > v + 1u) {
                    snprintf(msg, sizeof msg, "symmetric plan: meeting %u of window %u would be cut twice (wave %u runs %u steps inside it)", m, k, v + 1u, bounds[v + 2u] - bounds[v + 1u]);
                    err = msg;
                    return false;
                }
                const bool symmetric = w.g0 + m / JPI >= 1u;
                if (symmetric && vb != v && vb / waves != v / waves) split[m] = vb / waves;
            }
        }
    }
    // No weights, and a launch whose blocks each sit on ONE die (a rank's share of a sharded job: 8 blocks, block x on dispatch slot x
    // mod 8): the blocks with the half-ring group alternate with the others, i.e. they sit either all on the even or all on the odd
    // slots -- and on every box measured (eight of them, unsharded calibrations of round 4: profiles/r04_xcd_class_aware_ab.txt) the
    // odd slots are the faster dies by 2 - 3 %.  Ranks whose heavy blocks sat on the even slots ran 6 % behind in their heavy
    // blocks (3 % more steps x 3 % slower) and were the slower ranks of the job (rank 0 against rank 4: 92.7 against 91.3 us).
    // The heavy blocks go to the odd slots on every rank; calibrated weights (class-aware mode) supersede this.
    if (!weighted && p.half && launch_blocks && launch_blocks < nb && B % 8u == 0u) {
        uint32_t heavy_even = 0, heavy_odd = 0;
        for (uint32_t la = 0; la < B; la++)
            if (sym_runs_half(launch_a0 + la, p.half)) ((la & 1u) ? heavy_odd : heavy_even)++;
        if (heavy_even && !heavy_odd) p.la_flip = 1u;
    }
    if (class_aware) {
        // which (block, part) workgroup (x, y) of the grid runs: it lands on die x mod 8 -- the k-th die of class c -- and takes,
        // in the order u = x / 8 + (B / 8) y, part 4 (u / (B / 2)) + k of the class's block number u mod (B / 2)
        p.wgmap_offset = (uint32_t)p.tables.size();
        p.wgmap_entries = B * parts;
        p.tables.resize(p.tables.size() + p.wgmap_entries);
        for (uint32_t y = 0; y < parts; y++)
            for (uint32_t x = 0; x < B; x++) {
                const uint32_t d = x & 7u, c = die_class[d], k = die_rank[d], u = x / 8u + (B / 8u) * y;
                const uint32_t la = cls_blocks[c][u % (B / 2u)], part = 4u * (u / (B / 2u)) + k;
                p.tables[p.wgmap_offset + (size_t)y * B + x] = (la << 16) | part;
            }
    }
    out = std::move(p);
    return true;
}

}  // namespace mapn

// ---- C ABI: the plan as data, without a device (tests, the order-matched oracle) ------------------
extern "C" int mapn_sym_plan_describe(uint32_t nb, uint32_t groups_per_window, uint32_t parts, uint32_t taper1, uint32_t taper2,
                                      uint32_t waves, uint32_t wave_bias_hi, uint32_t wave_bias_lo, const uint32_t *xcd_weights, uint32_t launch_blocks,
                                      uint32_t launch_a0, uint32_t xcd_mode, mapn_sym_plan_info *info,
                                      uint32_t *windows, uint64_t windows_capacity, uint32_t *tables, uint64_t tables_capacity)
{
    if (!info) return MAPN_ERR_INVALID_ARGUMENT;
    mapn::SymPlanHost p;
    std::string err;
    if (!mapn::build_sym_plan(nb, groups_per_window, parts, taper1, taper2, waves, wave_bias_hi, wave_bias_lo, xcd_weights, launch_blocks, launch_a0, xcd_mode, p, err)) {
        snprintf(info->error, sizeof info->error, "%s", err.c_str());
        return MAPN_ERR_INVALID_ARGUMENT;
    }
    info->error[0] = 0;
    info->nb = p.nb; info->groups = p.groups; info->windows = (uint32_t)p.windows.size();
    info->parts = p.parts; info->taper1 = p.taper1; info->taper2 = p.taper2; info->waves = p.waves; info->wave_bias[0] = p.bias_hi; info->wave_bias[1] = p.bias_lo;
    info->brows = p.brows; info->max_meetings = p.max_meetings; info->table_stride = p.table_stride;
    info->sets = p.sets; for (int k = 0; k < 8; k++) info->xcd_weight[k] = p.xcd_weight[k];
    info->xcd_mode = p.xcd_mode; info->wgmap_offset = p.wgmap_offset; info->wgmap_entries = p.wgmap_entries; info->la_flip = p.la_flip;
    for (int k = 0; k < 8; k++) info->class_die[k] = p.class_die[k / 4][k % 4];
    info->a0 = 0; info->nbl = 0; info->active_compute_units = 0; info->exchange_workgroups = 0; info->scratch_bytes = 0;
    if (windows && windows_capacity < 4u * p.windows.size()) {
        snprintf(info->error, sizeof info->error, "windows_capacity %llu < %zu", (unsigned long long)windows_capacity, 4u * p.windows.size());
        return MAPN_ERR_INVALID_ARGUMENT;
    }
    if (windows)
        for (size_t k = 0; k < p.windows.size(); k++) {
            windows[4 * k + 0] = p.windows[k].g0; windows[4 * k + 1] = p.windows[k].g1;
            windows[4 * k + 2] = p.windows[k].meetings[0]; windows[4 * k + 3] = p.windows[k].meetings[1];
        }
    if (tables) {
        if (tables_capacity < p.tables.size()) return MAPN_ERR_INVALID_ARGUMENT;
        std::copy(p.tables.begin(), p.tables.end(), tables);
    }
    return MAPN_OK;
}
