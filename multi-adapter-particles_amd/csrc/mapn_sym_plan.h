// mapn_sym_plan.h -- host-side launch plan of the symmetric all-pairs step (mapn_sym.hip): which meetings
// every wave runs, cut to the STEP so that all waves of a launch carry the same cost, and in how many
// launches ("windows" of partner distance) a step is made so that the reaction scratch stays O(N).
// Plain C++ (no HIP): built once when the context is created / wired for an exchange, uploaded as tables the
// kernels read, exported through the C ABI (mapn_sym_plan_describe) so that the CPU tests check its
// combinatorics and the order-matched oracle restates the device's summation order from the SAME tables.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#if defined(__HIPCC__)
#define MAPN_HOST_DEVICE __host__ __device__
#else
#define MAPN_HOST_DEVICE
#endif

namespace mapn {

// Which block of a half-ring pair (p, p + nb / 2) runs their meetings (even block counts; `half` = nb / 2, else 0): the pairs
// ALTERNATE -- p even: block p, p odd: block p + nb / 2 -- so that the extra group is spread evenly over the two halves of the
// ring, i.e. over the ranks of a sharded job (until round 3 the first half ran all of them: at 65 536 / 8 ranks 0 .. 3 ran 528
// meetings a block and ranks 4 .. 7 512).  The runner's class is 0, the other's 1.
MAPN_HOST_DEVICE inline bool sym_runs_half(uint32_t a, uint32_t half)
{
    if (!half) return false;
    const bool low = a < half;
    const uint32_t p = low ? a : a - half;
    return ((p & 1u) == 0u) == low;
}

constexpr uint32_t SYM_SPLIT_NONE = 0xffffffffu;
constexpr uint32_t SYM_COST_SELF = 6, SYM_COST_SYM = 6;   // relative cost of a step of a block against itself / of a symmetric step: the kernel
                                                           // runs ONE loop for both (see force_sym_kernel), so they cost the same

// Meetings of I-block a (1024 bodies) are numbered by GROUP g and J-block t (16 per group, 64 bodies each):
//   g = 0: the block itself, one-sided;  g = 1 .. D = (nb - 1) / 2: partner block a + g (mod nb), symmetric;
//   g = D + 1 (even nb only): the half-ring partner a + nb / 2, run by ONE block of each pair (p, p + nb / 2): p when p is even, p + nb / 2 when p is odd
//   (mapn_kernels.h sym_runs_half -- alternating, so that the extra group is spread evenly over the ranks of a sharded job).
// CLASS 0 = the blocks that have the half-ring group, class 1 = the others (all blocks when nb is odd).
// SETS.  The eight XCDs of an MI355X do not run at one speed (measured: 0.538 - 0.570 us per step under this kernel, the same
// dies slow on every launch of a box) and a launch gives every XCD the same work, so it ends with the slowest die.  With XCD
// weights (mapn_set_sym_xcd_weights; from a calibration, mapn_calibrate_sym_xcds) the parts of every block are spread over the
// dies -- workgroup (x, y) of the grid runs part y of block (x + y) mod blocks, and lands on XCD x mod 8 when the launch's block
// count is a multiple of 8 -- and a part's share of the block's steps is proportional to the speed of the die it runs on.  The
// tables then exist per SET = class + 2 * (block mod 8): 16 sets instead of 2 ("spread" mode, xcd_mode 1).
// CLASS-AWARE mode (xcd_mode 2, round 4; chosen where it applies): the blocks that run the half-ring group carry 3.1 % more steps
// than the others at 65 536 bodies (one group of 33), in a ONE-round launch nothing hides that, and it is the same size as the
// spread of the dies' speeds -- so the two are played against each other: every class-0 ("heavy") block puts its parts on four
// of the dies, every class-1 block on the other four -- of the 70 ways to split the dies 4 : 4 the one whose speed ratio comes
// closest to the classes' work ratio (65 536 bodies: 33 : 32 groups -- the faster four for the heavy blocks; 262 144: 129 : 128 -- a
// nearly even split) -- a quarter of a block's parts on each of its class's dies, each part sized by the speed of its die (part s
// runs on die class_die[class][s mod 4]).  Which workgroup runs which (block, part) is a
// table the kernel reads (wgmap, appended to `tables`): workgroup (x, y) lands on die x mod 8; the k-th fastest die's workgroups
// take, in order u = x / 8 + (blocks / 8) y, part 4 (u / (blocks / 2)) + k of the class's block number u mod (blocks / 2).  The
// tables stay per class (2 sets).  Needs a half-ring group, parts that are a multiple of 4, a launch of a multiple of 8 blocks half
// of which are class 0 (always so unsharded; a rank's share when it is an even number of blocks).
// A WINDOW is one force launch: the groups [g0, g1).  Inside a window meeting m = (g - g0) * 16 + t, step k of
// meeting m has the linear index 64 m + k, and wave v = part * waves + wave-in-workgroup runs the steps
// [bounds[v], bounds[v + 1]) -- every wave at least 64 of them, so a meeting is cut at most once:
//   * cut between two waves of ONE workgroup: the two partial reactions are added in LDS (first steps + last steps);
//   * cut between two workgroups: the first steps go to the meeting's row, the last steps to the head row
//     brow1[I-block][split[class][m]] (split = the later workgroup's part index, SYM_SPLIT_NONE otherwise), and
//     whoever sums the rows adds row, then head row.
struct SymWindow {
    uint32_t g0, g1;
    uint32_t meetings[2];        // per class
};

struct SymPlanHost {
    uint32_t nb = 0, D = 0, half = 0;        // half = nb / 2 when nb is even, else 0
    uint32_t groups = 0;                     // 1 + D (+ 1 when nb is even)
    uint32_t parts = 0, taper1 = 0, taper2 = 0, waves = 0;
    uint32_t bias_hi = 1, bias_lo = 1;       // share of a workgroup's steps: first half of its waves : second half (1 : 1 = equal waves)
    uint32_t nwaves = 0;                     // parts * waves
    uint32_t brows = 0;                      // reaction-row slots per J-block and window (most symmetric groups in one window)
    uint32_t max_meetings = 0;               // most meetings of a block in one window
    uint32_t sets = 2;                       // table sets per window: 2 (one per class) or 16 (class + 2 * (block mod 8): XCD-weighted parts)
    uint32_t xcd_weight[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // relative speed of the dies the parts were weighted with (xcd_mode != 0)
    uint32_t xcd_mode = 0;                   // 0: no weights, 1: spread (16 sets), 2: class-aware (2 sets + wgmap)
    uint32_t class_die[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // class-aware: the dispatch slots (dies) class 0 / class 1 blocks run on, fastest first
    uint32_t la_flip = 0;                    // no XCD weights, a launch of ONE block per die (a rank's share): 1 = workgroup (x, y) runs block x ^ 1, which puts
                                             // the launch's class-0 blocks on the odd dispatch slots (see build_sym_plan)
    uint32_t wgmap_offset = 0, wgmap_entries = 0;        // class-aware: tables[wgmap_offset + y * blocks + x] = (block of the launch << 16) | part for workgroup (x, y)
    uint32_t table_stride = 0;               // uint32 per window: bounds[sets][nwaves + 1], split[sets][max_meetings]
    std::vector<SymWindow> windows;
    std::vector<uint32_t> tables;            // windows.size() * table_stride (+ wgmap_entries behind them)

    const uint32_t *bounds(size_t window, uint32_t set) const { return tables.data() + window * table_stride + set * (nwaves + 1); }
    const uint32_t *split(size_t window, uint32_t set) const { return tables.data() + window * table_stride + sets * (nwaves + 1) + set * max_meetings; }
};

// groups_per_window: most SYMMETRIC groups one launch may hold (0 = all in one launch).  parts workgroups per
// I-block whose sizes taper 4 : 2 : 1 (the first taper1 parts weigh 4, the next taper2 weigh 2, the rest 1;
// taper1 = parts: equal parts).  Fails (false + err) when a wave would get fewer than 64 steps.
// xcd_weight: null or 8 relative speeds (all equal = none); they take effect only when launch_blocks (the blocks ONE launch
// covers: nb, or a rank's share) is a multiple of 8 -- otherwise the plan is the unweighted one.
// WAVE BIAS.  A SIMD holds two of this kernel's waves (248 registers each) and issues the OLDER one whenever it can: the younger
// runs in the gaps.  Measured (rank 0 of 65 536 / 8, every SIMD alike): two waves of 132 steps each -- the older is done after
// 44 us at its own full speed (22.3 us per 64 steps), the younger, then alone, after 78.  In an 8-wave workgroup the waves
// 0 .. 3 are the older ones of their SIMDs, so the plan can give them bias_hi : bias_lo of the workgroup's steps -- 3 : 1 lets both
// finish together (198 + 66 steps: 69 and 73 us).  bias_hi = bias_lo: equal waves (the 4-wave shape: which of two WORKGROUPS
// of a compute unit is the older one is not known to the host).
// launch_a0: the first block of the launch in the whole job (a rank's first block; 0 unsharded) -- the classes of the launch's
// blocks follow from it.  xcd_mode: 0 = class-aware where it applies, else spread; 1 = spread only.
bool build_sym_plan(uint32_t nb, uint32_t groups_per_window, uint32_t parts, uint32_t taper1, uint32_t taper2, uint32_t waves,
                    uint32_t bias_hi, uint32_t bias_lo, const uint32_t *xcd_weight, uint32_t launch_blocks, uint32_t launch_a0, uint32_t xcd_mode,
                    SymPlanHost &out, std::string &err);

}  // namespace mapn
