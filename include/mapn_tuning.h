/*
 * mapn_tuning.h -- tuning and introspection entry points of libmapn.so: launch plans of the force kernels, XCD calibration,
 * kernel statistics, step timers, the stamped clock diagnostic.  NOT part of the drop-in boundary (include/mapn.h: the surface of
 * the reference's `class Compute` + the sharded mode): nothing here replaces a reference member; the bench harness (bench.py), the
 * parity tests (the order-matched oracle restates the device's summation order from the plans returned here) and the tools use it.
 * Versioned on its own: MAPN_TUNING_ABI_VERSION changes when a struct or signature below does, MAPN_ABI_VERSION does not.
 */
#ifndef MAPN_TUNING_H
#define MAPN_TUNING_H

#include "mapn.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define MAPN_TUNING_ABI_VERSION 2   /* 1 (round 5): split off mapn.h (ABI 3); new: mapn_get_split_plan, mapn_step_form_describe, mapn_kernel_stats.split_active, mapn_shard_describe
                                       2 (round 6): mapn_kernel_stats.split_plans_built + reserved (the struct grew by 8 bytes) */
int mapn_tuning_abi_version(void);

/* Force-kernel statistics accumulated by mapn_simulate: every step records HIP events on the
 * compute stream around the all-pairs force launch(es); avg_seconds is their mean device time
 * since the last reset.  This is what bench.py's roofline line is computed from. */
typedef struct mapn_kernel_stats {
    char     kernel_name[64];
    uint64_t launches;
    double   avg_seconds;
    uint32_t grid_x, grid_y, block_x;   /* of the force launch mapn_simulate enqueued last */
    uint32_t bodies_per_lane, j_splits; /* j_splits = grid_y * (block_x / 64) chunks of 64-body tiles */
    uint32_t fused;              /* 1: the integrator runs inside the force launch (one launch per step) */
    uint32_t grid_z;             /* j-segments per launch (1 unless the sharded overlap structure) */
    uint32_t epilogue;           /* 0 partial rows + reduce_integrate launch, 1 fused in the workgroup,
                                    2 last-arriver ticket (rows summed by the last workgroup of the i-tile),
                                    3 the symmetric kernel's rows + sym_reduce_integrate launch */
    uint32_t force_launches_per_step;
    uint32_t split_active;       /* != 0: the step enqueued last was a PARTIALLY ACTIVE one in its split form -- these many bodies met each other under
                                    the symmetric kernel (a plan of the active blocks alone), the frozen ones acted on them through one one-sided
                                    launch in front (kernel_name "force_sym_kernel", grid of the symmetric launch) */
    uint32_t split_plans_built;  /* host plans built for the split form since the context was created.  A plan belongs to a RING of active blocks
                                    (ceil(active / 1024)), not to a count: the last four rings keep theirs, so a slider dragged in 64-body steps
                                    re-plans once per 1024 bodies and one moving between a few values stops building (and a NEW ring never blocks
                                    mapn_simulate: its plan is uploaded stream-ordered into its own table buffer) */
    uint32_t reserved;
} mapn_kernel_stats;
int mapn_get_kernel_stats(mapn_ctx *ctx, int reset, mapn_kernel_stats *out);
/* The individual samples behind those means: for every step since the last reset that carried timer events (every T-th,
 * mapn_set_timers) its index among the steps since the reset, the step's device time and its force launch's device time
 * (0 when the step had none), in step order -- how the step time is SPREAD over a run.  *count = samples held (at most 4096);
 * at most `capacity` are copied. */
int mapn_get_step_samples(mapn_ctx *ctx, uint32_t *step_index, float *step_ms, float *force_ms, uint32_t capacity, uint32_t *count);

/* Tuning hooks (tests exercise every kernel variant through these; AUTO restores the default).
 * bodies_per_lane in {2,4,8}, waves in {1,2,4,8,16}, sb >= 1 (j-split across workgroups).
 * fused: 0 = two launches (partial rows + reduce_integrate_kernel), 1 = one launch (integrator in
 * the workgroup when sb == 1, else by the last-arriver ticket), 2 = ticket form even when sb == 1.
 * Summation order (what an order-matched checker must reproduce): the 64-body tiles of the j-range
 * are cut into S = sb * waves chunks (the first tiles % S chunks take one tile more); a chunk is
 * summed over ascending j into a zero accumulator with fused multiply-adds; the `waves` chunk sums
 * of a workgroup are added in ascending order, then the sb row sums in ascending order; the mass
 * multiplies the total. */
int mapn_set_force_plan(mapn_ctx *ctx, int kernel, uint32_t bodies_per_lane, uint32_t waves,
                        uint32_t sb, int fused);
/*
 * The SYMMETRIC kernel's launch plan (csrc/mapn_sym_plan.h).  A step is made in `windows` force launches (partner
 * distance groups [g0, g1) each; one launch while the reaction rows fit MAPN_SYM_MAX_MB, default 1024), each
 * followed by a reduce launch that carries the running sum; inside a launch `parts` workgroups of `waves` waves
 * share the meetings of every 1024-body block, cut to the STEP so that every wave carries the same cost.
 * windows[4 k ..] = {g0, g1, meetings of a class-0 block, of a class-1 block}; tables = per window
 * bounds[sets][parts * waves + 1] (first linear step of every wave) then split[sets][max_meetings] (the part whose
 * head row holds the last steps of a meeting cut between two workgroups, 0xffffffff otherwise); class 0 = the blocks
 * that also run the half-ring group (even block count; of the pair (p, p + nb / 2) block p when p is even, block p + nb / 2 when p is odd); set = class, or class + 2 * (block mod 8) when the
 * parts are XCD-weighted (sets = 16; the block counted within its launch).  What an order-matched checker must reproduce
 * (the CPU checker restates exactly this): per wave one fused-multiply-add chain per body over its steps in
 * order; the workgroup's waves added in ascending order into ONE row per (block, part); the reaction of a meeting as
 * two chains (even / odd bodies of the lane) folded once per piece, pieces of a cut meeting added first steps + last
 * steps; per body: rows of its block in ascending part order, then per group in ascending order the meeting's row and
 * its head row, windows in ascending order; the mass multiplies the total.
 * (The tables array holds info->windows * info->table_stride words, then the info->wgmap_entries words of the class-aware workgroup map.
 *  launch_a0: first block of the launch in the whole job -- a rank's first block, 0 unsharded; xcd_mode 0: class-aware where it applies,
 *  else spread; 1: spread only.)
 * mapn_sym_plan_describe computes the plan of a shape WITHOUT a device (CPU tests, the oracle);
 * mapn_get_sym_plan returns the plan a context runs (a0 / nbl: first block and block count of this rank when sharded);
 * mapn_set_sym_plan is the tuning hook (waves 4 or 8; taper1 = taper2 = 0: equal parts; groups_per_window 0: as many
 * as fit; waves = parts = 0: back to the default shape) -- it re-allocates the scratch, never call it per step.
 * WAVE BIAS (wave_bias_hi : wave_bias_lo; 0 : 0 or equal = none).  A SIMD holds two of this kernel's waves and serves the older
 * one first; in an 8-wave workgroup (one per compute unit) waves 0 .. 3 are the older wave of their SIMDs, and the plan gives
 * them hi / lo times the steps of waves 4 .. 7 so that both end together (measured optimum about 10 : 3).  The defaults: 8-wave
 * workgroups with 10 : 3 where a launch's workgroups fill whole rounds of the compute units, 3 : 1 for a sharded launch; the
 * equal-wave 4-wave shape otherwise and whenever several ranks share one device.  Still one linear run of steps per wave, so
 * everything above holds unchanged.
 */
typedef struct mapn_sym_plan_info {
    uint32_t nb, groups, windows;
    uint32_t parts, taper1, taper2, waves;
    uint32_t wave_bias[2];       /* share of a workgroup's steps: first half of its waves : second half (1 : 1 = equal) */
    uint32_t brows, max_meetings, table_stride;
    uint32_t sets;               /* table sets per window: 2 (one per class) or 16 (class + 2 * (block mod 8): XCD-weighted parts) */
    uint32_t xcd_weight[8];      /* the relative die speeds the parts were weighted with (xcd_mode != 0), else 0 */
    uint32_t xcd_mode;           /* 0: no XCD weights; 1: "spread" (16 table sets: the parts of every block spread over the dies); 2: "class-aware"
                                    (round 4): the blocks that run the half-ring group -- 3.1 % more steps at 65 536 bodies -- put their parts on
                                    the four FASTEST dies, the others on the four slowest, every part sized by its die (2 table sets + wgmap) */
    uint32_t wgmap_offset, wgmap_entries;   /* class-aware: tables[wgmap_offset + y * blocks + x] = (block of the launch << 16) | part that
                                               workgroup (x, y) of the grid runs; wgmap_entries = blocks * parts (0: none) */
    uint32_t la_flip;            /* no weights, a sharded launch of one block per die: 1 = workgroup (x, y) runs block x ^ 1, which puts the blocks with the
                                    half-ring group on the odd dispatch slots -- the faster dies by 2 - 3 % on every box measured */
    uint32_t class_die[8];       /* class-aware: dispatch slots (workgroup number mod 8) of class 0's four dies, then class 1's, fastest first */
    uint32_t a0, nbl;
    uint32_t active_compute_units;  /* sharded: compute units that really take this process's workgroups (probed; a CU mask leaves fewer) */
    uint32_t exchange_workgroups;   /* sharded: most workgroups the exchange launch may have (they must all be resident at once) */
    uint64_t scratch_bytes;      /* device memory the symmetric step holds (rows, running sum, tables) */
    char     error[256];         /* why a shape was refused / why the kernel does not run */
} mapn_sym_plan_info;
int mapn_sym_plan_describe(uint32_t nb, uint32_t groups_per_window, uint32_t parts, uint32_t taper1, uint32_t taper2,
                           uint32_t waves, uint32_t wave_bias_hi, uint32_t wave_bias_lo, const uint32_t *xcd_weights,
                           uint32_t launch_blocks, uint32_t launch_a0, uint32_t xcd_mode, mapn_sym_plan_info *info,
                           uint32_t *windows, uint64_t windows_capacity, uint32_t *tables, uint64_t tables_capacity);
/* Two-call pattern: first with windows = tables = NULL to learn info->windows and info->windows * info->table_stride, then with
 * buffers; BOTH capacities are counted in uint32 (4 per window) and checked -- a caller that sized its arrays from an earlier plan
 * (before mapn_set_sym_plan / another MAPN_SYM_MAX_MB changed the window count) gets MAPN_ERR_INVALID_ARGUMENT, not an overflow. */
int mapn_get_sym_plan(mapn_ctx *ctx, mapn_sym_plan_info *info, uint32_t *windows, uint64_t windows_capacity, uint32_t *tables, uint64_t tables_capacity);
int mapn_set_sym_plan(mapn_ctx *ctx, uint32_t waves, uint32_t parts, uint32_t taper1, uint32_t taper2, uint32_t groups_per_window,
                      uint32_t wave_bias_hi, uint32_t wave_bias_lo);
/*
 * The plan of a PARTIALLY ACTIVE step in its split form (mapn_kernel_stats.split_active != 0): the `active` = roundup64(num_active)
 * bodies meet each other under the symmetric kernel with the plan returned in info / windows / tables (a job of `active` bodies: nb =
 * ceil(active / 1024) blocks, the default shape for that size, the context's XCD weights where they apply), and the `frozen` bodies
 * [active, N) act on them through ONE launch of the one-sided kernel in front whose partial rows the first window's reduce launch
 * adds before its own.  What an order-matched checker must reproduce: per active body first the frozen rows -- the j-range
 * [active, N) cut and summed exactly as mapn_set_force_plan describes for (frozen_waves, frozen_sb), rows added in ascending order to
 * zero -- then the symmetric plan's order over the bodies [0, active) as for mapn_get_sym_plan; the mass multiplies the total.
 * MAPN_ERR_STATE until such a step has run.  Same two-call pattern and capacity checks as mapn_get_sym_plan.
 */
/* Which of the three forms an unsharded all-pairs step of a context whose symmetric kernel runs (MAPN_KERNEL_AUTO / SYMMETRIC, N >= 1024)
 * takes for this (N, num_active), WITHOUT a device: 0 = the one-sided kernel over active x N, 1 = the full symmetric step (the reduce
 * launch stops at roundup64(num_active)), 2 = the split form.  A pure function -- the cost model the library itself consults
 * (csrc/mapn_sym_host.cpp, sym_form_by_cost) -- so a given count always sums in the same order; negative = an error. */
int mapn_step_form_describe(uint32_t num_particles, int32_t num_active);
typedef struct mapn_split_info {
    uint32_t active, frozen;
    uint32_t frozen_kernel;          /* mapn_kernel of the launch over the frozen bodies (MAPN_KERNEL_SCALAR / MAPN_KERNEL_LDS) */
    uint32_t frozen_bodies_per_lane, frozen_waves, frozen_sb;
    uint32_t frozen_first;           /* first body of the one-sided launch's j-range: `active` unsharded; SHARDED: the first frozen body THIS rank owns
                                        (`frozen` of them: a rank computes what ITS frozen bodies do to all active ones, and sends the sums to their owners) */
    uint32_t has_plan;               /* 1: info / windows / tables describe the symmetric launch; 0 (sharded only): this rank owns no active body and runs no meetings */
} mapn_split_info;
/* SHARDED context (round 6): the plan of THIS rank's blocks in the active ring (info->a0 / nbl; split->has_plan 0 and no plan where the rank owns
 * no active body), and split->frozen / frozen_first = the frozen bodies this rank owns: see mapn_shard_split_describe. */
int mapn_get_split_plan(mapn_ctx *ctx, mapn_split_info *split, mapn_sym_plan_info *info, uint32_t *windows, uint64_t windows_capacity,
                        uint32_t *tables, uint64_t tables_capacity);
/*
 * XCD-aware parts.  The eight XCDs of an MI355X do not run at one speed under this kernel (measured 0.538 - 0.570 us per
 * step, the same dies slow on every launch of a box) while a launch gives every die the same work, so it ends with the
 * slowest one.  mapn_calibrate_sym_xcds runs `steps` stamped steps (REAL steps, like mapn_measure_clock) and returns the
 * dies' relative speeds (1024 = the fastest), indexed by DISPATCH SLOT -- workgroup number mod 8, not the XCC_ID register; mapn_set_sym_xcd_weights makes the plan spread the parts of every block over
 * the dies (workgroup (x, y) of the grid runs part y of block (x + y) mod blocks) with a share of the block's steps
 * proportional to the speed of the die a part runs on (NULL or equal weights: back to the default plan).  Takes effect where
 * a launch covers a multiple of 8 blocks.  The weights are part of the plan: results are bit-reproducible for given weights,
 * and differ between weightings like between any two summation orders.  bench.py calibrates and says so in its line.
 */
int mapn_calibrate_sym_xcds(mapn_ctx *ctx, int steps, uint32_t out_weights[8]);
int mapn_set_sym_xcd_weights(mapn_ctx *ctx, const uint32_t *weights8);

/*
 * The sharded mode's host arithmetic as data, WITHOUT a device (the CPU multi-process test composes a sharded run from it; the
 * library's own mapn_create and step go through the same helpers): the slice rank `rank` of `world_size` owns, the bodies of it a step
 * with num_active advances (Compute.cpp:1041's rounding: bodies [0, roundup64(num_active)) of the whole job), whether the sharded
 * symmetric step (gather algorithms 4 / 5 / 6) applies to that shape, this rank's blocks, and which ranks it produces reactions for /
 * receives reactions from (bit q of send_mask / recv_mask; mapn_sym_plan_describe(..., launch_blocks = nbl, launch_a0 = a0, ...) gives
 * the rank's launch plan).
 */
typedef struct mapn_shard_info {
    uint32_t first, count;
    uint32_t active_first, active_count;
    uint32_t sym_applies;
    uint32_t nb, nbl, a0;
    uint32_t send_mask, recv_mask;
    uint32_t reserved[2];
} mapn_shard_info;
int mapn_shard_describe(uint32_t num_particles, int32_t rank, int32_t world_size, int32_t num_active, mapn_shard_info *out);
/*
 * ... and of a PARTIALLY ACTIVE step of a sharded job in its split form (gather algorithms 4 / 5, round 6; Particles.cpp:391-394's slider on P ranks):
 * the bodies [0, active) of the whole job, active = roundup64(num_active), advance and form a ring of `ring_blocks` 1024-body blocks of
 * their own.  Rank `rank` runs the meetings of ITS blocks in that ring (`blocks` of them from block `first_block`; 0: its slice is
 * frozen) under the plan mapn_sym_plan_describe(ring_blocks, ..., launch_blocks = blocks, launch_a0 = first_block) gives, and integrates
 * its `active_count` active bodies; the `frozen_count` FROZEN bodies it owns (from `frozen_first`) still exert force, and their OWNER
 * computes it -- one one-sided launch over active x (its frozen bodies) -- and sends the sums to the active bodies' owners in the same
 * rows as the reactions (summation order per destination body: its frozen rows ascending from zero, then the reactions of its blocks as in
 * the all-active sharded step).  send_mask / recv_mask: whom it sends rows to / receives rows from.  `applies`: 1 when the step takes this
 * form for the shape (a pure function of N, P and the count: every rank decides alike), 0: it runs the one-sided kernel and pulls.
 */
typedef struct mapn_shard_split_info {
    uint32_t applies, active;
    uint32_t ring_blocks, blocks, first_block, active_count;
    uint32_t frozen_first, frozen_count;
    uint32_t send_mask, recv_mask;
    uint32_t reserved[2];
} mapn_shard_split_info;
int mapn_shard_split_describe(uint32_t num_particles, int32_t rank, int32_t world_size, int32_t num_active, mapn_shard_split_info *out);

/* Sharded mode: switch the own/remote overlap structure (MAPN_FLAG_SHARD_OVERLAP) at run time, so a
 * launcher can time both structures on the node it runs on; all ranks must agree. */
int mapn_set_shard_overlap(mapn_ctx *ctx, int enabled);

/* The shader clock the chip HOLDS under this kernel (it lowers its clock under load): runs `steps`
 * ordinary steps whose force launch additionally stamps s_memtime / s_memrealtime around every
 * wave's pair loop into a scratch buffer nothing else reads (no stamp executes in a normal launch),
 * and reports the median over waves of d(s_memtime) / d(s_memrealtime) x 100 MHz.
 * SIDE EFFECTS: these are real steps -- positions, velocities, fence value and buffer index advance exactly as by
 * `steps` calls of mapn_simulate(ctx, N, 0) (no consumer wait); in a sharded job every rank must call it.  Scalar-cache
 * and symmetric force kernels, all-pairs mode only: anything else is refused BEFORE a step is taken. */
typedef struct mapn_clock_info {
    double   shader_clock_ghz;       /* median over the stamped waves of the last diagnostic launch */
    double   shader_clock_ghz_p10, shader_clock_ghz_p90;
    double   median_wave_cycles;     /* shader cycles one wave spent in its pair loop */
    uint32_t waves_stamped, steps;
} mapn_clock_info;
int mapn_measure_clock(mapn_ctx *ctx, int steps, mapn_clock_info *out);

/* Step timers: 0 = off, T >= 1 = record the event pair on every T-th step (default 1: every
 * step, like the reference's D3D12GpuTimer).  Each hipEventRecord costs a few microseconds of
 * queue time, which matters once a sharded step is ~0.1 ms. */
int mapn_set_timers(mapn_ctx *ctx, int interval);
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MAPN_TUNING_H */
