/*
 * mapn.h -- C ABI of the MI355X-native n-body compute step ("libmapn.so").
 *
 * Drop-in boundary: the public surface of the reference's `class Compute`
 * (reference/Particles/Compute.h:33-78 + AdapterShared.h:51-60), whose only caller is
 * `class Particles` (Particles.cpp) from one host thread.  Each entry point below names the
 * reference member it replaces.  Plain pointers and sizes only; no C++ or torch types.
 * What is NOT the reference's surface -- launch plans, XCD calibration, kernel statistics, stamped diagnostics: the entry points the
 * bench harness and the parity tests use to look inside -- lives in mapn_tuning.h and is versioned separately: a caller that
 * replaces `class Compute` (compat/Compute.hpp) needs this header alone.
 *
 * Conventions
 *   - every function returning `int` returns MAPN_OK (0) or a negative mapn_status; the
 *     reference throws HrException / asserts instead (DXSampleHelper.h:29-46) -- the C++ shim
 *     compat/Compute.hpp turns a non-zero status back into std::runtime_error.
 *   - mapn_last_error() returns a thread-local description of the last failure.
 *   - a context is NOT thread-safe (the reference has no locks; all calls come from the UI
 *     thread, Main-Particles.cpp:76-90).
 *   - state layout at the ABI (kept verbatim from the reference):
 *       positions  float4[N], 16-byte stride, w = |accel| of the last step
 *                  (Render.h:85-88, nBodyGravityCS.hlsl:67-70,107)
 *       velocities float3[N], packed 12-byte stride (Compute.h:66-69, nBodyGravityCS.hlsl:72-75)
 *     two ping-pong buffers of each (Compute.h:107-108); a step reads buffer 1-index and
 *     writes buffer index, then flips (Compute.cpp:1022,1034-1035,1003; hlsl:77-81).
 */
#ifndef MAPN_H
#define MAPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* libmapn.so is built with -fvisibility=hidden: what this header declares is ALL it exports */
#endif

#define MAPN_ABI_VERSION 4   /* 4 (round 5): this header is the Compute surface + the sharded mode only -- plans, calibration, kernel statistics and the
                                other tuning / introspection entry points moved to mapn_tuning.h, versioned on their own (MAPN_TUNING_ABI_VERSION) */

typedef struct mapn_ctx mapn_ctx;

typedef enum mapn_status {
    MAPN_OK = 0,
    MAPN_ERR_INVALID_ARGUMENT = -1,
    MAPN_ERR_HIP = -2,          /* a HIP runtime call failed; text in mapn_last_error() */
    MAPN_ERR_NO_DEVICE = -3,    /* no usable gfx950 device: there is NO CPU fallback */
    MAPN_ERR_COMM = -4,         /* RCCL missing or a collective failed */
    MAPN_ERR_STATE = -5         /* call not valid in the context's current mode */
} mapn_status;

typedef enum mapn_force_mode {
    MAPN_FORCE_ALL_PAIRS = 0,   /* sum of bodyBodyInteraction over all bodies, hlsl:44-57 */
    MAPN_FORCE_CENTRAL_WELL = 1 /* CSMain exactly as shipped, hlsl:92-101 */
} mapn_force_mode;

/* Kernel selection for the all-pairs force (MAPN_KERNEL_AUTO picks per N and device). */
typedef enum mapn_kernel {
    MAPN_KERNEL_AUTO = 0,       /* the symmetric kernel where it applies, else the scalar-cache kernel */
    MAPN_KERNEL_LDS = 1,        /* j-tiles staged through LDS, broadcast ds_read */
    MAPN_KERNEL_SCALAR = 2,     /* j-bodies through the scalar cache into SGPRs */
    MAPN_KERNEL_SYMMETRIC = 3   /* Newton's third law: every unordered pair evaluated once, feeding both bodies
                                   (csrc/mapn_sym.hip).  Applies to the unsharded step with N >= 1024 (and to the sharded
                                   step under gather algorithms 4 / 5 / 6).  A step with num_active < N takes whichever of
                                   THREE forms is cheapest for that count -- the full symmetric step (the reduce launch stops
                                   at the active bodies), the split form (active x active symmetric, active x frozen
                                   one-sided) or the one-sided kernel over active x N: mapn_simulate below,
                                   mapn_step_form_describe.  Its
                                   scratch is O(N): a step is made in as many launches (windows of partner distance) as
                                   keep the reaction rows within MAPN_SYM_MAX_MB (default 1024), and is allocated by
                                   mapn_create -- which fails if the memory is not to be had.  Under
                                   MAPN_KERNEL_AUTO the same failure only selects the one-sided kernel
                                   (mapn_get_sym_plan, mapn_tuning.h, tells why). */
    /* No MFMA variant (BASELINE configs[4] A/B; re-measured by `pytest -m gpu` on every run since round 4:
       tests/test_gpu_mfma_ab.py): on gfx950 the f32 MFMA shapes do NOT run beside the packed fp32 VALU stream
       of the same SIMD -- their times add (16 v_pk_fma_f32 + one v_mfma_f32_16x16x4_f32: 106 cycles against
       75 + 32) -- so every recast of the pair term is slower than the packed-VALU form (register-resident
       upper bounds: -7 % ... -65 %; the product's scalar kernel at one rank's share of the 1 048 576-body
       job, loads and integrator included, beats all of them), and the two that remove the most VALU work
       lose 3 decimal digits (profiles/r04_ubench_ab.txt, profiles/r04_ubench_ab_rocprofv3_kernel_stats.csv,
       profiles/r02_mfma_recast_error.txt). */
} mapn_kernel;

/* The three #if variants of LoadParticles (Compute.cpp:581-583), all seeded per body. */
typedef enum mapn_init_variant {
    MAPN_INIT_LCG = 0,          /* USE_SCALAR_OPTIMIZED: fast_rand LCG, Compute.cpp:711-749 (default) */
    MAPN_INIT_SSE = 1,          /* USE_SIMD_OPTIMIZED: rand_sse 4-lane LCG, Compute.cpp:751-793 */
    MAPN_INIT_MT = 2            /* USE_ORIG: mt19937 + uniform_real_distribution, Compute.cpp:686-708 */
} mapn_init_variant;

#define MAPN_FLAG_USE_GRAPH   0x1u  /* replay the step from a captured hipGraph */
#define MAPN_FLAG_NO_INIT     0x2u  /* leave state zeroed; caller will mapn_upload_state() */
#define MAPN_FLAG_SHARD_OVERLAP 0x4u /* sharded mode: own-segment launch overlapped with the all-gather */
#define MAPN_FLAG_STRICT_CONSUMER 0x8u /* mapn_simulate(wait_value) returns MAPN_ERR_STATE instead of queueing
                                          the wait when the consumer has not yet signalled wait_value - 1 */
#define MAPN_FLAG_XCD_CALIBRATE 0x10u  /* mapn_create / mapn_create_from measure the eight dies' speeds under the symmetric kernel
                                          (about 0.2 s: a clock-ramp phase, then mapn_calibrate_sym_xcds on the context's own state,
                                          which is put back bit for bit together with the fence value and the buffer index) and
                                          size every part of the launch plan by the die it runs on (mapn_set_sym_xcd_weights) --
                                          the plan bench.py's headline number is measured with, for any C-ABI caller
                                          (compat/Compute.hpp: one config bit).  Up to 262 144 bodies the weighted plan is then VERIFIED
                                          (another 0.2 s): plain steps under it and under the default plan, interleaved, best of two
                                          bursts each -- it stays only if it wins by 0.2 %; a reading that loses gets ONE second reading (another
                                          0.4 s, only then), and if that loses too the default plan runs and mapn_last_error() says so,
                                          with the weights that lost (the calibration reads lone stamped launches and can catch a
                                          transient).  A hint: where XCD weights do not apply
                                          (one-sided kernel, block count not a multiple of 8, a partitioned
                                          device) creation succeeds with the default plan and mapn_last_error() says why.  On a SHARDED
                                          context the flag acts when the sharded symmetric step is prepared (mapn_set_gather_algorithm
                                          4 / 5 / 6): a temporary UNSHARDED context -- of the same size up to 262 144 bodies, of 65 536 bodies beyond
                                          (the same dies; about 0.4 s either way) -- measures this rank's GPU (no collective in it) and the rank's
                                          launch is planned with those weights; where that fails or the weights are not kept mapn_last_error() says why;
                                          mapn_get_sym_plan(...)->xcd_mode != 0 tells whether the weights are in use. */

/*
 * Everything `Compute::Compute(numParticles, adapter, useIntelExt, old)` (Compute.h:36-39)
 * takes, plus what the reference hard-codes: the cbuffer values (Compute.cpp:542-546), the
 * shader constants (nBodyGravityCS.hlsl:37-38) and the initial-state constants
 * (defines.h:39,42).  mapn_config_default() fills in exactly those values.
 */
typedef struct mapn_config {
    uint32_t struct_size;        /* sizeof(mapn_config), for ABI growth */
    uint32_t num_particles;      /* N; reference range 256Ki..4Mi (defines.h:44-45), any N >= 1 here */
    int32_t  device;             /* HIP device ordinal (replaces IDXGIAdapter1*) */
    int32_t  force_mode;         /* mapn_force_mode */
    float    mass;               /* 70000  (hlsl:38) */
    float    softening_squared;  /* 25     (hlsl:37) */
    float    dt;                 /* 0.1f   (Compute.cpp:545) */
    float    damping;            /* 1.0f   (Compute.cpp:546) */
    uint32_t seed;               /* initial-state seed (reference: random_device, Compute.cpp:679) */
    float    spread;             /* 400    (defines.h:42 PARTICLE_SPREAD) */
    float    initial_speed;      /* 15     (defines.h:39 INITIAL_PARTICLE_SPEED) */
    uint32_t flags;              /* MAPN_FLAG_* */
    int32_t  kernel;             /* mapn_kernel */
    int32_t  rank;               /* shard index p in [0, world_size) */
    int32_t  world_size;         /* number of shards P (1 = unsharded); P must divide N */
    int32_t  init_variant;       /* mapn_init_variant: which LoadParticles variant generates the state */
    int32_t  reserved[4];
} mapn_config;

int mapn_abi_version(void);
const char *mapn_last_error(void);

/* Fills cfg with the reference's literal constants, device 0, ALL_PAIRS, seed 1, 1 shard. */
int mapn_config_default(mapn_config *cfg);

/*
 * Compute::Compute(...) fresh path (Compute.cpp:72-98 -> Initialize, InitializeParticles,
 * WaitForGpu).  Generates the seeded two-shell initial state (unless MAPN_FLAG_NO_INIT),
 * uploads it to BOTH ping-pong buffers (Compute.cpp:881-882,903-904) and blocks until ready.
 */
int mapn_create(const mapn_config *cfg, mapn_ctx **out_ctx);

/*
 * Compute::Compute(..., Compute* old) migrate path (Compute.cpp:91-94 -> CopyState :303-410):
 * new context on cfg->device takes both position buffers, both velocity buffers and the
 * buffer index from `old` (device-to-device, possibly across GPUs).  `old` stays valid and
 * is drained first, as Particles.cpp:467-471 does.
 */
int mapn_create_from(const mapn_config *cfg, mapn_ctx *old, mapn_ctx **out_ctx);

/* Compute::~Compute (Compute.cpp:102-123): drain the device, then free. */
int mapn_destroy(mapn_ctx *ctx);

/*
 * Compute::Simulate(int numActive, UINT64 sharedFenceValue) (Compute.h:48, Compute.cpp:1009-1055).
 * Asynchronous: enqueues one step and returns.  Bodies [0, min(roundup64(num_active), N))
 * advance (Compute.cpp:1041), the rest of the written buffer is left untouched.  If a consumer
 * is attached (mapn_set_consumer), the step first waits, ON THE DEVICE, until the consumer has
 * signalled `wait_value - 1` (Compute.cpp:1012 queues `Wait(sharedRenderFence, v - 1)`
 * unconditionally, and so does this: if the consumer has neither signalled nor registered an event
 * yet, the compute stream parks until it does -- bounded by mapn_set_timeouts, default 10 s, after
 * which the next call returns MAPN_ERR_STATE).  Pass 0 when there is no consumer.  With
 * MAPN_FLAG_STRICT_CONSUMER an unsignalled value is a loud MAPN_ERR_STATE instead of a queued wait.
 * A PARTIALLY ACTIVE all-pairs step (num_active < N) runs the cheapest of three forms for that (N, num_active) -- always the same one,
 * so results stay bit-reproducible: the full symmetric step whose reduce launch stops early; the one-sided kernel over active x N; or
 * the SPLIT form (active x active under the symmetric kernel with a plan of the active blocks alone + active x frozen one-sided: measured
 * 1.16 x the one-sided step at half of the bodies active -- the bound there is 1.18 x -- to 1.34 x at 7/8).  The split form's plans belong
 * to the RING of active blocks (ceil(active / 1024)) and are kept for the last four rings; a NEW ring builds its plan on the host and uploads it stream-ordered into its own table buffer, rows that
 * have to grow are allocated anew and the old ones freed later, once the stream has run dry: the call never waits for the device, also
 * while a step is parked behind the consumer's fence (mapn_kernel_stats.split_plans_built counts the plans built).
 * Afterwards the fence value is +1 and the buffer index flipped (MoveToNextFrame,
 * Compute.cpp:993-1004).  A device-side wait that timed out in an EARLIER step (peer-to-peer
 * exchange, consumer fence) makes this call fail with MAPN_ERR_COMM / MAPN_ERR_STATE.
 */
int mapn_simulate(mapn_ctx *ctx, int num_active, uint64_t wait_value);

/* Compute::GetFenceValue (Compute.h:64): the value the NEXT simulate will signal. */
uint64_t mapn_fence_value(const mapn_ctx *ctx);

/* Highest fence value the device has completed (ID3D12Fence::GetCompletedValue analogue). */
uint64_t mapn_completed_value(mapn_ctx *ctx);

/* Compute::WaitForGpu (Compute.cpp:928-940): signal, fence value +1, host-block until idle. */
int mapn_wait_idle(mapn_ctx *ctx);

/* m_bufferIndex: the buffer the NEXT step writes; the latest results are in 1 - index. */
uint32_t mapn_buffer_index(const mapn_ctx *ctx);
uint32_t mapn_num_particles(const mapn_ctx *ctx);

/*
 * Compute::GetSharedHandles (Compute.h:54-62, Compute.cpp:944-950) -> {heap, fence,
 * alignedDataSize, bufferIndex}.  HIP analogue: device pointers of the two position buffers,
 * ONE hipEvent_t that the compute stream re-records after every step from this call on (fetch it
 * once, like Render::SetShared opens the fence once, and hipStreamWaitEvent on it every frame),
 * the per-buffer byte size and the buffer index.  Borrowed views, valid until mapn_destroy /
 * mapn_adopt_position_buffers.  Velocities are never exported (Compute.cpp:231-237).
 */
typedef struct mapn_shared_handles {
    void    *positions[2];       /* device float4[N] */
    void    *step_done_event;    /* hipEvent_t, the same handle for the context's lifetime, re-recorded on the
                                    compute stream after each step */
    uint64_t aligned_data_size;  /* bytes per position buffer, 64 KiB aligned (Compute.cpp:185-194) */
    uint32_t buffer_index;
    uint32_t reserved;
} mapn_shared_handles;
int mapn_get_shared_handles(mapn_ctx *ctx, mapn_shared_handles *out);

/*
 * The consumer's fence (the `in_fenceHandle` of GetSharedHandles / `in_fence` of SetAsync):
 * a host-visible monotonically increasing counter the consumer bumps with
 * mapn_consumer_signal(); mapn_simulate(wait_value) does not overwrite a position buffer
 * before the counter reaches wait_value - 1.
 */
int mapn_set_consumer(mapn_ctx *ctx, int enabled);
int mapn_consumer_signal(mapn_ctx *ctx, uint64_t value);
/* device-side form: the consumer reaches `value` when hip_event (a hipEvent_t it has already
 * recorded on its own stream) fires; simulate then waits for it on the GPU, not on the host */
int mapn_consumer_signal_event(mapn_ctx *ctx, uint64_t value, void *hip_event);
/* bounds of the device-side waits in milliseconds (0 = leave unchanged): the peer-to-peer
 * exchange's wait for a peer's slice (default 2000: it also covers a peer whose HOST is late
 * enqueueing the step; bench.py sets 1000) and the queued consumer-fence wait (default 10 000).  A wait that gives up is reported by the next mapn_simulate / mapn_wait_idle /
 * mapn_download_* as MAPN_ERR_COMM (naming the peer) / MAPN_ERR_STATE. */
int mapn_set_timeouts(mapn_ctx *ctx, uint32_t p2p_ms, uint32_t consumer_ms);

/*
 * Compute::SetAsync (Compute.h:74-77, Compute.cpp:956-987): compute straight into two
 * caller-owned device float4[N] buffers (same device); the next step writes
 * buffers[1 - buffer_index].  Compute::ResetFromAsyncHelper (Compute.cpp:260-298) undoes it,
 * copying the current positions back into the context's own buffers.
 */
int mapn_adopt_position_buffers(mapn_ctx *ctx, void *buffers[2], uint32_t buffer_index);
int mapn_reset_from_async(mapn_ctx *ctx);

/* AdapterShared::GetGpuTimes (AdapterShared.h:51; D3D12GpuTimer.h:133-160): EMA-20 of the
 * step's device time in seconds, timer name "simulate ms". */
float mapn_last_step_seconds(mapn_ctx *ctx);
const char *mapn_timer_name(void);

/* Vendor-hint stubs (Compute.h:51, AdapterShared.h:54,60): no AMD analogue, always false. */
int mapn_set_use_intel_command_queue_extension(mapn_ctx *ctx, int desired);
int mapn_get_using_intel_command_queue_extension(const mapn_ctx *ctx);
int mapn_get_is_uma(const mapn_ctx *ctx);

/*
 * State hand-off used by the parity harness and by checkpoint/restore (the reference has only
 * the in-memory CopyState).  upload writes the same data into BOTH ping-pong buffers, like
 * InitializeParticles; download reads the latest state (buffer 1 - index) after draining.
 * pos4: N*4 floats, vel3: N*3 floats, host memory; either may be NULL.
 * Sharded with a peer-to-peer exchange (gather algorithms 2 - 5): the peers read from / store into this context's
 * position buffers, so upload (and mapn_load_snapshot) is COLLECTIVE -- every rank must have drained its own
 * work (mapn_wait_idle) and the ranks must have met at a barrier of the launcher before any of them uploads.
 */
int mapn_upload_state(mapn_ctx *ctx, const float *pos4, const float *vel3);
int mapn_download_state(mapn_ctx *ctx, float *pos4, float *vel3);
/* Raw access to one ping-pong buffer pair (index 0 or 1), for exact state comparison. */
int mapn_download_buffer(mapn_ctx *ctx, uint32_t index, float *pos4, float *vel3);

/*
 * Consumer hand-off, what Render::CopySimulationResults does (Render.cpp:789-831): on the
 * CONSUMER's stream (hipStream_t), wait for the latest step's completion event, then copy the
 * first num_copied positions (num_copied * 16 bytes, Render.cpp:814) of the latest buffer to
 * `dst` (device or pinned host memory).  Asynchronous; the consumer then records an event on its
 * stream and reports it with mapn_consumer_signal_event(value = the fence value it waited for),
 * which is what lets the next mapn_simulate overwrite that buffer (Compute.cpp:1012).
 */
int mapn_copy_positions_async(mapn_ctx *ctx, uint32_t num_copied, void *dst, void *consumer_stream);

/*
 * The same hand-off across a PROCESS boundary (the reference shares an NT handle of the heap and
 * of the fences, Compute.cpp:163-201,434-435,944-950; Render.cpp:222-251,612-617): the compute
 * process exports a blob (hipIpc handles of the position heap and of a small uncached block holding
 * both fences as memory words); a renderer / analysis process on the same GPU opens it and gets a
 * read-only view: where the latest results are (mapn_ipc_latest), an asynchronous copy of the first
 * num_copied positions of a buffer on ITS stream, queued behind a GPU-side wait for the compute
 * fence to reach wait_fence_value (Render.cpp:796,814; 0 = no wait), and Signal(consumerFence,
 * value) ordered on its stream (Render.cpp:826), which is what mapn_simulate(wait_value) of the
 * exporting process waits for (Compute.cpp:1012).
 * Exporting attaches the consumer (mapn_set_consumer(1)) and makes every step publish
 * {fence value, buffer index} to the status block (one extra one-lane launch per step).
 */
#define MAPN_IPC_BLOB_BYTES 256
typedef struct mapn_ipc_view mapn_ipc_view;
int mapn_ipc_export(mapn_ctx *ctx, void *out_blob);
int mapn_ipc_open(const void *blob, int device, mapn_ipc_view **out_view);
int mapn_ipc_close(mapn_ipc_view *view);
/* fence value signalled by the latest published step and the buffer holding its positions */
int mapn_ipc_latest(mapn_ipc_view *view, uint64_t *fence_value, uint32_t *buffer_index);
void *mapn_ipc_positions(mapn_ipc_view *view, uint32_t buffer_index);
int mapn_ipc_copy_positions_async(mapn_ipc_view *view, uint32_t buffer_index, uint32_t num_copied, void *dst,
                                  uint64_t wait_fence_value, void *consumer_stream);
int mapn_ipc_consumer_signal(mapn_ipc_view *view, uint64_t value, void *consumer_stream);

/*
 * On-disk snapshot (the reference has only the in-memory CopyState, Compute.cpp:303-410):
 * little-endian, 32-byte header {"MAPNSNAP", u32 version = 1, u32 N, u32 buffer_index,
 * u32 reserved, u64 fence_value}, then for buffer 0 and 1: float4[N] positions, float3[N]
 * velocities.  Both ping-pong buffers are stored so that a restored context continues
 * bit-identically (including bodies frozen by num_active < N).
 */
int mapn_save_snapshot(mapn_ctx *ctx, const char *path);
int mapn_load_snapshot(mapn_ctx *ctx, const char *path);

/*
 * LoadParticles / InitializeParticles (Compute.cpp:667-812, 820-844), made deterministic:
 * host-side generator, no device needed.  See csrc/mapn_init.cpp for the specification.
 */
int mapn_generate_initial_state(uint32_t seed, uint32_t num_particles, float spread,
                                float initial_speed, float *pos4, float *vel3);
int mapn_generate_initial_state_ex(int init_variant, uint32_t seed, uint32_t num_particles, float spread,
                                   float initial_speed, float *pos4, float *vel3);

/* The 32-byte constant block of Compute.cpp:542-546 as this context would upload it. */
int mapn_get_cbuffer(const mapn_ctx *ctx, uint32_t out_param[4], float out_paramf[4]);

/* ---- sharded (multi-GPU) mode: one process per GPU, bodies [p*N/P, (p+1)*N/P) per rank ---- */

#define MAPN_UNIQUE_ID_BYTES 128
/* rank 0 creates the id (ncclGetUniqueId), the launcher broadcasts it to all ranks */
int mapn_comm_get_unique_id(void *out_id128);
/* collective over all ranks of the job; afterwards every mapn_simulate ends with an
 * all-gather of the new position slices over RCCL/xGMI on the context's comm stream */
int mapn_comm_init(mapn_ctx *ctx, const void *id128);
/* how the native exchange is issued: 0 = ncclAllGather (default), 1 = one group of ncclSend /
 * ncclRecv pairs (a single direct xGMI hop per peer instead of a ring), 2 = the direct
 * peer-to-peer exchange below (after mapn_p2p_import), 3 = the same peer-to-peer exchange
 * overlapped INSIDE the force launch ("flow" mode: the pull runs beside the launch on the comm
 * stream, the launch starts on its own slice and its remote chunks wait for each peer's arrival
 * flag, the last integrated tile publishes to the peers -- no separate exchange step, no
 * cross-stream event), 4 = the SYMMETRIC step sharded over the ranks (every unordered pair evaluated
 * once in the whole job: a rank runs the meetings of its own 1024-body blocks, stores the reactions it
 * produced for another rank's bodies -- summed over its blocks first, one row per destination -- straight
 * into that rank's receive region, and integrates its bodies from its own rows plus the rows received;
 * the same launch then publishes the rank's new slice and pulls the peers'; needs N / world_size to be a multiple of
 * 1024, otherwise the step runs as 2.  A PARTIALLY ACTIVE step (num_active < N, at least 2048 bodies active) keeps the symmetric kernel
 * since round 6: the active bodies form a ring of blocks of their own which their owners run, and every rank that owns FROZEN bodies
 * computes what those do to all active bodies -- one one-sided launch -- and sends the sums in the same rows as the reactions
 * (mapn_shard_split_describe, mapn_tuning.h, says who does what; before: the one-sided kernel over (a rank's active bodies) x N and a pull)),
 * 5 = as 4, but that launch also STORES the new positions
 * into every peer's replica (posted writes instead of read round trips) and the peers' next force launch waits for
 * this rank's counter before it reads them, 6 = the sharded symmetric step over RCCL alone (after mapn_comm_init, no
 * mapped peer memory): a pack launch, one group of ncclSend / ncclRecv carrying the per-destination reaction rows into
 * the same [sender][body] layout, a reduce launch that adds them in the same fixed order, then ncclAllGather of the
 * positions as in 0; all ranks must agree.  4, 5 and 6 allocate the sharded symmetric step's
 * scratch here (never inside mapn_simulate); under MAPN_KERNEL_AUTO a failed allocation only means the steps run as 2. */
int mapn_set_gather_algorithm(mapn_ctx *ctx, int algorithm);
/*
 * Direct peer-to-peer exchange (algorithm 2 of mapn_set_gather_algorithm), no collective library:
 * every rank exports a blob describing its position heap and its flag array (hipIpc handles),
 * the launcher all-gathers the blobs (rank order) and every rank imports them.  Afterwards each
 * step ends with one small kernel that publishes a per-peer flag, waits for the peers' flags and
 * pulls their slices over xGMI (csrc/mapn_kernels.hip, p2p_gather_kernel).  Device-side waits are
 * bounded (mapn_set_timeouts, default 2 s): a wait that timed out makes the next mapn_simulate /
 * mapn_wait_idle / mapn_download_* return MAPN_ERR_COMM naming the peer; mapn_p2p_status() != 0
 * reports the same without failing (peer q = status - 1).
 * One rank per GPU is the intended use.  The blob also carries the exporting GPU's PCI id: ranks found to SHARE a device (the
 * multi-process tests on a one-GPU box) size every launch that waits for a peer so that all of them fit the device together
 * (processes have separate hardware queues; their launches run side by side, not in turns) and keep the equal-wave 4-wave
 * plan of the symmetric kernel.
 */
#define MAPN_P2P_BLOB_BYTES 192
int mapn_p2p_export(mapn_ctx *ctx, void *out_blob);
int mapn_p2p_import(mapn_ctx *ctx, const void *blobs, int count);
int mapn_p2p_status(mapn_ctx *ctx);
/* alternative transport: the caller all-gathers the written position buffer itself after every
 * step (e.g. torch.distributed.all_gather_into_tensor on the exported buffers) */
int mapn_set_external_gather(mapn_ctx *ctx, int enabled);
/*
 * Sharded mode: every rank keeps a full replica of both position buffers, and after any correct exchange the replicas are
 * bit-identical on all ranks.  out[b] = the sum of the 4 N 32-bit words of position buffer b as a 64-bit integer (drains the
 * context first, like mapn_download_buffer: a timed-out or failed device-side wait is reported here).  The launcher compares
 * the two numbers across ranks (bench.py does after every trial and after the timed run); equal sums on all ranks do not prove
 * the exchange delivered the RIGHT data -- the pushed positions of gather algorithm 5 are therefore also checked on the device,
 * every step, against checksums their pusher stores behind them (a mismatch makes the next mapn_simulate / mapn_wait_idle /
 * mapn_download_* fail with MAPN_ERR_COMM naming the pusher) -- the analogue of the reference's fence protocol between the
 * two adapters (Compute.cpp:1012, Render.cpp:796-826), which has no data check at all.
 * COLLECTIVE in effect: every rank must have drained its own work (mapn_wait_idle) and all ranks must have met at a barrier of the
 * launcher, at the same step, before any of them calls this -- buffer_index names the buffer the NEXT step writes, and a peer that has
 * already enqueued that step stores its new slice into it while it is being summed here (a torn sum, a false "replicas differ").
 * bench.py barriers first.
 */
int mapn_replica_checksum(mapn_ctx *ctx, uint64_t out[2]);
/* the slice [first, first+count) of bodies this context owns */
int mapn_shard_range(const mapn_ctx *ctx, uint32_t *first, uint32_t *count);

/* ---- the device (replaces the adapter enumeration of Particles.cpp:96-123) ---- */
typedef struct mapn_device_info {
    char     name[128];
    char     arch[64];
    int32_t  compute_units;
    int32_t  clock_khz;
    int32_t  wavefront_size;
    int32_t  reserved;
    double   peak_fp32_flops;    /* CUs x clock x 256 flop/clk/CU */
    uint64_t total_memory_bytes;
} mapn_device_info;
int mapn_get_device_info(int device, mapn_device_info *out);
int mapn_device_count(void);

/* The compute stream (hipStream_t) steps are enqueued on, for callers that record events. */
void *mapn_compute_stream(mapn_ctx *ctx);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MAPN_H */
