// mapn_shard.cpp -- the sharded (multi-GPU) mode of libmapn.so, one process per GPU: the RCCL communicator, the direct
// peer-to-peer set-up over hipIpc, the exchange launches of gather algorithms 0 - 3 (4 - 6: mapn_sym_host.cpp), the replica checksum.
// The reference's own two-adapter hand-off this stands in for: Compute.cpp:163-201,434-435, Render.cpp:789-831.
#include "mapn_internal.h"

using namespace mapn::host;

namespace {
struct P2PBlob {
    char magic[8];
    uint32_t rank, world, n, device_id;       // device_id: PCI domain / bus / device of the exporting rank's GPU (+1), 0 = unknown
    uint64_t aligned_data_size;
    hipIpcMemHandle_t heap, flags;
};
static_assert(sizeof(P2PBlob) <= MAPN_P2P_BLOB_BYTES, "MAPN_P2P_BLOB_BYTES too small");
}  // namespace

// which GPU a rank runs on, so that ranks SHARING one device (tests, a partitioned box) can be told from a real job
static uint32_t p2p_device_id(int device)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) { (void)hipGetLastError(); return 0u; }
    return 1u + (((uint32_t)p.pciDomainID & 0xffu) << 16 | ((uint32_t)p.pciBusID & 0xffu) << 8 | ((uint32_t)p.pciDeviceID & 0xffu));
}

namespace mapn {
namespace host {

int enqueue_p2p(mapn_ctx *c)
{
    const uint32_t w = c->buffer_index;
    mapn::P2PArgs a{};
    a.local = c->pos[w];
    for (int q = 0; q < c->cfg.world_size; q++) {
        // both position buffers live in one heap allocation: buffer w sits w * aligned_data_size in
        a.peer[q] = reinterpret_cast<const float4 *>(static_cast<char *>(c->p2p_peer_heap[q]) + (size_t)w * c->aligned_data_size);
        a.peer_flags[q] = c->p2p_peer_flags[q];
    }
    a.my_flags = c->p2p_flags;
    a.status = c->async_status;
    a.rank = (uint32_t)c->cfg.rank;
    a.world = (uint32_t)c->cfg.world_size;
    a.count = c->count;
    a.step = ++c->p2p_step;
    a.timeout_ticks = c->p2p_timeout_ticks;                // s_memrealtime ticks (100 MHz); mapn_set_timeouts
    HIP_TRY(mapn::launch_p2p_gather(a, c->compute));
    return MAPN_OK;
}

// flow mode: the pull half of this step's exchange, on the comm stream, no stream dependencies --
// it waits for the peers' flags on the device and marks arrived[q] for the NEXT force launch
int enqueue_flow_pull(mapn_ctx *c)
{
    const uint32_t w = c->buffer_index;
    mapn::P2PArgs a{};
    a.local = c->pos[w];
    for (int q = 0; q < c->cfg.world_size; q++) {
        a.peer[q] = reinterpret_cast<const float4 *>(static_cast<char *>(c->p2p_peer_heap[q]) + (size_t)w * c->aligned_data_size);
        a.peer_flags[q] = c->p2p_peer_flags[q];
    }
    a.my_flags = c->p2p_flags;
    a.status = c->async_status;
    a.rank = (uint32_t)c->cfg.rank;
    a.world = (uint32_t)c->cfg.world_size;
    a.count = c->count;
    a.step = ++c->p2p_step;
    a.timeout_ticks = c->p2p_timeout_ticks;
    HIP_TRY(mapn::launch_flow_pull(a, c->flow_block, c->comm_stream));
    return MAPN_OK;
}

int enqueue_gather(mapn_ctx *c)
{
    if (c->step_pulled) { c->step_pulled = false; return MAPN_OK; }   // sym_shard_exchange_kernel has published and pulled
    if (c->p2p_ready && c->gather_algo == 3) return enqueue_flow_pull(c);
    if (c->p2p_ready && c->p2p_loopback) return MAPN_OK;   // no peers to pull from
    if (c->p2p_ready && (c->gather_algo == 2 || c->gather_algo == 4 || c->gather_algo == 5)) return enqueue_p2p(c);   // 4 / 5 where the symmetric step does not apply: positions travel as in 2
    if (!c->comm) return MAPN_OK;
    const uint32_t w = c->buffer_index;
    const bool overlap = (c->cfg.flags & MAPN_FLAG_SHARD_OVERLAP) != 0 && c->gather_algo < 2;
    // in place: every rank's slice sits at its own offset of the full buffer.
    // Default structure: the collective is enqueued on the COMPUTE stream right behind the
    // integrate kernel -- plain stream order, no cross-stream event hops (each costs 5-10 us of
    // queue latency, which matters when a sharded step is ~0.12 ms).  Overlap structure: on the
    // comm stream, behind this step's fence event, so that the next step's own-segment launch
    // runs beside it.
    hipStream_t st = overlap ? c->comm_stream : c->compute;
    if (overlap) HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->step_done, 0));   // = this step's fence event
    const int rc_gather = c->gather_algo == 1
        ? mapn::comm_gather_sendrecv_inplace(c->comm, c->pos[w], (size_t)c->count * 4, st)
        : mapn::comm_all_gather_inplace(c->comm, c->pos[w], (size_t)c->count * 4, st);
    if (rc_gather) return fail(MAPN_ERR_COMM, "all-gather failed: %s", mapn::comm_last_error());
    if (overlap) {
        HIP_TRY(hipEventRecord(c->gather_done[w], c->comm_stream));
        c->gather_recorded[w] = true;
    }
    return MAPN_OK;
}

}  // namespace host
}  // namespace mapn

extern "C" {

int mapn_comm_get_unique_id(void *out_id128)
{
    if (!out_id128) return fail(MAPN_ERR_INVALID_ARGUMENT, "null id");
    if (mapn::comm_get_unique_id(out_id128)) return fail(MAPN_ERR_COMM, "%s", mapn::comm_last_error());
    return MAPN_OK;
}

int mapn_comm_init(mapn_ctx *c, const void *id128)
{
    if (!c || !id128) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    if (c->adopted) return fail(MAPN_ERR_STATE, "comm_init: context is in adopted-buffer (async) mode");
    HIP_TRY(hipSetDevice(c->device));
    if (c->comm) return MAPN_OK;
    // MAPN_COMM_LOOPBACK=1 (timing experiments on a 1-GPU box only): rank 0 of a P-way sharded
    // context joins a ONE-rank communicator, so the step runs its real structure (own-segment
    // kernel, remote-segment kernel, reduce, ncclAllGather launch) at the true shard size; the
    // other ranks' slices are then never refreshed, so results are not a simulation.
    const char *loop = test_hook("MAPN_COMM_LOOPBACK");
    if (loop && loop[0] == '1' && c->cfg.rank == 0)
        c->comm = mapn::comm_create(id128, 0, 1);
    else
        c->comm = mapn::comm_create(id128, c->cfg.rank, c->cfg.world_size);
    if (!c->comm) return fail(MAPN_ERR_COMM, "%s", mapn::comm_last_error());
    return MAPN_OK;
}

int mapn_set_gather_algorithm(mapn_ctx *c, int algorithm)
{
    if (!c || algorithm < 0 || algorithm > 6) return fail(MAPN_ERR_INVALID_ARGUMENT, "set_gather_algorithm: bad argument");
    if (algorithm >= 2 && algorithm != 6 && !c->p2p_ready) return fail(MAPN_ERR_STATE, "set_gather_algorithm(%d): call mapn_p2p_import first", algorithm);
    if ((algorithm < 2 || algorithm == 6) && c->cfg.world_size > 1 && !c->comm && !c->external_gather)
        return fail(MAPN_ERR_STATE, "set_gather_algorithm(%d): no RCCL communicator (mapn_comm_init)", algorithm);
    if (int rc = mapn_wait_idle(c)) return rc;
    if (algorithm == 3) {
        // every exchange so far has completed (wait_idle): both replicas are whole, so every peer's
        // slice counts as arrived for the exchange number reached
        HIP_TRY(hipSetDevice(c->device));
        uint32_t arrived[17];
        for (int q = 0; q < 16; q++) arrived[q] = c->p2p_step;
        arrived[16] = 0;                                   // tiles_done
        HIP_TRY(hipMemcpy(c->flow_block, arrived, sizeof arrived, hipMemcpyHostToDevice));
    }
    // algorithm 4: plan and scratch of the sharded symmetric step are made HERE (never inside mapn_simulate); if they
    // cannot be had under MAPN_KERNEL_AUTO the step runs as algorithm 2 (one-sided kernel + peer-to-peer pull)
    if (algorithm >= 4 && algorithm <= 6) {
        if (!(c->sym_ready && c->sym_sharded)) {
            if (int rc = calibrate_for_shard(c)) return rc;            // (MAPN_FLAG_XCD_CALIBRATE only: die weights from a temporary unsharded context)
            if (int rc = prepare_sym(c, true)) return rc;
        }
    }
    if (algorithm == 6 && c->sym_ready && !c->sym_send) {
        // send / receive rows of the RCCL form, and who exchanges with whom (mapn_p2p_import computes the same masks for 4 / 5)
        HIP_TRY(hipSetDevice(c->device));
        const size_t bytes = (size_t)c->cfg.world_size * c->count * sizeof(float4);
        if (hipMalloc(&c->sym_send, bytes) != hipSuccess || hipMalloc(&c->sym_recv, bytes) != hipSuccess) {
            (void)hipGetLastError();
            if (c->sym_send) (void)hipFree(c->sym_send);
            c->sym_send = c->sym_recv = nullptr;
            if (c->cfg.kernel == MAPN_KERNEL_SYMMETRIC) return fail(MAPN_ERR_HIP, "set_gather_algorithm(6): the reaction rows could not be allocated");
        } else {
            HIP_TRY(hipMemset(c->sym_recv, 0, bytes));
            sym_shard_masks(c->n / mapn::SYM_BLOCK, (uint32_t)c->cfg.world_size, (uint32_t)c->cfg.rank, c->sym_send_mask, c->sym_recv_mask);
            const char *loop = test_hook("MAPN_COMM_LOOPBACK");                     // a 1-rank communicator: nobody to exchange with
            if (loop && loop[0] == '1') { c->sym_send_mask &= 1u << c->cfg.rank; c->sym_recv_mask = 1u << c->cfg.rank; }
        }
    }
    if (algorithm < 4 && c->sym_sharded) release_sym(c);               // the other algorithms run the one-sided kernels: give the scratch back
    c->gather_algo = algorithm;
    return MAPN_OK;
}

int mapn_p2p_export(mapn_ctx *c, void *out_blob)
{
    if (!c || !out_blob) return fail(MAPN_ERR_INVALID_ARGUMENT, "p2p_export: null argument");
    if (c->cfg.world_size < 2 || c->cfg.world_size > mapn::P2P_MAX_RANKS)
        return fail(MAPN_ERR_STATE, "p2p_export: world_size %d (2..%d supported)", c->cfg.world_size, mapn::P2P_MAX_RANKS);
    if (c->adopted) return fail(MAPN_ERR_STATE, "p2p_export: context is in adopted-buffer (async) mode");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->p2p_flags) {
        // publication counters: uncached device memory, so that a peer's store over xGMI and this
        // GPU's polling loads meet in memory, never in a cache
        // Behind the counters (same allocation, same hipIpc handle): the receive region of the sharded symmetric
        // step, one float4 row per sender rank and body of this rank -- peers store into it, this GPU reads it.
        // ... and behind that the arrival flags of the reaction rows, one word per sender and 256-body chunk
        // ... and behind those the checksums of pushed positions, one word per sender and 32 bodies (mapn_kernels.h: sym_region_*)
        const size_t bytes = mapn::sym_region_bytes((uint32_t)c->cfg.world_size, c->count);
        HIP_TRY(hipExtMallocWithFlags(reinterpret_cast<void **>(&c->p2p_flags), bytes, hipDeviceMallocUncached));
        HIP_TRY(hipMemset(c->p2p_flags, 0, bytes));
        HIP_TRY(hipDeviceSynchronize());
    }
    P2PBlob b{};
    memcpy(b.magic, "MAPNP2P1", 8);
    b.rank = (uint32_t)c->cfg.rank; b.world = (uint32_t)c->cfg.world_size; b.n = c->n;
    b.device_id = p2p_device_id(c->device);
    b.aligned_data_size = c->aligned_data_size;
    HIP_TRY(hipIpcGetMemHandle(&b.heap, c->pos_heap));
    HIP_TRY(hipIpcGetMemHandle(&b.flags, c->p2p_flags));
    memset(out_blob, 0, MAPN_P2P_BLOB_BYTES);
    memcpy(out_blob, &b, sizeof b);
    return MAPN_OK;
}

int mapn_p2p_import(mapn_ctx *c, const void *blobs, int count)
{
    if (!c || !blobs) return fail(MAPN_ERR_INVALID_ARGUMENT, "p2p_import: null argument");
    if (count != c->cfg.world_size) return fail(MAPN_ERR_INVALID_ARGUMENT, "p2p_import: %d blobs for world_size %d", count, c->cfg.world_size);
    if (!c->p2p_flags) return fail(MAPN_ERR_STATE, "p2p_import: call mapn_p2p_export first");
    if (c->p2p_ready) return MAPN_OK;
    HIP_TRY(hipSetDevice(c->device));
    // MAPN_P2P_LOOPBACK=1 (timing experiments on a 1-GPU box only, like MAPN_COMM_LOOPBACK): rank 0 of a P-way
    // job maps every peer to ITSELF, so a step runs its real kernels at the true shard size (force, send with
    // all its destinations, reduce); the position pull is skipped and only this rank's own row is waited for --
    // the other slices are never refreshed, so results are not a simulation.
    const char *loop = test_hook("MAPN_P2P_LOOPBACK");
    // MAPN_P2P_LOOPBACK=2 (tests): the same, but nothing is SENT to the other ranks either (their rows would land on this rank's
    // own), so this rank's bodies come out exactly as the schedule says: own meetings plus reactions between own blocks.
    c->p2p_loopback = loop && (loop[0] == '1' || loop[0] == '2');
    for (int q = 0; q < count; q++) {
        if (c->p2p_loopback) { c->p2p_peer_heap[q] = c->pos_heap; c->p2p_peer_flags[q] = c->p2p_flags; continue; }
        P2PBlob b;
        memcpy(&b, static_cast<const char *>(blobs) + (size_t)q * MAPN_P2P_BLOB_BYTES, sizeof b);
        if (memcmp(b.magic, "MAPNP2P1", 8) != 0 || (int)b.rank != q || (int)b.world != count || b.n != c->n ||
            b.aligned_data_size != c->aligned_data_size)
            return fail(MAPN_ERR_INVALID_ARGUMENT, "p2p_import: blob %d does not describe rank %d of this job", q, q);
        if (q == c->cfg.rank) {
            c->p2p_peer_heap[q] = c->pos_heap;
            c->p2p_peer_flags[q] = c->p2p_flags;
            continue;
        }
        if (b.device_id && b.device_id == p2p_device_id(c->device)) { c->p2p_shared_device = true; c->p2p_ranks_on_device++; }
        HIP_TRY(hipIpcOpenMemHandle(&c->p2p_peer_heap[q], b.heap, hipIpcMemLazyEnablePeerAccess));
        HIP_TRY(hipIpcOpenMemHandle(reinterpret_cast<void **>(&c->p2p_peer_flags[q]), b.flags, hipIpcMemLazyEnablePeerAccess));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->flow_block), 256));
    HIP_TRY(hipMemset(c->flow_block, 0, 256));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->sym_shard_ticket), 256));
    HIP_TRY(hipMemset(c->sym_shard_ticket, 0, 256));
    if (c->count % mapn::SYM_BLOCK == 0 && c->count * (uint32_t)count == c->n)
        sym_shard_masks(c->n / mapn::SYM_BLOCK, (uint32_t)count, (uint32_t)c->cfg.rank, c->sym_send_mask, c->sym_recv_mask);
    if (c->p2p_loopback) c->sym_recv_mask = 1u << c->cfg.rank;
    if (c->p2p_loopback && loop[0] == '2') c->sym_send_mask &= 1u << c->cfg.rank;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->p2p_flag_table), sizeof(uint32_t *) * mapn::P2P_MAX_RANKS));
    HIP_TRY(hipMemcpy(c->p2p_flag_table, c->p2p_peer_flags, sizeof(uint32_t *) * mapn::P2P_MAX_RANKS, hipMemcpyHostToDevice));
    c->p2p_ready = true;
    return MAPN_OK;
}

int mapn_p2p_status(mapn_ctx *c)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    if (!c->async_status) return 0;
    return (int)reinterpret_cast<volatile uint32_t *>(c->async_status)[0];
}

int mapn_replica_checksum(mapn_ctx *c, uint64_t out[2])
{
    if (!c || !out) return fail(MAPN_ERR_INVALID_ARGUMENT, "replica_checksum: null argument");
    // (both buffers, the one the NEXT step writes included: a peer that has already enqueued that step may be pushing into it while it
    //  is summed -- the header makes the barrier the caller's duty: every rank drained, all at the same step)
    std::vector<uint32_t> host((size_t)c->n * 4);
    for (uint32_t b = 0; b < 2; b++) {
        if (int rc = mapn_download_buffer(c, b, reinterpret_cast<float *>(host.data()), nullptr)) return rc;
        uint64_t s = 0;
        for (uint32_t w : host) s += w;
        out[b] = s;
    }
    return MAPN_OK;
}

// The sharded mode's host arithmetic WITHOUT a device: the same helpers mapn_create and the step use (shard_slice,
// shard_active_count, active_bodies, sym_shard_masks) -- what the CPU multi-process test composes a sharded run from.
int mapn_shard_describe(uint32_t num_particles, int32_t rank, int32_t world_size, int32_t num_active, mapn_shard_info *out)
{
    if (!out) return fail(MAPN_ERR_INVALID_ARGUMENT, "shard_describe: null argument");
    memset(out, 0, sizeof *out);
    if (num_particles == 0 || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "rank %d / world_size %d", rank, world_size);
    if (num_particles % (uint32_t)world_size)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "world_size %d must divide num_particles %u", world_size, num_particles);
    shard_slice(num_particles, (uint32_t)rank, (uint32_t)world_size, out->first, out->count);
    const uint32_t active = active_bodies(num_active, num_particles);
    out->active_first = out->first;
    out->active_count = shard_active_count(out->first, out->count, active);
    out->sym_applies = (world_size >= 2 && world_size <= mapn::P2P_MAX_RANKS && out->count % mapn::SYM_BLOCK == 0u) ? 1u : 0u;
    if (out->sym_applies) {
        out->nb = num_particles / mapn::SYM_BLOCK; out->nbl = out->count / mapn::SYM_BLOCK; out->a0 = (uint32_t)rank * out->nbl;
        sym_shard_masks(out->nb, (uint32_t)world_size, (uint32_t)rank, out->send_mask, out->recv_mask);
    }
    return MAPN_OK;
}

int mapn_shard_split_describe(uint32_t num_particles, int32_t rank, int32_t world_size, int32_t num_active, mapn_shard_split_info *out)
{
    if (!out) return fail(MAPN_ERR_INVALID_ARGUMENT, "shard_split_describe: null argument");
    memset(out, 0, sizeof *out);
    if (num_particles == 0 || world_size < 2 || world_size > mapn::P2P_MAX_RANKS || rank < 0 || rank >= world_size)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "rank %d / world_size %d", rank, world_size);
    if (num_particles % (uint32_t)world_size || (num_particles / (uint32_t)world_size) % mapn::SYM_BLOCK)
        return fail(MAPN_ERR_INVALID_ARGUMENT, "a rank's slice must be whole 1024-body blocks (%u bodies over %d ranks)", num_particles, world_size);
    const uint32_t active = active_bodies(num_active, num_particles);
    out->active = active;
    out->applies = (active < num_particles && active >= 2u * mapn::SYM_BLOCK) ? 1u : 0u;      // (sym_shard_split_eligible's part that depends on the shape)
    if (!out->applies) return MAPN_OK;
    const ShardSplit r = shard_split_describe(num_particles, (uint32_t)world_size, (uint32_t)rank, active);
    out->ring_blocks = r.nba; out->blocks = r.nbl; out->first_block = r.a0; out->active_count = r.ac;
    out->frozen_first = r.fz_first; out->frozen_count = r.fz_count; out->send_mask = r.send_mask; out->recv_mask = r.recv_mask;
    return MAPN_OK;
}

int mapn_set_external_gather(mapn_ctx *c, int enabled)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    c->external_gather = enabled != 0;
    return MAPN_OK;
}

int mapn_shard_range(const mapn_ctx *c, uint32_t *first, uint32_t *count)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    if (first) *first = c->first;
    if (count) *count = c->count;
    return MAPN_OK;
}

int mapn_set_shard_overlap(mapn_ctx *c, int enabled)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    if (int rc = mapn_wait_idle(c)) return rc;
    if (enabled) c->cfg.flags |= MAPN_FLAG_SHARD_OVERLAP; else c->cfg.flags &= ~MAPN_FLAG_SHARD_OVERLAP;
    c->gather_recorded[0] = c->gather_recorded[1] = false;
    return MAPN_OK;
}

}  // extern "C"
