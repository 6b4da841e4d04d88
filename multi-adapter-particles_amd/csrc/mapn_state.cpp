// mapn_state.cpp -- state hand-off of libmapn.so: upload / download (InitializeParticles' uploads, Compute.cpp:881-904), the on-disk
// snapshot (the reference has only the in-memory CopyState, Compute.cpp:303-410), the consumer's copy (Render::CopySimulationResults,
// Render.cpp:789-831) and the same hand-off across a process boundary (mapn_ipc_*: Compute.cpp:944-950, Render.cpp:222-251).
#include "mapn_internal.h"

using namespace mapn::host;

namespace {
struct SnapshotHeader {
    char magic[8];
    uint32_t version, n, buffer_index, reserved;
    uint64_t fence_value;
};
static_assert(sizeof(SnapshotHeader) == 32, "snapshot header is 32 bytes");
}  // namespace

namespace {
struct IpcBlob {
    char magic[8];
    uint32_t n, device;
    uint64_t aligned_data_size;
    hipIpcMemHandle_t heap, block;
};
static_assert(sizeof(IpcBlob) <= MAPN_IPC_BLOB_BYTES, "MAPN_IPC_BLOB_BYTES too small");
}  // namespace

struct mapn_ipc_view {
    int device = 0;
    uint32_t n = 0;
    uint64_t aligned_data_size = 0;
    void *heap = nullptr;
    uint32_t *block = nullptr;
    uint32_t *status = nullptr;               // pinned host word: a bounded device-side wait gave up
    uint64_t timeout_ticks = 1000ull * 1000ull * 1000ull;   // 10 s
};

extern "C" {

int mapn_upload_state(mapn_ctx *c, const float *pos4, const float *vel3)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = settle_push(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->compute));
    HIP_TRY(hipStreamSynchronize(c->comm_stream));
    for (int b = 0; b < 2; b++) {                                      // Compute.cpp:881-882,903-904
        if (pos4) HIP_TRY(hipMemcpy(c->pos[b], pos4, (size_t)c->n * 16, hipMemcpyHostToDevice));
        if (vel3) HIP_TRY(hipMemcpy(c->vel[b], vel3, (size_t)c->n * 12, hipMemcpyHostToDevice));
    }
    c->gather_recorded[0] = c->gather_recorded[1] = false;
    return MAPN_OK;
}

int mapn_download_buffer(mapn_ctx *c, uint32_t index, float *pos4, float *vel3)
{
    if (!c || index > 1) return fail(MAPN_ERR_INVALID_ARGUMENT, "download_buffer: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = settle_push(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->compute));
    HIP_TRY(hipStreamSynchronize(c->comm_stream));
    if (int rc = check_async_errors(c)) return rc;
    if (pos4) HIP_TRY(hipMemcpy(pos4, c->pos[index], (size_t)c->n * 16, hipMemcpyDeviceToHost));
    if (vel3) HIP_TRY(hipMemcpy(vel3, c->vel[index], (size_t)c->n * 12, hipMemcpyDeviceToHost));
    return MAPN_OK;
}

int mapn_download_state(mapn_ctx *c, float *pos4, float *vel3)
{
    if (!c) return fail(MAPN_ERR_INVALID_ARGUMENT, "null context");
    return mapn_download_buffer(c, 1 - c->buffer_index, pos4, vel3);
}

int mapn_copy_positions_async(mapn_ctx *c, uint32_t num_copied, void *dst, void *consumer_stream)
{
    if (!c || !dst) return fail(MAPN_ERR_INVALID_ARGUMENT, "copy_positions_async: null argument");
    if (num_copied > c->n) num_copied = c->n;
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = observe_steps(c)) return rc;
    hipStream_t st = static_cast<hipStream_t>(consumer_stream);
    // Render.cpp:796,814: copyQueue.Wait(computeFence, v); CopyBufferRegion(dst, shared[1 - idx], nCopy * 16)
    HIP_TRY(hipStreamWaitEvent(st, c->exported_done, 0));
    if (num_copied)
        HIP_TRY(hipMemcpyAsync(dst, c->pos[1 - c->buffer_index], (size_t)num_copied * 16, hipMemcpyDefault, st));
    return MAPN_OK;
}

int mapn_save_snapshot(mapn_ctx *c, const char *path)
{
    if (!c || !path) return fail(MAPN_ERR_INVALID_ARGUMENT, "save_snapshot: null argument");
    std::vector<float> pos((size_t)c->n * 4), vel((size_t)c->n * 3);
    FILE *f = fopen(path, "wb");
    if (!f) return fail(MAPN_ERR_INVALID_ARGUMENT, "save_snapshot: cannot open %s", path);
    SnapshotHeader h{};
    memcpy(h.magic, "MAPNSNAP", 8);
    h.version = 1; h.n = c->n; h.buffer_index = c->buffer_index; h.fence_value = c->fence_value;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1;
    for (uint32_t b = 0; b < 2 && ok; b++) {
        if (int rc = mapn_download_buffer(c, b, pos.data(), vel.data())) { fclose(f); return rc; }
        ok = fwrite(pos.data(), 16, c->n, f) == c->n && fwrite(vel.data(), 12, c->n, f) == c->n;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? MAPN_OK : fail(MAPN_ERR_INVALID_ARGUMENT, "save_snapshot: short write to %s", path);
}

int mapn_load_snapshot(mapn_ctx *c, const char *path)
{
    if (!c || !path) return fail(MAPN_ERR_INVALID_ARGUMENT, "load_snapshot: null argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MAPN_ERR_INVALID_ARGUMENT, "load_snapshot: cannot open %s", path);
    SnapshotHeader h{};
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "MAPNSNAP", 8) != 0 || h.version != 1) {
        fclose(f);
        return fail(MAPN_ERR_INVALID_ARGUMENT, "load_snapshot: %s is not a version-1 mapn snapshot", path);
    }
    if (h.n != c->n || h.buffer_index > 1) {
        fclose(f);
        return fail(MAPN_ERR_INVALID_ARGUMENT, "load_snapshot: snapshot holds %u bodies, context %u", h.n, c->n);
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->compute));
    HIP_TRY(hipStreamSynchronize(c->comm_stream));
    std::vector<float> pos((size_t)c->n * 4), vel((size_t)c->n * 3);
    for (uint32_t b = 0; b < 2; b++) {
        if (fread(pos.data(), 16, c->n, f) != c->n || fread(vel.data(), 12, c->n, f) != c->n) {
            fclose(f);
            return fail(MAPN_ERR_INVALID_ARGUMENT, "load_snapshot: %s is truncated", path);
        }
        HIP_TRY(hipMemcpy(c->pos[b], pos.data(), (size_t)c->n * 16, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->vel[b], vel.data(), (size_t)c->n * 12, hipMemcpyHostToDevice));
    }
    fclose(f);
    c->buffer_index = h.buffer_index;
    c->gather_recorded[0] = c->gather_recorded[1] = false;
    return MAPN_OK;
}

int mapn_get_cbuffer(const mapn_ctx *c, uint32_t out_param[4], float out_paramf[4])
{
    if (!c || !out_param || !out_paramf) return fail(MAPN_ERR_INVALID_ARGUMENT, "null argument");
    out_param[0] = c->n;                                               // Compute.cpp:543
    out_param[1] = (c->n + kBlock - 1) / kBlock;                       // Compute.cpp:544
    out_param[2] = out_param[3] = 0;
    out_paramf[0] = c->cfg.dt;                                         // Compute.cpp:545
    out_paramf[1] = c->cfg.damping;                                    // Compute.cpp:546
    out_paramf[2] = out_paramf[3] = 0.f;
    return MAPN_OK;
}

int mapn_ipc_export(mapn_ctx *c, void *out_blob)
{
    if (!c || !out_blob) return fail(MAPN_ERR_INVALID_ARGUMENT, "ipc_export: null argument");
    if (c->adopted) return fail(MAPN_ERR_STATE, "ipc_export: context computes into adopted buffers it does not own");
    HIP_TRY(hipSetDevice(c->device));
    IpcBlob b{};
    memcpy(b.magic, "MAPNIPC1", 8);
    b.n = c->n; b.device = (uint32_t)c->device; b.aligned_data_size = c->aligned_data_size;
    HIP_TRY(hipIpcGetMemHandle(&b.heap, c->pos_heap));
    HIP_TRY(hipIpcGetMemHandle(&b.block, c->fence_dev_block));
    c->ipc_exported = true;
    c->consumer_enabled = true;                            // the importer's fence is attached (GetSharedHandles(renderFence))
    if (int rc = observe_steps(c)) return rc;
    if (int rc = publish_ipc_status(c, c->fence_value - 1, 1 - c->buffer_index)) return rc;
    HIP_TRY(hipStreamSynchronize(c->compute));
    memset(out_blob, 0, MAPN_IPC_BLOB_BYTES);
    memcpy(out_blob, &b, sizeof b);
    return MAPN_OK;
}

int mapn_ipc_open(const void *blob, int device, mapn_ipc_view **out_view)
{
    if (!blob || !out_view) return fail(MAPN_ERR_INVALID_ARGUMENT, "ipc_open: null argument");
    *out_view = nullptr;
    IpcBlob b;
    memcpy(&b, blob, sizeof b);
    if (memcmp(b.magic, "MAPNIPC1", 8) != 0) return fail(MAPN_ERR_INVALID_ARGUMENT, "ipc_open: not a mapn ipc blob");
    HIP_TRY(hipSetDevice(device));
    mapn_ipc_view *v = new mapn_ipc_view();
    v->device = device; v->n = b.n; v->aligned_data_size = b.aligned_data_size;
    hipError_t e = hipIpcOpenMemHandle(&v->heap, b.heap, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) e = hipIpcOpenMemHandle(reinterpret_cast<void **>(&v->block), b.block, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&v->status), 64, hipHostMallocMapped);
    if (e == hipSuccess) memset(v->status, 0, 64);
    if (e != hipSuccess) {
        std::string msg = hipGetErrorString(e);
        mapn_ipc_close(v);
        return fail(MAPN_ERR_HIP, "ipc_open: %s", msg.c_str());
    }
    *out_view = v;
    return MAPN_OK;
}

int mapn_ipc_close(mapn_ipc_view *v)
{
    if (!v) return MAPN_OK;
    (void)hipSetDevice(v->device);
    if (v->status) (void)hipHostFree(v->status);
    if (v->block) (void)hipIpcCloseMemHandle(v->block);
    if (v->heap) (void)hipIpcCloseMemHandle(v->heap);
    delete v;
    return MAPN_OK;
}

int mapn_ipc_latest(mapn_ipc_view *v, uint64_t *fence_value, uint32_t *buffer_index)
{
    if (!v) return fail(MAPN_ERR_INVALID_ARGUMENT, "null view");
    HIP_TRY(hipSetDevice(v->device));
    uint32_t w[2] = {0, 0};
    HIP_TRY(hipMemcpy(w, v->block + 16, sizeof w, hipMemcpyDeviceToHost));
    if (fence_value) *fence_value = w[0];
    if (buffer_index) *buffer_index = w[1];
    return MAPN_OK;
}

void *mapn_ipc_positions(mapn_ipc_view *v, uint32_t buffer_index)
{
    if (!v || buffer_index > 1) return nullptr;
    return static_cast<char *>(v->heap) + (size_t)buffer_index * v->aligned_data_size;
}

int mapn_ipc_copy_positions_async(mapn_ipc_view *v, uint32_t buffer_index, uint32_t num_copied, void *dst,
                                  uint64_t wait_fence_value, void *consumer_stream)
{
    if (!v || !dst || buffer_index > 1) return fail(MAPN_ERR_INVALID_ARGUMENT, "ipc_copy_positions_async: bad argument");
    if (*reinterpret_cast<volatile uint32_t *>(v->status))
        return fail(MAPN_ERR_STATE, "ipc view: an earlier wait for the compute fence timed out");
    if (num_copied > v->n) num_copied = v->n;
    HIP_TRY(hipSetDevice(v->device));
    hipStream_t st = static_cast<hipStream_t>(consumer_stream);
    // Render.cpp:796 copyQueue.Wait(computeFence, v): the compute fence across the process boundary
    // is the status block's fence word, published on the compute stream behind each step
    if (wait_fence_value)
        HIP_TRY(mapn::launch_fence_wait(v->block + 16, v->block + 16, (uint32_t)wait_fence_value, v->timeout_ticks, v->status, st));
    if (num_copied)
        HIP_TRY(hipMemcpyAsync(dst, mapn_ipc_positions(v, buffer_index), (size_t)num_copied * 16, hipMemcpyDefault, st));
    return MAPN_OK;
}

int mapn_ipc_consumer_signal(mapn_ipc_view *v, uint64_t value, void *consumer_stream)
{
    if (!v) return fail(MAPN_ERR_INVALID_ARGUMENT, "null view");
    HIP_TRY(hipSetDevice(v->device));
    // Render.cpp:826 copyQueue.Signal(copyFence, value): ordered behind the consumer's copies
    HIP_TRY(mapn::launch_fence_signal(v->block, (uint32_t)value, static_cast<hipStream_t>(consumer_stream)));
    return MAPN_OK;
}

}  // extern "C"
