// consumer_fence.cpp -- the RENDER side of Particles::Draw (Particles.cpp:446-448 +
// Render::CopySimulationResults, Render.cpp:789-831) on real HIP streams and events, against
// compat/Compute.hpp.  Checks the two boundary behaviours VERDICT r1 / ADVICE r1 flagged:
//   * the exported step-done event is ONE handle fetched once (like Render::SetShared opens the
//     shared fence once) and stays valid for hundreds of steps;
//   * Simulate(n, v) queues its GPU-side wait on the consumer's fence BEFORE the consumer has
//     signalled (Compute.cpp:1012), and a slow consumer is never overrun.
// The consumer is made deliberately slow (a 64 MiB device copy in front of every position copy),
// the compute side is never host-synchronised inside the loop, and every captured copy is
// compared bit for bit with a second context stepped in lockstep without a consumer.
// Built with plain g++ (host HIP API only).  Exit code 0 = every check held.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "Compute.hpp"

using mapn::Compute;

static int g_fail = 0;
#define CHECK(cond)                                                      \
    do {                                                                 \
        const bool ok_ = (cond);                                         \
        if (!ok_) { std::printf("FAIL  %s (line %d)\n", #cond, __LINE__); g_fail++; } \
    } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("FAIL  %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// late = true: the consumer's signal for frame f - 1 is registered only AFTER Simulate of frame f was
// called, so every Simulate queues its wait before the signal exists (the deferred path);
// late = false: it is registered before (the hipStreamWaitEvent fast path).
static int run(bool late)
{
    const uint32_t n = 4096, ncopy = 1000;
    const int frames = 150;                                   // > 64: the old fence-event ring wrapped there
    mapn_config cfg;
    mapn_config_default(&cfg);
    cfg.mass = 70000.0f / n;
    Compute compute(n, 0, false, nullptr, &cfg), reference(n, 0, false, nullptr, &cfg);

    hipStream_t copyQ;
    HIP(hipStreamCreateWithFlags(&copyQ, hipStreamNonBlocking));
    const size_t slow_bytes = 64u << 20;
    void *slow_a = nullptr, *slow_b = nullptr;
    float *captured = nullptr;                                // [frames][ncopy] float4
    HIP(hipMalloc(&slow_a, slow_bytes));
    HIP(hipMalloc(&slow_b, slow_bytes));
    HIP(hipMalloc(reinterpret_cast<void **>(&captured), (size_t)frames * ncopy * 16));
    std::vector<hipEvent_t> copied(frames);
    for (auto &e : copied) HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    // Render::SetShared: fetch the handles ONCE
    const Compute::SharedHandles h = compute.GetSharedHandles(true);
    hipEvent_t stepDone = static_cast<hipEvent_t>(h.step_done_event);
    uint32_t latest = 1 - h.buffer_index;                     // both buffers hold the initial state
    compute.ConsumerSignal(compute.GetFenceValue() - 1);      // nothing to wait for before frame 0

    std::vector<Compute::Particle> want((size_t)frames * ncopy), got((size_t)frames * ncopy), tmp(n);
    std::vector<Compute::ParticleVelocity> tmpv(n);
    uint64_t prevFence = 0;
    for (int f = 0; f < frames; f++) {
        const uint64_t fence = compute.GetFenceValue();       // Particles.cpp:446
        if (!late && f > 0) compute.ConsumerSignal(prevFence, copied[f - 1]);
        compute.Simulate((int)n, fence);                      // :448 -- late: nobody has signalled fence - 1 yet, the wait is queued anyway
        if (late && f > 0) compute.ConsumerSignal(prevFence, copied[f - 1]);   // Render.cpp:826 Signal(copyFence), registered late
        latest ^= 1u;                                         // MoveToNextFrame flipped the index: results land in `latest`
        // Render::CopySimulationResults on the copy queue: wait for the compute fence, copy, signal
        HIP(hipStreamWaitEvent(copyQ, stepDone, 0));          // Render.cpp:796
        HIP(hipMemcpyAsync(slow_b, slow_a, slow_bytes, hipMemcpyDeviceToDevice, copyQ));   // a slow consumer
        HIP(hipMemcpyAsync(captured + (size_t)f * ncopy * 4, h.positions[latest], (size_t)ncopy * 16,
                           hipMemcpyDeviceToDevice, copyQ));  // Render.cpp:814
        HIP(hipEventRecord(copied[f], copyQ));
        prevFence = fence;
        // lockstep reference without a consumer, synchronously downloaded
        reference.Simulate((int)n, 0);
        reference.DownloadState(tmp.data(), tmpv.data());
        std::memcpy(&want[(size_t)f * ncopy], tmp.data(), (size_t)ncopy * 16);
    }
    compute.WaitForGpu();
    HIP(hipStreamSynchronize(copyQ));
    HIP(hipMemcpy(got.data(), captured, (size_t)frames * ncopy * 16, hipMemcpyDeviceToHost));
    int bad_frames = 0;
    for (int f = 0; f < frames; f++)
        if (std::memcmp(&got[(size_t)f * ncopy], &want[(size_t)f * ncopy], (size_t)ncopy * 16) != 0) bad_frames++;
    std::printf("%s signal: %d frames, %d captured copies differ from the lockstep reference\n", late ? "late" : "early", frames, bad_frames);
    CHECK(bad_frames == 0);
    CHECK(compute.GetFenceValue() == reference.GetFenceValue() + 1);   // + the one WaitForGpu above
    for (auto &e : copied) (void)hipEventDestroy(e);
    (void)hipFree(slow_a); (void)hipFree(slow_b); (void)hipFree(captured);
    (void)hipStreamDestroy(copyQ);
    return 0;
}

int main()
{
    if (mapn_device_count() == 0) { std::printf("no device\n"); return 1; }
    if (run(true) || run(false)) return 1;
    std::printf(g_fail ? "FAILED\n" : "ok\n");
    return g_fail;
}
