// Compute.hpp -- C++ host-side mirror of the reference's `class Compute`
// (reference/Particles/Compute.h:33-78 + AdapterShared.h:51-60) over the C ABI of
// include/mapn.h.  Header-only; link with -lmapn.
//
// Same member names, argument meaning and error behaviour as the reference, so that the
// caller sequence of Particles.cpp compiles against it unchanged:
//
//     m_pCompute = new Compute(m_numParticlesSimulated, adapter, useExt);        // Particles.cpp:131
//     UINT64 fence = m_pCompute->GetFenceValue();                                 // Particles.cpp:446
//     m_pCompute->Simulate(m_numParticlesSimulated, fence);                       // Particles.cpp:448
//     m_pCompute->WaitForGpu();                                                   // Particles.cpp:470
//     m_pCompute = new Compute(n, adapter, useExt, pOldCompute);                  // Particles.cpp:515-516
//
// What differs, by necessity:
//   * the adapter is a HIP device ordinal instead of IDXGIAdapter1*;
//   * SharedHandles carries device pointers + ONE hipEvent_t (fetched once, re-recorded after
//     every step) instead of NT HANDLEs (Compute.h:54-62); the consumer's fence is signalled with
//     ConsumerSignal(); Simulate(n, v) queues its wait on it whether or not it has been signalled
//     yet (Compute.cpp:1012), unless the context was created with MAPN_FLAG_STRICT_CONSUMER;
//   * SetAsync takes two device pointers instead of ComPtr<ID3D12Resource>*;
//   * the launch plan bench.py's headline number is measured with (parts sized by the speed of the die they run on) is
//     one config bit away: in_pConfig->flags |= MAPN_FLAG_XCD_CALIBRATE (about 0.2 s more at construction);
//   * failures throw mapn::MapnException : std::runtime_error carrying the status code, the
//     counterpart of HrException (DXSampleHelper.h:29-46).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mapn.h"

namespace mapn {

class MapnException : public std::runtime_error {
public:
    explicit MapnException(int status)
        : std::runtime_error(std::string("mapn status ") + std::to_string(status) + ": " + mapn_last_error()),
          m_status(status) {}
    int Error() const { return m_status; }

private:
    const int m_status;
};

inline void ThrowIfFailed(int status)
{
    if (status != MAPN_OK) throw MapnException(status);
}

class Compute {
public:
    // Compute.h:36-39.  `in_device` replaces IDXGIAdapter1*; everything the reference hard-codes
    // can be overridden through `in_pConfig` (nullptr = the reference's constants).
    Compute(uint32_t in_numParticles, int in_device, bool in_useIntelCommandQueueExtension,
            Compute *in_pCompute = nullptr, const mapn_config *in_pConfig = nullptr)
        : m_numParticles(in_numParticles), m_usingIntelCommandQueueExtension(in_useIntelCommandQueueExtension)
    {
        mapn_config cfg;
        if (in_pConfig) cfg = *in_pConfig; else ThrowIfFailed(mapn_config_default(&cfg));
        cfg.num_particles = in_numParticles;
        cfg.device = in_device;
        if (in_pCompute) ThrowIfFailed(mapn_create_from(&cfg, in_pCompute->m_ctx, &m_ctx));   // CopyState
        else             ThrowIfFailed(mapn_create(&cfg, &m_ctx));                            // InitializeParticles
    }
    virtual ~Compute() { mapn_destroy(m_ctx); }                       // Compute.cpp:102-123

    Compute(const Compute &) = delete;                                // Compute.h:42-45
    Compute(Compute &&) = delete;
    Compute &operator=(const Compute &) = delete;
    Compute &operator=(Compute &&) = delete;

    // input is fence value of other adapter. waits to overwrite shared buffer.   (Compute.h:47-48)
    void Simulate(int in_numActiveParticles, uint64_t in_sharedFenceValue)
    {
        ThrowIfFailed(mapn_simulate(m_ctx, in_numActiveParticles, in_sharedFenceValue));
    }

    void SetUseIntelCommandQueueExtension(bool in_desiredSetting)     // Compute.h:51
    {
        ThrowIfFailed(mapn_set_use_intel_command_queue_extension(m_ctx, in_desiredSetting ? 1 : 0));
    }
    bool GetUsingIntelCommandQueueExtension() const { return mapn_get_using_intel_command_queue_extension(m_ctx) != 0; }
    bool GetIsUMA() const { return mapn_get_is_uma(m_ctx) != 0; }     // AdapterShared.h:60

    using SharedHandles = mapn_shared_handles;                        // Compute.h:54-61
    // Compute.h:62.  in_attachConsumerFence plays the role of in_fenceHandle: once attached,
    // Simulate(n, v) waits until the consumer signalled v - 1 (Compute.cpp:1012).
    const SharedHandles &GetSharedHandles(bool in_attachConsumerFence = true)
    {
        ThrowIfFailed(mapn_set_consumer(m_ctx, in_attachConsumerFence ? 1 : 0));
        ThrowIfFailed(mapn_get_shared_handles(m_ctx, &m_sharedHandles));
        return m_sharedHandles;
    }
    void ConsumerSignal(uint64_t in_value, void *in_hipEvent = nullptr)
    {
        ThrowIfFailed(in_hipEvent ? mapn_consumer_signal_event(m_ctx, in_value, in_hipEvent)
                                  : mapn_consumer_signal(m_ctx, in_value));
    }

    // bounds of the device-side waits in ms (0 = unchanged); D3D12's Queue::Wait has none
    void SetTimeouts(uint32_t in_p2pMs, uint32_t in_consumerMs) { ThrowIfFailed(mapn_set_timeouts(m_ctx, in_p2pMs, in_consumerMs)); }

    uint64_t GetFenceValue() const { return mapn_fence_value(m_ctx); }   // Compute.h:64

    struct ParticleVelocity { float velocity[3]; };                   // Compute.h:66-69 (XMFLOAT3)
    struct Particle { float position[4]; };                           // Render.h:85-88 (XMFLOAT4)

    virtual void WaitForGpu() { ThrowIfFailed(mapn_wait_idle(m_ctx)); }   // Compute.h:72

    void SetAsync(void *in_buffers[2], uint32_t in_bufferIndex)       // Compute.h:74-77
    {
        ThrowIfFailed(mapn_adopt_position_buffers(m_ctx, in_buffers, in_bufferIndex));
    }
    void ResetFromAsyncHelper() { ThrowIfFailed(mapn_reset_from_async(m_ctx)); }   // Compute.h:78

    // AdapterShared.h:51 / D3D12GpuTimer.h:54-55
    std::vector<std::pair<float, std::string>> GetGpuTimes() const
    {
        return {{mapn_last_step_seconds(m_ctx), std::string(mapn_timer_name())}};
    }

    // state hand-off (the parity harness; the reference only has the in-memory CopyState)
    void UploadState(const Particle *in_positions, const ParticleVelocity *in_velocities)
    {
        ThrowIfFailed(mapn_upload_state(m_ctx, reinterpret_cast<const float *>(in_positions),
                                        reinterpret_cast<const float *>(in_velocities)));
    }
    void DownloadState(Particle *out_positions, ParticleVelocity *out_velocities)
    {
        ThrowIfFailed(mapn_download_state(m_ctx, reinterpret_cast<float *>(out_positions),
                                          reinterpret_cast<float *>(out_velocities)));
    }

    mapn_ctx *Handle() const { return m_ctx; }
    uint32_t GetNumParticles() const { return m_numParticles; }

private:
    const uint32_t m_numParticles;
    bool m_usingIntelCommandQueueExtension;
    mapn_ctx *m_ctx = nullptr;
    SharedHandles m_sharedHandles{};
};

}  // namespace mapn
