// HIP-event helpers exported through the C ABI so bench.py can time kernels on the launch stream.
#include "common.h"
extern "C" int tvts_event_create(void** ev) {
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    *ev = (void*)e;
    return (int)rc;
}
extern "C" int tvts_event_record(void* ev, hipStream_t stream) { return (int)hipEventRecord((hipEvent_t)ev, stream); }
extern "C" int tvts_event_elapsed_ms(void* start, void* stop, float* ms) {
    hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
    if (rc != hipSuccess) return (int)rc;
    return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
extern "C" int tvts_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }

// rate of the constant counter (s_memrealtime / wall_clock64) in kHz, and the CU count of the device
extern "C" int tvts_device_clock_info(int device, int* wall_clock_khz, int* cu_count, int* max_shader_khz) {
    hipError_t e = hipDeviceGetAttribute(wall_clock_khz, hipDeviceAttributeWallClockRate, device);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceGetAttribute(cu_count, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceGetAttribute(max_shader_khz, hipDeviceAttributeClockRate, device);
}

// zero fill through the runtime's own memset (a blit; capturable as a memset node): the flat gradient buffer at the start of a step
// (optimizer.zero_grad(), v2/trainer/trainer.py:476), loss accumulators, the dense gradient buffers of the non-pruned paths
extern "C" int tvts_zero_bytes(void* p, long nbytes, hipStream_t stream) {
    if (!p || nbytes < 0) return TVTS_EINVAL;
    if (nbytes == 0) return TVTS_OK;
    return (int)hipMemsetAsync(p, 0, (size_t)nbytes, stream);
}
