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

// Clock probe for bench.py ("confirm the peak on the box"): one lane samples the shader-clock counter (s_memtime: one tick
// per shader cycle) against the constant-rate counter (s_memrealtime) over `ref_ticks` of the latter.  Launched on the GEMMs' stream
// right behind a GEMM launch it reads the clock the power management is holding under that load (DVFS moves on a millisecond scale,
// the probe lasts ~10 us).  out[0] = shader cycles, out[1] = constant-rate ticks.
__global__ void clock_probe_kernel(unsigned long long* out, int ref_ticks) {
    if (threadIdx.x == 0) {
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
        unsigned long long r1;
        do {
            __builtin_amdgcn_s_sleep(4);
            r1 = __builtin_amdgcn_s_memrealtime();
        } while (r1 - r0 < (unsigned long long)ref_ticks);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
}
extern "C" int tvts_clock_probe(void* out2_u64, int ref_ticks, hipStream_t stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, (unsigned long long*)out2_u64, ref_ticks);
    TVTS_LAUNCH_CHECK();
    return TVTS_OK;
}
// rate of the constant counter (s_memrealtime / wall_clock64) in kHz, and the CU count of the device
extern "C" int tvts_device_clock_info(int device, int* wall_clock_khz, int* cu_count, int* max_shader_khz) {
    hipError_t e = hipDeviceGetAttribute(wall_clock_khz, hipDeviceAttributeWallClockRate, device);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceGetAttribute(cu_count, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return (int)e;
    return (int)hipDeviceGetAttribute(max_shader_khz, hipDeviceAttributeClockRate, device);
}
