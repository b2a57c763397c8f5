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
