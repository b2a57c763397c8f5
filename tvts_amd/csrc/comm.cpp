// libtvts_comm.so: RCCL exchange steps of the data-parallel step on a library-owned side stream (include/tvts_comm.h).
// Host code only (HIP runtime API + RCCL); one communicator per process.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "tvts_comm.h"

#define TVTS_EINVAL (-22)
#define HIP_TRY(x)                                 \
    do {                                           \
        hipError_t e__ = (x);                      \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)
#define NCCL_TRY(x)                                          \
    do {                                                     \
        ncclResult_t r__ = (x);                              \
        if (r__ != ncclSuccess) return -(1000 + (int)r__);   \
    } while (0)

struct TvtsComm {
    ncclComm_t nccl;
    hipStream_t side;   // every collective runs here
    hipEvent_t fork;    // compute stream -> side stream
    hipEvent_t join;    // side stream -> compute stream
    int rank, world;
};

extern "C" int tvts_comm_unique_id(void* id128) {
    if (!id128) return TVTS_EINVAL;
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

static void comm_free(TvtsComm* c) {
    if (c->fork) hipEventDestroy(c->fork);
    if (c->join) hipEventDestroy(c->join);
    if (c->side) hipStreamDestroy(c->side);
    if (c->nccl) ncclCommDestroy(c->nccl);
    delete c;
}

// ncclCommInitRank is a rendezvous of every rank: a peer that died after the id broadcast leaves the others inside it forever.
// The bring-up therefore runs on a helper thread and the caller waits for it with a deadline.  On a timeout the caller gets
// TVTS_COMM_ETIMEDOUT and is free again (to agree with the other ranks on the fallback transport over ITS OWN channel); the helper
// stays parked in RCCL and, should the call ever return, aborts the communicator it was given itself.
struct InitJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclResult_t res = ncclSuccess;
    ncclComm_t comm = nullptr;
};

static int init_rank_deadline(ncclComm_t* out, int world, ncclUniqueId id, int rank, int device, int timeout_ms) {
    if (timeout_ms <= 0) {
        ncclResult_t r = ncclCommInitRank(out, world, id, rank);
        return r == ncclSuccess ? 0 : -(1000 + (int)r);
    }
    auto job = std::make_shared<InitJob>();
    std::thread([job, world, id, rank, device]() {
        ncclComm_t c = nullptr;
        ncclResult_t r = hipSetDevice(device) == hipSuccess ? ncclCommInitRank(&c, world, id, rank) : ncclUnhandledCudaError;
        std::unique_lock<std::mutex> lk(job->mu);
        if (job->abandoned) {  // nobody is waiting any more
            lk.unlock();
            if (r == ncclSuccess && c) ncclCommAbort(c);
            return;
        }
        job->res = r; job->comm = c; job->done = true;
        job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return job->done; })) {
        job->abandoned = true;
        return TVTS_COMM_ETIMEDOUT;
    }
    if (job->res != ncclSuccess) return -(1000 + (int)job->res);
    *out = job->comm;
    return 0;
}

extern "C" int tvts_comm_create(const void* id128, int rank, int world, void** comm_out) {
    return tvts_comm_create_deadline(id128, rank, world, 0, comm_out);
}

extern "C" int tvts_comm_create_deadline(const void* id128, int rank, int world, int timeout_ms, void** comm_out) {
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return TVTS_EINVAL;
    *comm_out = nullptr;
    TvtsComm* c = new TvtsComm();
    c->nccl = nullptr; c->side = nullptr; c->fork = nullptr; c->join = nullptr;
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    // every early return releases what exists so far (communicator, stream, events) and leaves *comm_out NULL
#define CREATE_TRY_NCCL(x) do { ncclResult_t r__ = (x); if (r__ != ncclSuccess) { comm_free(c); return -(1000 + (int)r__); } } while (0)
#define CREATE_TRY_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { comm_free(c); return (int)e__; } } while (0)
    int device = 0;
    CREATE_TRY_HIP(hipGetDevice(&device));
    {
        const int rc = init_rank_deadline(&c->nccl, world, id, rank, device, timeout_ms);
        if (rc) { c->nccl = nullptr; comm_free(c); return rc; }
    }
    // a high-priority side stream: its (few, short) kernels should not queue behind the persistent GEMM blocks
    int lo = 0, hi = 0;
    CREATE_TRY_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CREATE_TRY_HIP(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, hi));
    CREATE_TRY_HIP(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    CREATE_TRY_HIP(hipEventCreateWithFlags(&c->join, hipEventDisableTiming));
#undef CREATE_TRY_NCCL
#undef CREATE_TRY_HIP
    *comm_out = c;
    return 0;
}

extern "C" int tvts_comm_destroy(void* comm) {
    if (!comm) return TVTS_EINVAL;
    TvtsComm* c = (TvtsComm*)comm;
    // the first failure is reported, everything is released regardless
    int rc = 0;
    hipError_t e = hipStreamSynchronize(c->side);
    if (e != hipSuccess) rc = (int)e;
    ncclResult_t r = ncclCommDestroy(c->nccl);
    if (r != ncclSuccess && !rc) rc = -(1000 + (int)r);
    c->nccl = nullptr;
    e = hipEventDestroy(c->fork); if (e != hipSuccess && !rc) rc = (int)e;
    e = hipEventDestroy(c->join); if (e != hipSuccess && !rc) rc = (int)e;
    e = hipStreamDestroy(c->side); if (e != hipSuccess && !rc) rc = (int)e;
    delete c;
    return rc;
}

// Tears a communicator down WITHOUT waiting for its outstanding work: a collective whose peer is gone never completes, and
// with it neither would tvts_comm_destroy's stream synchronisation.  ncclCommAbort ends the kernels in flight.
extern "C" int tvts_comm_abort(void* comm) {
    if (!comm) return TVTS_EINVAL;
    TvtsComm* c = (TvtsComm*)comm;
    int rc = 0;
    ncclResult_t r = ncclCommAbort(c->nccl);
    if (r != ncclSuccess) rc = -(1000 + (int)r);
    c->nccl = nullptr;
    hipEventDestroy(c->fork);
    hipEventDestroy(c->join);
    hipStreamDestroy(c->side);
    delete c;
    return rc;
}

// 1 when everything issued on the side stream so far has completed, 0 when not (a poll: never blocks), < 0 on an error
extern "C" int tvts_comm_idle(void* comm) {
    if (!comm) return TVTS_EINVAL;
    TvtsComm* c = (TvtsComm*)comm;
    hipError_t e = hipStreamQuery(c->side);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return -(int)e;
}

extern "C" int tvts_comm_world(void* comm, int* rank, int* world) {
    if (!comm) return TVTS_EINVAL;
    TvtsComm* c = (TvtsComm*)comm;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return 0;
}

static int fork_from(TvtsComm* c, hipStream_t compute_stream) {
    HIP_TRY(hipEventRecord(c->fork, compute_stream));
    HIP_TRY(hipStreamWaitEvent(c->side, c->fork, 0));
    return 0;
}

extern "C" int tvts_comm_allgather_embeds(void* comm, const float* video, const float* text, int B, int E, float* video_all,
                                          float* text_all, hipStream_t compute_stream) {
    if (!comm || !video || !text || !video_all || !text_all || B <= 0 || E <= 0) return TVTS_EINVAL;
    TvtsComm* c = (TvtsComm*)comm;
    int rc = fork_from(c, compute_stream);
    if (rc) return rc;
    const size_t n = (size_t)B * E;
    NCCL_TRY(ncclGroupStart());  // both tensors in one launch
    NCCL_TRY(ncclAllGather(video, video_all, n, ncclFloat, c->nccl, c->side));
    NCCL_TRY(ncclAllGather(text, text_all, n, ncclFloat, c->nccl, c->side));
    NCCL_TRY(ncclGroupEnd());
    return 0;
}

extern "C" int tvts_comm_allreduce_bucket(void* comm, void* buf, long count, int dtype, hipStream_t compute_stream) {
    if (!comm || !buf || count <= 0 || (dtype != TVTS_COMM_F32 && dtype != TVTS_COMM_BF16)) return TVTS_EINVAL;
    TvtsComm* c = (TvtsComm*)comm;
    int rc = fork_from(c, compute_stream);
    if (rc) return rc;
    NCCL_TRY(ncclAllReduce(buf, buf, (size_t)count, dtype == TVTS_COMM_F32 ? ncclFloat : ncclBfloat16, ncclSum, c->nccl, c->side));
    return 0;
}

extern "C" int tvts_comm_wait(void* comm, hipStream_t compute_stream) {
    if (!comm) return TVTS_EINVAL;
    TvtsComm* c = (TvtsComm*)comm;
    HIP_TRY(hipEventRecord(c->join, c->side));
    HIP_TRY(hipStreamWaitEvent(compute_stream, c->join, 0));
    return 0;
}
