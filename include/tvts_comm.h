/* C ABI of libtvts_comm.so -- the two exchange steps of the data-parallel TVTSv2 step on RCCL, issued on a side HIP
 * stream the library owns, so that the traffic over xGMI runs under the backward GEMMs of the compute stream.
 *
 * What each entry point replaces in the reference (TencentARC/TVTS v2):
 *   tvts_comm_allgather_embeds   AllGather_multi.forward, v2/trainer/trainer.py:41-51 (dist.all_gather + cat of the
 *                                [B,E] video and text embeddings, called twice at :481-482) -- here ONE grouped RCCL
 *                                launch for both tensors, no packing copy.  The backward (:52-57) is a local row slice and
 *                                needs no collective.
 *   tvts_comm_allreduce_bucket   the gradient reduction DistributedDataParallel performs for the model wrapped at
 *                                v2/base/base_trainer.py:20-25 -- here a SUM over a range of the flat gradient buffer,
 *                                issued by the hand-written backward as soon as the range is final; the 1/world average
 *                                is folded into the AdamW kernel (tvts_adamw_hf's grad_scale).
 * Conventions as in tvts_hip.h: raw device pointers, no allocation of data buffers inside, asynchronous, return 0 or an
 * error code (hipError_t > 0, -22 invalid argument, -(1000 + ncclResult_t) for an RCCL error).  Ordering: every call
 * first makes the side stream wait for the work already enqueued on `compute_stream` (an event), so producers need no
 * host synchronisation; tvts_comm_wait makes `compute_stream` wait for everything issued on the side stream so far.
 * One communicator per process (= per GPU); not thread-safe per communicator. */
#ifndef TVTS_COMM_H
#define TVTS_COMM_H
#include <hip/hip_runtime_api.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { TVTS_COMM_F32 = 0, TVTS_COMM_BF16 = 1 };
#define TVTS_COMM_ETIMEDOUT (-110)

/* 128-byte RCCL unique id, created on rank 0 and handed to every rank by the host side (torch.distributed broadcast) */
int tvts_comm_unique_id(void* id128);
/* collective over all ranks: creates the communicator for the CURRENT HIP device plus the side stream and its events */
int tvts_comm_create(const void* id128, int rank, int world, void** comm_out);
/* the same with a deadline on the rendezvous (round 5): ncclCommInitRank runs on a helper thread; if it has not returned after
 * timeout_ms (> 0) the call gives up with TVTS_COMM_ETIMEDOUT and *comm_out = NULL -- a peer that died after the id broadcast costs
 * the survivors the deadline, not the job: they are free to agree on the torch.distributed transport.  (The helper thread stays
 * parked inside RCCL; should its call ever return it aborts the communicator itself.)  timeout_ms <= 0: tvts_comm_create. */
int tvts_comm_create_deadline(const void* id128, int rank, int world, int timeout_ms, void** comm_out);
int tvts_comm_destroy(void* comm);
/* tear-down that does NOT wait for outstanding collectives (ncclCommAbort): for a communicator whose peer is gone */
int tvts_comm_abort(void* comm);
/* poll: 1 = every collective issued so far has completed, 0 = not yet, < 0 = error; never blocks */
int tvts_comm_idle(void* comm);
int tvts_comm_world(void* comm, int* rank, int* world);
/* video_all[W*B, E] and text_all[W*B, E] <- all ranks' video[B, E] / text[B, E] (fp32), rank r owns rows r*B..r*B+B-1 */
int tvts_comm_allgather_embeds(void* comm, const float* video, const float* text, int B, int E, float* video_all,
                               float* text_all, hipStream_t compute_stream);
/* in-place SUM all-reduce of buf[0..count) (dtype TVTS_COMM_F32 / TVTS_COMM_BF16) */
int tvts_comm_allreduce_bucket(void* comm, void* buf, long count, int dtype, hipStream_t compute_stream);
/* compute_stream waits (on the device) for every collective issued so far */
int tvts_comm_wait(void* comm, hipStream_t compute_stream);

#ifdef __cplusplus
}
#endif
#endif
