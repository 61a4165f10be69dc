/* svo_rccl.h -- the exchange steps of the multi-GPU path as a C-ABI companion library (libsvo_rccl.so, links RCCL;
 * libsvo_hip.so itself does not, so a single-GPU host never loads it).
 *
 * The path shards by STREAM: one rso::CStereoOdometryEstimator per stream, all of an estimator's state per instance
 * (libstereo-odometry.h:732-831), no data-path exchange.  Two exchanges exist around it:
 *   (1) every rank wants every stream's pose each frame: ONE all-gather of the fixed-size svo_result records
 *       (svo_group_allgather_results), a few KB, latency only;
 *   (2) frame-parallelism within one stream: the owner of frame t-1 hands "what the next call would find as its previous
 *       frame" (svo_export_frame's record, svo_hip.h) to the owner of frame t: one point-to-point send / receive of that
 *       record between neighbouring ranks (svo_group_send_frame / svo_group_recv_frame) over xGMI.
 * Ranks are GPUs.  A group is made either inside one process (one host thread per GPU: svo_group_create_local, the
 * reference's own threading model, SURVEY.md 8b "Threading") or across processes (svo_group_unique_id +
 * svo_group_create_rank: the id travels by whatever the host already has, e.g. torch.distributed's store).
 * All calls enqueue on the stream given and return; nothing here synchronises the host.  Return codes: svo_hip.h's
 * (SVO_OK = 0, SVO_ERR_ARG, SVO_ERR_HIP for a HIP or RCCL failure; svo_group_last_error has the text).
 */
#ifndef SVO_RCCL_H
#define SVO_RCCL_H
#include "svo_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct svo_group svo_group;
#define SVO_GROUP_ID_BYTES 128

/* one process, n GPUs: devices[r] is rank r's GPU (all different: RCCL refuses two ranks on one device) */
int svo_group_create_local(const int* devices, int n, svo_group** out);
/* one rank per process: rank 0 calls svo_group_unique_id and ships the bytes to the others */
int svo_group_unique_id(char id[SVO_GROUP_ID_BYTES]);
int svo_group_create_rank(const char id[SVO_GROUP_ID_BYTES], int n_ranks, int rank, int device, svo_group** out);
void svo_group_destroy(svo_group* g);
int svo_group_size(const svo_group* g);
const char* svo_group_last_error(const svo_group* g);

/* (1) results of the frame just enqueued on `ctx`, from every rank, into `dev_records` on this rank's GPU:
 *     n_ranks x lanes_per_rank svo_result records in rank order.  Every rank calls it with the same lanes_per_rank
 *     (= its context's n_lanes) once per frame; `rank` is the caller's rank (ignored for a per-process group).
 *     Enqueued on `stream` (NULL: the context's own) after the frame's kernels, so it overlaps the host's next call. */
int svo_group_allgather_results(svo_group* g, int rank, svo_ctx* ctx, void* stream, void* dev_records, size_t bytes);

/* the same exchange for records that are already in place: `dev_table` holds n_ranks equal slots (bytes in all), this rank's
 * slot was written on this GPU by work that `stream` already waits for (e.g. svo_batch_set_results_buffer pointed a batch at
 * it and svo_batch_wait_on_stream ordered `stream` behind the step); in-place ncclAllGather on `stream`. */
int svo_group_allgather_inplace(svo_group* g, int rank, void* dev_table, size_t bytes, void* stream);
/* what the communicator itself says about the group (ncclCommCount / ncclCommUserRank / ncclCommCuDevice of `rank`'s
 * communicator): lets a host print, next to its numbers, how many ranks RCCL really connected */
int svo_group_comm_count(const svo_group* g, int rank);
int svo_group_comm_device(const svo_group* g, int rank);

/* (2) hand-over record of svo_export_frame / svo_import_frame between two ranks; bytes = svo_handover_bytes(ctx) */
int svo_group_send_frame(svo_group* g, int rank, int to_rank, const void* dev_blob, size_t bytes, void* stream);
int svo_group_recv_frame(svo_group* g, int rank, int from_rank, void* dev_blob, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
