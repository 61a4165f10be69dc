// svo_rccl.cpp -- libsvo_rccl.so: the two exchange steps of the multi-GPU path over RCCL (include/svo_rccl.h).
// The data path needs neither (streams are independent); this is the pose-record all-gather of
// BASELINE.json configs[3] and the neighbour hand-over of frame-parallelism within one stream, for hosts that stay C / C++.
#include "../../include/svo_rccl.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

static_assert(sizeof(ncclUniqueId) <= SVO_GROUP_ID_BYTES, "unique id slot");

struct svo_group {
    int n_ranks = 0;
    bool local = false;                    // all ranks in this process (one communicator per rank) or just ours
    int my_rank = 0;                       // per-process group: our rank
    std::vector<ncclComm_t> comm;
    std::vector<int> device;
    std::vector<hipEvent_t> ev;            // orders the gather after the context's stream when another stream carries it
    // a local group is driven by one host thread per GPU and ranks tend to fail together: the text of the last failure is kept
    // under a lock (svo_group_last_error hands out a per-thread copy)
    std::mutex err_lock;
    std::string last_error;
};

static int fail(svo_group* g, const char* what, const char* text)
{
    if (g) { std::lock_guard<std::mutex> lk(g->err_lock); g->last_error = std::string(what) + ": " + text; }
    return SVO_ERR_HIP;
}
#define NCCLCHECK(g, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return fail(g, #expr, ncclGetErrorString(_r)); } while (0)
#define HIPCHK(g, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(g, #expr, hipGetErrorString(_e)); } while (0)

static inline int slot_of(const svo_group* g, int rank) { return g->local ? rank : 0; }

extern "C" int svo_group_create_local(const int* devices, int n, svo_group** out)
{
    if (!devices || n <= 0 || !out) return SVO_ERR_ARG;
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (devices[i] == devices[j]) return SVO_ERR_ARG;
    svo_group* g = new svo_group();
    *out = g;
    g->n_ranks = n; g->local = true; g->comm.resize(n); g->device.assign(devices, devices + n); g->ev.assign(n, nullptr);
    NCCLCHECK(g, ncclCommInitAll(g->comm.data(), n, devices));
    for (int r = 0; r < n; r++) { HIPCHK(g, hipSetDevice(devices[r])); HIPCHK(g, hipEventCreateWithFlags(&g->ev[r], hipEventDisableTiming)); }
    return SVO_OK;
}

extern "C" int svo_group_unique_id(char id[SVO_GROUP_ID_BYTES])
{
    if (!id) return SVO_ERR_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return SVO_ERR_HIP;
    std::memset(id, 0, SVO_GROUP_ID_BYTES); std::memcpy(id, &u, sizeof(u));
    return SVO_OK;
}

extern "C" int svo_group_create_rank(const char id[SVO_GROUP_ID_BYTES], int n_ranks, int rank, int device, svo_group** out)
{
    if (!id || n_ranks <= 0 || rank < 0 || rank >= n_ranks || !out) return SVO_ERR_ARG;
    svo_group* g = new svo_group();
    *out = g;
    g->n_ranks = n_ranks; g->local = false; g->my_rank = rank; g->comm.assign(1, nullptr); g->device.assign(1, device); g->ev.assign(1, nullptr);
    HIPCHK(g, hipSetDevice(device));
    ncclUniqueId u; std::memcpy(&u, id, sizeof(u));
    NCCLCHECK(g, ncclCommInitRank(&g->comm[0], n_ranks, u, rank));
    HIPCHK(g, hipEventCreateWithFlags(&g->ev[0], hipEventDisableTiming));
    return SVO_OK;
}

extern "C" void svo_group_destroy(svo_group* g)
{
    if (!g) return;
    for (size_t i = 0; i < g->comm.size(); i++) {
        (void)hipSetDevice(g->device[i]);
        if (g->ev[i]) (void)hipEventDestroy(g->ev[i]);
        if (g->comm[i]) (void)ncclCommDestroy(g->comm[i]);
    }
    delete g;
}

extern "C" int svo_group_size(const svo_group* g) { return g ? g->n_ranks : SVO_ERR_ARG; }
extern "C" const char* svo_group_last_error(const svo_group* g)
{
    static thread_local std::string copy;
    if (!g) return "";
    { std::lock_guard<std::mutex> lk(const_cast<svo_group*>(g)->err_lock); copy = g->last_error; }
    return copy.c_str();
}

extern "C" int svo_group_allgather_results(svo_group* g, int rank, svo_ctx* ctx, void* stream, void* dev_records, size_t bytes)
{
    if (!g || !ctx || !dev_records) return SVO_ERR_ARG;
    if (!g->local) rank = g->my_rank;
    if (rank < 0 || rank >= g->n_ranks || bytes % ((size_t)g->n_ranks * sizeof(svo_result)) != 0) return SVO_ERR_ARG;
    const int s = slot_of(g, rank);
    const size_t chunk = bytes / (size_t)g->n_ranks;
    // the context must live on this rank's GPU: the communicator, the event and the receive buffer do
    if (svo_get_device(ctx) != g->device[s]) { fail(g, "svo_group_allgather_results", "the context's device is not this rank's"); return SVO_ERR_ARG; }
    // in place: this rank's records go straight into their slot of the receive buffer, on the context's stream
    const int rc = svo_copy_results_async(ctx, (char*)dev_records + (size_t)rank * chunk, chunk);
    if (rc != SVO_OK) { fail(g, "svo_copy_results_async", svo_last_error(ctx)); return rc; }
    void* cs = nullptr;
    if (svo_get_stream(ctx, &cs) != SVO_OK) return SVO_ERR_ARG;
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)cs;
    HIPCHK(g, hipSetDevice(g->device[s]));
    if (st != (hipStream_t)cs) { HIPCHK(g, hipEventRecord(g->ev[s], (hipStream_t)cs)); HIPCHK(g, hipStreamWaitEvent(st, g->ev[s], 0)); }
    NCCLCHECK(g, ncclAllGather((const char*)dev_records + (size_t)rank * chunk, dev_records, chunk, ncclChar, g->comm[s], st));
    return SVO_OK;
}

extern "C" int svo_group_allgather_inplace(svo_group* g, int rank, void* dev_table, size_t bytes, void* stream)
{
    if (!g || !dev_table) return SVO_ERR_ARG;
    if (!g->local) rank = g->my_rank;
    if (rank < 0 || rank >= g->n_ranks || bytes == 0 || bytes % (size_t)g->n_ranks != 0) return SVO_ERR_ARG;
    const int s = slot_of(g, rank);
    const size_t chunk = bytes / (size_t)g->n_ranks;
    HIPCHK(g, hipSetDevice(g->device[s]));
    NCCLCHECK(g, ncclAllGather((const char*)dev_table + (size_t)rank * chunk, dev_table, chunk, ncclChar, g->comm[s], (hipStream_t)stream));
    return SVO_OK;
}

extern "C" int svo_group_comm_count(const svo_group* g, int rank)
{
    if (!g) return SVO_ERR_ARG;
    if (!g->local) rank = g->my_rank;
    if (rank < 0 || rank >= g->n_ranks) return SVO_ERR_ARG;
    int n = 0;
    return ncclCommCount(g->comm[slot_of(g, rank)], &n) == ncclSuccess ? n : SVO_ERR_HIP;
}

extern "C" int svo_group_comm_device(const svo_group* g, int rank)
{
    if (!g) return SVO_ERR_ARG;
    if (!g->local) rank = g->my_rank;
    if (rank < 0 || rank >= g->n_ranks) return SVO_ERR_ARG;
    int d = -1;
    return ncclCommCuDevice(g->comm[slot_of(g, rank)], &d) == ncclSuccess ? d : SVO_ERR_HIP;
}

extern "C" int svo_group_send_frame(svo_group* g, int rank, int to_rank, const void* dev_blob, size_t bytes, void* stream)
{
    if (!g || !dev_blob) return SVO_ERR_ARG;
    if (!g->local) rank = g->my_rank;
    if (rank < 0 || rank >= g->n_ranks || to_rank < 0 || to_rank >= g->n_ranks || to_rank == rank) return SVO_ERR_ARG;
    const int s = slot_of(g, rank);
    HIPCHK(g, hipSetDevice(g->device[s]));
    NCCLCHECK(g, ncclSend(dev_blob, bytes, ncclChar, to_rank, g->comm[s], (hipStream_t)stream));
    return SVO_OK;
}

extern "C" int svo_group_recv_frame(svo_group* g, int rank, int from_rank, void* dev_blob, size_t bytes, void* stream)
{
    if (!g || !dev_blob) return SVO_ERR_ARG;
    if (!g->local) rank = g->my_rank;
    if (rank < 0 || rank >= g->n_ranks || from_rank < 0 || from_rank >= g->n_ranks || from_rank == rank) return SVO_ERR_ARG;
    const int s = slot_of(g, rank);
    HIPCHK(g, hipSetDevice(g->device[s]));
    NCCLCHECK(g, ncclRecv(dev_blob, bytes, ncclChar, from_rank, g->comm[s], (hipStream_t)stream));
    return SVO_OK;
}
