// Multi-GPU exchange: ONE RCCL sum of the packed partial sketch [Psi_0..Psi_{d-1},
// Omega_0..Omega_{d-2}] over xGMI -- the device form of SketchContainer.__add__
// (sketch_container.py:61-69) across ranks.  One process per GPU; the 128-byte RCCL id
// is created on rank 0 and carried to the other ranks by the caller (any host channel).
#include <rccl/rccl.h>
#include <cstring>
#include "common.h"

namespace ttsk {
static ncclComm_t g_comm = nullptr;
static int g_nranks = 0;
}  // namespace ttsk
using namespace ttsk;

#define TTSK_NCCL(call)                                                              \
    do {                                                                             \
        ncclResult_t r_ = (call);                                                    \
        if (r_ != ncclSuccess) {                                                     \
            ttsk::set_error("%s failed: %s", #call, ncclGetErrorString(r_));         \
            return TTSK_ERR_COMM;                                                    \
        }                                                                            \
    } while (0)

extern "C" {

int ttsk_comm_unique_id(void *host_id128)
{
    TTSK_ARG(host_id128, "ttsk_comm_unique_id: NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id size");
    ncclUniqueId id;
    TTSK_NCCL(ncclGetUniqueId(&id));
    memcpy(host_id128, &id, 128);
    return TTSK_OK;
}

int ttsk_comm_init(const void *host_id128, int rank, int nranks)
{
    if (ensure_init() != TTSK_OK) return TTSK_ERR_HIP;
    TTSK_ARG(host_id128 && rank >= 0 && rank < nranks, "ttsk_comm_init: bad rank %d/%d", rank, nranks);
    TTSK_ARG(g_comm == nullptr, "ttsk_comm_init: communicator already exists");
    ncclUniqueId id;
    memcpy(&id, host_id128, 128);
    // The handle goes to the global only once the communicator is complete: a failed init must leave
    // the library exactly as it was (no half-built handle for ttsk_comm_destroy or a later call to
    // find).  RCCL 2.27 frees its own state and nulls the handle when ncclCommInitRank fails; a
    // non-null handle next to an error would be a communicator it still owns, and is aborted here.
    ncclComm_t c = nullptr;
    const ncclResult_t r = ncclCommInitRank(&c, nranks, id, rank);
    if (r != ncclSuccess) {
        if (c) (void)ncclCommAbort(c);
        ttsk::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, ncclGetErrorString(r));
        return TTSK_ERR_COMM;
    }
    g_comm = c;
    g_nranks = nranks;
    return TTSK_OK;
}

int ttsk_comm_allreduce_sum(double *dev_buf, size_t n, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(g_comm, "ttsk_comm_allreduce_sum: communicator not initialised");
    if (n == 0) return TTSK_OK;
    TTSK_NCCL(ncclAllReduce(dev_buf, dev_buf, n, ncclDouble, ncclSum, g_comm, st));
    return TTSK_OK;
}

int ttsk_comm_reduce_sum(double *dev_buf, size_t n, int root, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(g_comm, "ttsk_comm_reduce_sum: communicator not initialised");
    TTSK_ARG(root >= 0 && root < g_nranks, "ttsk_comm_reduce_sum: bad root %d", root);
    if (n == 0) return TTSK_OK;
    TTSK_NCCL(ncclReduce(dev_buf, dev_buf, n, ncclDouble, ncclSum, root, g_comm, st));
    return TTSK_OK;
}

/* Every rank's n doubles, in rank order, into recv (n * nranks doubles): the placement step of a
 * rank-sharded sketch (blocked_stream_sketch, sketch.py:364-397,446-473: DRM rank slices give
 * disjoint blocks of Psi / Omega, assembled by placement -- no sum). */
int ttsk_comm_allgather(const double *dev_send, double *dev_recv, size_t n, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(g_comm, "ttsk_comm_allgather: communicator not initialised");
    TTSK_ARG(dev_send && dev_recv, "ttsk_comm_allgather: NULL buffer");
    if (n == 0) return TTSK_OK;
    TTSK_NCCL(ncclAllGather(dev_send, dev_recv, n, ncclDouble, g_comm, st));
    return TTSK_OK;
}

/* max over ranks of every element, in place (the bench's max-over-ranks clock) */
int ttsk_comm_allreduce_max(double *dev_buf, size_t n, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(g_comm, "ttsk_comm_allreduce_max: communicator not initialised");
    if (n == 0) return TTSK_OK;
    TTSK_NCCL(ncclAllReduce(dev_buf, dev_buf, n, ncclDouble, ncclMax, g_comm, st));
    return TTSK_OK;
}

int ttsk_comm_destroy(void)
{
    if (g_comm) {
        ncclComm_t c = g_comm;
        g_comm = nullptr;          // cleared first: a failing destroy must not leave a dangling handle
        g_nranks = 0;
        TTSK_NCCL(ncclCommDestroy(c));
    }
    return TTSK_OK;
}

}  // extern "C"
