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
    TTSK_NCCL(ncclCommInitRank(&g_comm, nranks, id, rank));
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

int ttsk_comm_destroy(void)
{
    if (g_comm) {
        ncclCommDestroy(g_comm);
        g_comm = nullptr;
        g_nranks = 0;
    }
    return TTSK_OK;
}

}  // extern "C"
