// mapn_comm.h -- RCCL (over xGMI) communicator used by the sharded step.  Internal.
//
// librccl.so.1 is opened at run time (dlopen) the first time a communicator is asked for, so
// that the single-GPU path has no link-time dependency on RCCL and a process that already
// loaded RCCL (e.g. through torch.distributed) shares that copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mapn {

struct Comm;

int comm_get_unique_id(void *out_id128);                       // ncclGetUniqueId
Comm *comm_create(const void *id128, int rank, int nranks);    // ncclCommInitRank (collective)
void comm_destroy(Comm *c);
// all-gather of `floats_per_rank` floats per rank, in place: rank r's contribution already sits
// at buf + r * floats_per_rank (ncclAllGather with sendbuff == recvbuff + rank * count)
int comm_all_gather_inplace(Comm *c, void *buf, size_t floats_per_rank, hipStream_t stream);
int comm_gather_sendrecv_inplace(Comm *c, void *buf, size_t floats_per_rank, hipStream_t stream);
// the reaction rows of the sharded symmetric step as ONE group of point-to-point transfers: row [q] of `send` (floats_per_rank
// floats each) goes to every rank q of send_mask, row [q] of `recv` is filled by every rank q of recv_mask (never this rank)
int comm_exchange_rows(Comm *c, const void *send, void *recv, size_t floats_per_rank, unsigned send_mask, unsigned recv_mask, hipStream_t stream);
const char *comm_last_error();

}  // namespace mapn
