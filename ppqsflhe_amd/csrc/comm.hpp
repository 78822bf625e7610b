// comm.hpp -- RCCL behind the C-ABI: the exchange step of the aggregation over the GPUs of a node.
//
// Stands in for the serial per-client loop of the reference's server (orchestration/server_fns.sh:62-80,
// orchestration/run.sh:37-43: changeCipherDomain per client, then one aggregateEncryptedWeights): every GPU re-encrypts
// and sums its own clients, and the per-GPU partial sums meet in ONE ncclReduceScatter over 64-bit words.  RCCL has no
// modular-add reduction, but <= 8 canonical residues below 2^61 cannot wrap 2^64, so ncclSum on ncclUint64 followed by
// k_reduce (word mod q_i) equals the coefficient-wise modular sum.  Reduce-scatter rather than all-reduce: on the
// point-to-point xGMI fabric every GPU then receives only its 1/n shard, over all seven links at once.
//
// librccl.so.1 is resolved with dlopen at first use, not linked: a process that already carries an RCCL (PyTorch ships
// its own copy under the same SONAME) keeps exactly one, and hosts that never aggregate across GPUs need none.
#pragma once
#include <cstddef>
#include <cstdint>

namespace mk {

constexpr size_t COMM_ID_BYTES = 128;  // NCCL_UNIQUE_ID_BYTES

void comm_unique_id(void *h_id_out);                                             // ncclGetUniqueId
void *comm_create(int device, const void *h_id, int n_ranks, int rank);          // ncclCommInitRank on `device`
void comm_destroy(void *comm);                                                   // ncclCommDestroy
// recv[r-th block of recv_words] = sum over ranks of send[...] as uint64 (ncclSum), enqueued on `stream`
void comm_reduce_scatter_u64(void *comm, const uint64_t *d_send, uint64_t *d_recv, size_t recv_words, void *stream);
const char *comm_library_path();  // which librccl the process resolved (diagnostics)

}  // namespace mk
