// omc_comm.h -- RCCL communicator behind the C ABI (internal to libomc.so).
//
// librccl.so is opened lazily (dlopen) the first time a communicator is asked for: the library
// keeps its "no HIP / no heavy load at import" property for the reference's spawn-ed worker
// processes, and single-GPU users never map RCCL at all.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string>

namespace omc {

struct Comm;  // opaque: one RCCL communicator bound to one device

constexpr int kCommUidBytes = 128;  // NCCL_UNIQUE_ID_BYTES

// rank 0 of a job: fresh unique id (ncclGetUniqueId) -> out[kCommUidBytes]
int comm_unique_id(void* out, std::string* err);
// collective over all ranks (ncclCommInitRank) on the CURRENT device
int comm_create(int rank, int world, const void* uid, Comm** out, std::string* err);
void comm_destroy(Comm* c);
int comm_rank(const Comm* c);
int comm_world(const Comm* c);  // ncclCommCount of the live communicator
// in-place all-reduce of `count` device doubles, enqueued on `st`; op 0 = sum, 1 = max
int comm_allreduce_f64(Comm* c, double* dptr, size_t count, int op, hipStream_t st, std::string* err);

}  // namespace omc
