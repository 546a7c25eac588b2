// omc_p2p.h -- direct write-to-all-peers exchange of the per-step regression moments (internal to libomc.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string>

#include "omc_kernels.h"

namespace omc {

struct P2P;  // one rank's mailbox + the peers' mailboxes as mapped here

constexpr int kP2PMaxWorld = 16;      // ranks of one node
constexpr int kP2PMaxPricings = 32;   // pricings advanced per launch (omc_price_american_seq)
constexpr int kP2PSlotDoubles = 8 * kP2PMaxPricings;
constexpr int kP2PHandleBytes = 64;   // sizeof(hipIpcMemHandle_t)

// allocate this rank's mailbox (fine-grained device memory) on the CURRENT device -> IPC handle for the peers
int p2p_export(P2P** out, void* handle_out, std::string* err);
// map every peer's mailbox: handles = world x kP2PHandleBytes, in rank order (the own entry is ignored)
int p2p_connect(P2P* p, int rank, int world, const void* handles, std::string* err);
void p2p_destroy(P2P* p);
bool p2p_connected(const P2P* p);
int p2p_world(const P2P* p);
void p2p_set_deadline(P2P* p, double seconds, double first_seconds);
void p2p_begin_call(P2P* p);  // the next exchange is a call's first: it waits first_seconds
// one exchange = reduce this rank's partials of step t, publish to all peers, gather, sum in rank order -> gmom[t]
hipError_t p2p_exchange_step(P2P* p, hipStream_t st, const LsmWorkspace& w, int t, int nblk);
hipError_t p2p_set_jobs(P2P* p, hipStream_t st, const double* const* part, double* const* gmom, const int* nblk,
                        const int* gstride, int n);
hipError_t p2p_exchange_step_multi(P2P* p, hipStream_t st, int K, int t);
hipError_t p2p_error_word(P2P* p, hipStream_t st, unsigned long long* out);
// results[i * 8 + 6] = (this rank's error word != 0) for i < n, ahead of the all-reduce of the result sums
hipError_t p2p_stamp_results(P2P* p, hipStream_t st, double* results, int n);

}  // namespace omc
