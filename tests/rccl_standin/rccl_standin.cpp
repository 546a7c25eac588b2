// rccl_standin.cpp -- TEST INFRASTRUCTURE ONLY, never part of the product.
//
// A stand-in for the seven librccl entry points libomc.so binds (csrc/omc_comm.hip loads whatever
// OMC_RCCL_LIB names): ncclGetUniqueId / ncclCommInitRank / ncclCommCount / ncclCommUserRank /
// ncclAllReduce / ncclCommDestroy / ncclGetErrorString.  Real RCCL refuses two ranks on one device
// ("Duplicate GPU detected"), so the N > 1 code paths of libomc.so -- omc_comm_init, the all-reduces the
// pricing calls enqueue on the context's stream (251 per pricing in the per-step flows), the overlapped
// two-stream sequence -- cannot run on the one-GPU development box with it.  With this library they can:
// N processes that share ONE GPU exchange through a POSIX shared-memory segment.
//
// ncclAllReduce keeps RCCL's contract towards its caller: it only ENQUEUES on the given stream
//     device -> pinned host copy;  host function (the exchange);  pinned host -> device copy
// and returns.  The exchange sums the ranks' contributions in rank order, so every rank gets the same bits.
// Collectives of one communicator must be issued in the same order by every rank (as with RCCL); their
// sequence number is taken at enqueue time.  All waits are bounded (OMC_STANDIN_TIMEOUT_S, default 120):
// a rank that never arrives turns into ncclSystemError / a poisoned result, not a hang.
//
// Failure injection for the launcher tests: OMC_STANDIN_FAIL_RANK=r makes ncclCommInitRank fail on rank r
// at once while the other ranks wait for it (what a dead peer looks like to real RCCL).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

namespace {

constexpr int kMaxRanks = 16;
constexpr size_t kSlotDoubles = 4096;  // per rank and collective; larger counts go in several rounds
constexpr size_t kRingBytes = 64u << 20;

struct Shared {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> left;
    std::atomic<uint32_t> abort_flag;
    uint32_t pad;
    std::atomic<uint64_t> ready[kMaxRanks];  // last collective whose contribution rank r has published
    std::atomic<uint64_t> done[kMaxRanks];   // last collective rank r has finished reading
    double slot[kMaxRanks][kSlotDoubles];
};

double timeout_s()
{
    const char* e = getenv("OMC_STANDIN_TIMEOUT_S");
    const double v = e ? atof(e) : 120.0;
    return v > 0 ? v : 120.0;
}

double now_s()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

// spin (politely) until pred() or the deadline; false = timed out
template <class P>
bool wait_until(P pred, double limit_s)
{
    const double t0 = now_s();
    for (unsigned i = 0;; ++i) {
        if (pred()) return true;
        if ((i & 63) == 63) {
            if (now_s() - t0 > limit_s) return false;
            timespec ts{0, 20000};
            nanosleep(&ts, nullptr);
        }
    }
}

}  // namespace

struct ncclComm {
    Shared* sh = nullptr;
    int rank = 0, world = 1;
    uint64_t next_seq = 1;       // sequence number of the next collective round (enqueue order)
    std::atomic<int> error{0};
    char* ring = nullptr;        // pinned staging
    size_t ring_off = 0;
    std::mutex mu;
};

namespace {

struct Round {
    ncclComm* c;
    const double* send;  // pinned
    double* recv;        // pinned
    size_t count;
    int op;              // 0 sum, 2 max
    uint64_t seq;
};

void exchange(void* arg)
{
    Round* r = (Round*)arg;
    ncclComm* c = r->c;
    Shared* sh = c->sh;
    const double lim = timeout_s();
    auto poison = [&] {
        c->error.store(1);
        sh->abort_flag.store(1);
        for (size_t i = 0; i < r->count; ++i) r->recv[i] = NAN;
    };
    if (sh->abort_flag.load() || c->error.load()) {
        poison();
        delete r;
        return;
    }
    const uint64_t s = r->seq;
    // everybody has finished READING round s-1 before anybody overwrites a slot
    bool ok = wait_until([&] {
        for (int k = 0; k < c->world; ++k)
            if (sh->done[k].load(std::memory_order_acquire) + 1 < s) return false;
        return sh->abort_flag.load() == 0;
    }, lim);
    if (ok) {
        memcpy(sh->slot[c->rank], r->send, sizeof(double) * r->count);
        sh->ready[c->rank].store(s, std::memory_order_release);
        ok = wait_until([&] {
            for (int k = 0; k < c->world; ++k)
                if (sh->ready[k].load(std::memory_order_acquire) < s) return false;
            return sh->abort_flag.load() == 0;
        }, lim);
    }
    if (!ok || sh->abort_flag.load()) {
        fprintf(stderr, "rccl_standin rank %d: collective %llu timed out / aborted\n", c->rank, (unsigned long long)s);
        poison();
        delete r;
        return;
    }
    for (size_t i = 0; i < r->count; ++i) {
        double acc = sh->slot[0][i];
        for (int k = 1; k < c->world; ++k) {
            const double v = sh->slot[k][i];
            acc = r->op == 2 ? (v > acc ? v : acc) : acc + v;  // rank order: the same bits on every rank
        }
        r->recv[i] = acc;
    }
    sh->done[c->rank].store(s, std::memory_order_release);
    delete r;
}

char* ring_take(ncclComm* c, size_t bytes)
{
    bytes = (bytes + 255) & ~(size_t)255;
    if (c->ring_off + bytes > kRingBytes) {
        (void)hipDeviceSynchronize();  // every earlier round has consumed its staging
        c->ring_off = 0;
    }
    char* p = c->ring + c->ring_off;
    c->ring_off += bytes;
    return p;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/omc_standin_%d_%llx", (int)getpid(),
             (unsigned long long)t.tv_sec * 1000000000ull + (unsigned long long)t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (strncmp(id.internal, "/omc_standin_", 13) != 0) return ncclInvalidArgument;
    const char* fr = getenv("OMC_STANDIN_FAIL_RANK");
    if (fr && *fr && atoi(fr) == rank) {
        fprintf(stderr, "rccl_standin rank %d: ncclCommInitRank fails on request (OMC_STANDIN_FAIL_RANK)\n", rank);
        return ncclSystemError;
    }
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) {  // fresh segments are zero-filled: a valid initial state
        close(fd);
        return ncclSystemError;
    }
    void* m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    ncclComm* c = new (std::nothrow) ncclComm();
    if (!c) return ncclSystemError;
    c->sh = (Shared*)m;
    c->rank = rank;
    c->world = nranks;
    c->sh->arrived.fetch_add(1);
    const bool all = wait_until([&] { return c->sh->arrived.load() >= (uint32_t)nranks; }, timeout_s());
    if (rank == 0) shm_unlink(id.internal);  // everybody who will ever map it has it mapped (or never comes)
    if (!all) {
        fprintf(stderr, "rccl_standin rank %d: only %u of %d ranks joined\n", rank, c->sh->arrived.load(), nranks);
        munmap(m, sizeof(Shared));
        delete c;
        return ncclSystemError;
    }
    if (hipHostMalloc((void**)&c->ring, kRingBytes, hipHostMallocDefault) != hipSuccess) {
        munmap(m, sizeof(Shared));
        delete c;
        return ncclSystemError;
    }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    (void)hipDeviceSynchronize();
    if (comm->ring) (void)hipHostFree(comm->ring);
    comm->sh->left.fetch_add(1);
    munmap(comm->sh, sizeof(Shared));
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->world;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank)
{
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = comm->rank;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    if (datatype != ncclDouble || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;  // all libomc.so uses
    if (comm->error.load() || comm->sh->abort_flag.load()) return ncclSystemError;
    std::lock_guard<std::mutex> g(comm->mu);
    const char* src = (const char*)sendbuff;
    char* dst = (char*)recvbuff;
    for (size_t done = 0; done < count; done += kSlotDoubles) {
        const size_t n = count - done < kSlotDoubles ? count - done : kSlotDoubles;
        Round* r = new Round();
        r->c = comm;
        double* stage = (double*)ring_take(comm, 2 * sizeof(double) * n);
        r->send = stage;
        r->recv = stage + n;
        r->count = n;
        r->op = (int)op;
        r->seq = comm->next_seq++;
        if (hipMemcpyAsync(stage, src + sizeof(double) * done, sizeof(double) * n, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipLaunchHostFunc(stream, exchange, r) != hipSuccess ||
            hipMemcpyAsync(dst + sizeof(double) * done, stage + n, sizeof(double) * n, hipMemcpyHostToDevice, stream) != hipSuccess) {
            comm->error.store(1);
            return ncclSystemError;
        }
    }
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "unhandled system error (rccl_standin)";
    case ncclInvalidArgument: return "invalid argument (rccl_standin)";
    default: return "error (rccl_standin)";
    }
}

}  // extern "C"
