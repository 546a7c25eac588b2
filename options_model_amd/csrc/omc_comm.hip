// omc_comm.hip -- direct RCCL binding (no PyTorch): ncclGetUniqueId / ncclCommInitRank /
// ncclAllReduce on the context's stream.  The reference has no distributed code at all
// (SURVEY.md section 5.8); this is the "single RCCL all-reduce over xGMI" of BASELINE.json's
// north_star, reachable from any host language through include/omc.h.
#include "omc_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace omc {

namespace {

struct Api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string load_error;
};

Api g_api;
std::once_flag g_once;

void load_api()
{
    // OMC_RCCL_LIB overrides the library to open (tests, side-by-side ROCm installs)
    const char* names[] = {getenv("OMC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_api.handle) break;
        g_api.load_error = dlerror();
    }
    if (!g_api.handle) return;
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(g_api.handle, name);
        if (!p) {
            ok = false;
            g_api.load_error = std::string("missing symbol ") + name;
        }
        return p;
    };
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))sym("ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))sym("ncclCommInitRank");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))sym("ncclCommDestroy");
    g_api.CommCount = (decltype(g_api.CommCount))sym("ncclCommCount");
    g_api.CommUserRank = (decltype(g_api.CommUserRank))sym("ncclCommUserRank");
    g_api.AllReduce = (decltype(g_api.AllReduce))sym("ncclAllReduce");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(g_api.handle);
        g_api.handle = nullptr;
    }
}

bool api_ready(std::string* err)
{
    std::call_once(g_once, load_api);
    if (!g_api.handle) {
        if (err) *err = "librccl.so could not be loaded: " + g_api.load_error;
        return false;
    }
    return true;
}

int nccl_fail(const char* what, ncclResult_t r, std::string* err)
{
    if (err) *err = std::string(what) + " failed: " + (g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
    return 2000 + (int)r;
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

int comm_unique_id(void* out, std::string* err)
{
    static_assert(sizeof(ncclUniqueId) == kCommUidBytes, "unique id size");
    if (!api_ready(err)) return 2000;
    ncclUniqueId id;
    const ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r, err);
    memcpy(out, &id, sizeof id);
    return 0;
}

int comm_create(int rank, int world, const void* uid, Comm** out, std::string* err)
{
    if (!api_ready(err)) return 2000;
    ncclUniqueId id;
    memcpy(&id, uid, sizeof id);
    Comm* c = new Comm();
    const ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return nccl_fail("ncclCommInitRank", r, err);
    }
    // what the live communicator says, not what the caller asked for
    if (g_api.CommCount(c->comm, &c->world) != ncclSuccess) c->world = world;
    if (g_api.CommUserRank(c->comm, &c->rank) != ncclSuccess) c->rank = rank;
    *out = c;
    return 0;
}

void comm_destroy(Comm* c)
{
    if (!c) return;
    if (c->comm && g_api.handle) (void)g_api.CommDestroy(c->comm);
    delete c;
}

int comm_rank(const Comm* c) { return c ? c->rank : 0; }
int comm_world(const Comm* c) { return c ? c->world : 1; }

int comm_allreduce_f64(Comm* c, double* dptr, size_t count, int op, hipStream_t st, std::string* err)
{
    if (!c || !c->comm) {
        if (err) *err = "no communicator";
        return 2000;
    }
    const ncclResult_t r = g_api.AllReduce(dptr, dptr, count, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, st);
    if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r, err);
    return 0;
}

}  // namespace omc
