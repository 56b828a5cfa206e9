// RCCL inside the library: the one collective of the path, the sum of the moment Q over the ranks that share the contour nodes
// (src/feast.jl:59-71 is the loop that is sharded).  RCCL is resolved with dlopen at the first use, so libfeasthip.so has
// no link-time dependency on it and a host process that already loaded a copy (torch ships one) shares that copy.
#include "engine_impl.hpp"
#include <dlfcn.h>

namespace feasthip {

namespace {
// the slice of the RCCL API that is used (rccl.h: ncclUniqueId is 128 bytes, ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0)
struct NcclId { char internal[128]; };
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(void**, int, NcclId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);
struct RcclApi {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    std::string err;
};
std::mutex g_rccl_mu;
RcclApi g_rccl;

const RcclApi* rccl(std::string& err) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return &g_rccl;
    if (!g_rccl.err.empty()) { err = g_rccl.err; return nullptr; }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    const char* only = getenv("FEASTHIP_RCCL_LIB");      // a specific library (or, in the tests, a name that does not exist)
    if (only && *only) lib = dlopen(only, RTLD_NOW | RTLD_LOCAL);
    else for (const char* nm : names) { lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) {
        const char* de = dlerror();      // one call: dlerror() clears the message it returns
        g_rccl.err = std::string("RCCL not found (dlopen librccl.so.1): ") + (de ? de : "");
        err = g_rccl.err;
        return nullptr;
    }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(lib, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(lib, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(lib, "ncclCommDestroy");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(lib, "ncclAllReduce");
    g_rccl.error_string = (fn_error_string)dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce) {
        g_rccl.err = "RCCL: a symbol of the collective API is missing"; err = g_rccl.err; dlclose(lib); return nullptr;
    }
    g_rccl.lib = lib;
    return &g_rccl;
}

std::string rccl_msg(const RcclApi* api, const char* what, int code) {
    return std::string(what) + ": " + (api && api->error_string ? api->error_string(code) : "RCCL error") + " (" + std::to_string(code) + ")";
}
}  // namespace

int Engine::comm_unique_id(void* id128, std::string& err) {
    const RcclApi* api = rccl(err);
    if (!api) return FEASTHIP_ECOMM;
    NcclId id;
    const int rc = api->get_unique_id(&id);
    if (rc != 0) { err = rccl_msg(api, "ncclGetUniqueId", rc); return FEASTHIP_ECOMM; }
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int Engine::comm_init(const void* id128, int rank, int world) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(FEASTHIP_EARG, "comm_init: bad rank / world");
    if (comm_) { int r = comm_destroy(); if (r) return r; }
    HIPCHK(hipSetDevice(device_));
    std::string e;
    const RcclApi* api = rccl(e);
    if (!api) return fail(FEASTHIP_ECOMM, e);
    NcclId id;
    std::memcpy(&id, id128, sizeof(id));
    void* comm = nullptr;
    const int rc = api->comm_init_rank(&comm, world, id, rank);
    if (rc != 0 || !comm) return fail(FEASTHIP_ECOMM, rccl_msg(api, "ncclCommInitRank", rc));
    if (!comm_stream_) HIPCHK(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
    if (!ev_comm0_) HIPCHK(hipEventCreate(&ev_comm0_));
    if (!ev_comm1_) HIPCHK(hipEventCreate(&ev_comm1_));
    comm_ = comm; comm_rank_ = rank; comm_world_ = world;
    return 0;
}

int Engine::comm_destroy() {
    if (!comm_) return 0;
    (void)hipSetDevice(device_);
    if (comm_stream_) (void)hipStreamSynchronize(comm_stream_);
    std::string e;
    const RcclApi* api = rccl(e);
    int rc = api ? api->comm_destroy(comm_) : 0;
    comm_ = nullptr; comm_world_ = 1; comm_rank_ = 0;
    if (rc != 0) return fail(FEASTHIP_ECOMM, rccl_msg(api, "ncclCommDestroy", rc));
    return 0;
}

// rows [row0, row0 + rows) of the moment (original row order, n x ldw complex128 = 2 ldw doubles per row), summed in place over the
// ranks on the communication stream; the caller has made that stream wait for the event of the rows' accumulation
int Engine::comm_allreduce_rows(int row0, int rows) {
    if (!comm_ || rows <= 0) return 0;
    std::string e;
    const RcclApi* api = rccl(e);
    if (!api) return fail(FEASTHIP_ECOMM, e);
    cz* q = Qp() + (size_t)row0 * ldw;
    const int rc = api->all_reduce(q, q, (size_t)rows * ldw * 2, /* ncclFloat64 */ 8, /* ncclSum */ 0, comm_, comm_stream_);
    if (rc != 0) return fail(FEASTHIP_ECOMM, rccl_msg(api, "ncclAllReduce", rc));
    return 0;
}

}  // namespace feasthip
