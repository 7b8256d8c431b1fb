// The path's one collective behind the C ABI: an in-place sum over the ranks of one node through RCCL (xGMI), for a
// caller that binds this library without torch.distributed.  What it reduces: the per-resolution predictions that the
// reference adds up in a Python loop (src/MRGP.py:802-803, variance :902-905) -- here the fused [mean | var] buffer of
// the ranks' blocks -- and the per-layer training-point predictions of the residual chain (src/Stats.py:126-157).
//
// RCCL is loaded at the first use (dlopen: librccl.so.1, then librccl.so), so the library itself has no link-time
// dependency on it and a process that never reduces never loads it.  Payloads are a few MiB: latency-bound on the
// point-to-point xGMI links, one call per exchange, no bucketing.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.hpp"

namespace {

// the slice of RCCL's C API used here (rccl.h: NCCL_UNIQUE_ID_BYTES = 128, ncclSum = 0, ncclFloat32 = 7, ncclFloat64 = 8)
struct UniqueId { char internal[CIMRGP_COMM_ID_BYTES]; };
typedef void* Comm;
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_comm_destroy)(Comm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

const Rccl* rccl()
{
    std::lock_guard<std::mutex> guard(g_rccl_mutex);
    if (!g_rccl.tried) {
        g_rccl.tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.handle) break;
        }
        if (g_rccl.handle) {
            g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.handle, "ncclGetUniqueId");
            g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.handle, "ncclCommInitRank");
            g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.handle, "ncclCommDestroy");
            g_rccl.all_reduce = (fn_all_reduce)dlsym(g_rccl.handle, "ncclAllReduce");
            g_rccl.error_string = (fn_error_string)dlsym(g_rccl.handle, "ncclGetErrorString");
        }
    }
    const bool ok = g_rccl.handle && g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce;
    return ok ? &g_rccl : nullptr;
}

int rccl_fail(const Rccl* r, const char* fn, const char* what, int code)
{
    std::string msg = std::string(what) + ": " + ((r && r->error_string) ? r->error_string(code) : "RCCL error");
    return cimrgp::fail(fn, msg.c_str());
}

struct CommHandle { Comm comm; int world, rank; };

}  // namespace

extern "C" {

int cimrgp_comm_unique_id(void* id_out)
{
    const char* fn = "cimrgp_comm_unique_id";
    CIMRGP_REQUIRE(id_out != nullptr, fn, "null pointer");
    const Rccl* r = rccl();
    CIMRGP_REQUIRE(r != nullptr, fn, "librccl.so could not be loaded");
    UniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return rccl_fail(r, fn, "ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, CIMRGP_COMM_ID_BYTES);
    return 0;
}

int cimrgp_comm_create(int world_size, int rank, const void* id, void** comm_out)
{
    const char* fn = "cimrgp_comm_create";
    CIMRGP_REQUIRE(id != nullptr && comm_out != nullptr, fn, "null pointer");
    CIMRGP_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, fn, "rank outside the world");
    const Rccl* r = rccl();
    CIMRGP_REQUIRE(r != nullptr, fn, "librccl.so could not be loaded");
    UniqueId uid;
    memcpy(uid.internal, id, CIMRGP_COMM_ID_BYTES);
    Comm c = nullptr;
    const int rc = r->comm_init_rank(&c, world_size, uid, rank);          // binds the CURRENT HIP device of this process
    if (rc != 0) return rccl_fail(r, fn, "ncclCommInitRank", rc);
    *comm_out = new CommHandle{c, world_size, rank};
    return 0;
}

int cimrgp_comm_destroy(void* comm)
{
    const char* fn = "cimrgp_comm_destroy";
    CIMRGP_REQUIRE(comm != nullptr, fn, "null pointer");
    const Rccl* r = rccl();
    CIMRGP_REQUIRE(r != nullptr, fn, "librccl.so could not be loaded");
    CommHandle* h = static_cast<CommHandle*>(comm);
    const int rc = r->comm_destroy(h->comm);
    delete h;
    if (rc != 0) return rccl_fail(r, fn, "ncclCommDestroy", rc);
    return 0;
}

int cimrgp_allreduce_sum(void* comm, int dtype, void* buf_dev, int64_t count, void* stream)
{
    const char* fn = "cimrgp_allreduce_sum";
    CIMRGP_REQUIRE(comm != nullptr, fn, "null communicator");
    CIMRGP_REQUIRE(dtype == CIMRGP_F32 || dtype == CIMRGP_F64, fn, "unknown dtype");
    CIMRGP_REQUIRE(count >= 0 && (buf_dev != nullptr || count == 0), fn, "null pointer");
    if (count == 0) return 0;
    const Rccl* r = rccl();
    CIMRGP_REQUIRE(r != nullptr, fn, "librccl.so could not be loaded");
    CommHandle* h = static_cast<CommHandle*>(comm);
    const int rc = r->all_reduce(buf_dev, buf_dev, (size_t)count, dtype == CIMRGP_F64 ? 8 : 7, 0, h->comm,
                                 reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) return rccl_fail(r, fn, "ncclAllReduce", rc);
    return 0;
}

}  // extern "C"
