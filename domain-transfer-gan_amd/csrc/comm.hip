// acg_comm_*: the gradient exchange of the data-parallel step as four C entry points over RCCL, for hosts
// that do not run torch.distributed.  RCCL is bound at run time (dlopen): a process that already holds a librccl (PyTorch
// loads its own) keeps using that one instance; otherwise ACG_RCCL_LIB, librccl.so.1 or librccl.so is opened.  Nothing else
// in this library depends on RCCL.
#include <dlfcn.h>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace {
struct IdBlob { char b[ACG_COMM_ID_BYTES]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef int (*get_id_fn)(IdBlob *);
typedef int (*init_fn)(void **, int, IdBlob, int);
typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*destroy_fn)(void *);
typedef const char *(*errstr_fn)(int);
struct Rccl {
    get_id_fn get_id = nullptr;
    init_fn init = nullptr;
    allreduce_fn allreduce = nullptr;
    destroy_fn destroy = nullptr;
    errstr_fn errstr = nullptr;
    bool ok = false;
} g_rccl;
std::once_flag g_once;
constexpr int NCCL_FLOAT32 = 7, NCCL_AVG = 4; // ncclDataType_t / ncclRedOp_t values of rccl.h

void bind()
{
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);   // the instance the process already uses, if any
    const char *env = getenv("ACG_RCCL_LIB");
    if (h == nullptr && env != nullptr) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) return;
    g_rccl.get_id = (get_id_fn)dlsym(h, "ncclGetUniqueId");
    g_rccl.init = (init_fn)dlsym(h, "ncclCommInitRank");
    g_rccl.allreduce = (allreduce_fn)dlsym(h, "ncclAllReduce");
    g_rccl.destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (errstr_fn)dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.get_id && g_rccl.init && g_rccl.allreduce && g_rccl.destroy;
}
int need(const char *who)
{
    std::call_once(g_once, bind);
    if (!g_rccl.ok) {
        acg_set_error("%s: RCCL not available (librccl.so not loadable; set ACG_RCCL_LIB)", who);
        return ACG_ERR_COMM;
    }
    return ACG_OK;
}
int check(int rc, const char *who)
{
    if (rc == 0) return ACG_OK;
    acg_set_error("%s: RCCL error %d (%s)", who, rc, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    return ACG_ERR_COMM;
}
}

extern "C" int acg_comm_unique_id(void *id)
{
    ACG_REQUIRE(id != nullptr, "acg_comm_unique_id: null");
    int rc = need("acg_comm_unique_id");
    if (rc) return rc;
    return check(g_rccl.get_id((IdBlob *)id), "acg_comm_unique_id");
}

extern "C" int acg_comm_init(void **comm, const void *id, int nranks, int rank)
{
    ACG_REQUIRE(comm != nullptr && id != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "acg_comm_init: bad arguments (nranks=%d rank=%d)",
                nranks, rank);
    int rc = need("acg_comm_init");
    if (rc) return rc;
    IdBlob blob;
    memcpy(&blob, id, sizeof(blob));
    return check(g_rccl.init(comm, nranks, blob, rank), "acg_comm_init");
}

extern "C" int acg_comm_allreduce_mean(void *comm, float *buf, size_t n, void *stream)
{
    ACG_REQUIRE(comm != nullptr && buf != nullptr && n > 0, "acg_comm_allreduce_mean: bad arguments");
    int rc = need("acg_comm_allreduce_mean");
    if (rc) return rc;
    return check(g_rccl.allreduce(buf, buf, n, NCCL_FLOAT32, NCCL_AVG, comm, (hipStream_t)stream), "acg_comm_allreduce_mean");
}

extern "C" int acg_comm_destroy(void *comm)
{
    if (comm == nullptr) return ACG_OK;
    int rc = need("acg_comm_destroy");
    if (rc) return rc;
    return check(g_rccl.destroy(comm), "acg_comm_destroy");
}
