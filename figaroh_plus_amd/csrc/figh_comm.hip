// Multi-GPU exchange for the sample-sharded path: one process per GPU, RCCL over xGMI.
// The only data that crosses GPUs is O(n^2): each rank's R factor (all-gather, then every rank reduces the
// stack redundantly with figh_tsqr_merge) and the column-norm / residual sums (all-reduce).  librccl is
// opened lazily so single-GPU users never load it.
#include <dlfcn.h>

#include <cstring>

#include "figh_internal.h"

namespace figh {

typedef struct { char internal[128]; } UniqueId;
typedef void *Comm;
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(Comm *, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllGatherFn)(const void *, void *, size_t, int, Comm, hipStream_t);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);

static void *g_rccl = nullptr;
static GetUniqueIdFn p_unique = nullptr;
static CommInitRankFn p_init = nullptr;
static CommDestroyFn p_destroy = nullptr;
static AllGatherFn p_allgather = nullptr;
static AllReduceFn p_allreduce = nullptr;
static GetErrorStringFn p_errstr = nullptr;
static Comm g_comm = nullptr;
static int g_nranks = 0;

constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0;      // ncclSum

static int load_rccl() {
    if (g_rccl) return FIGH_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        g_rccl = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl) break;
    }
    if (!g_rccl) {
        set_error(std::string("cannot load librccl: ") + dlerror());
        return FIGH_ERR_COMM;
    }
    p_unique = (GetUniqueIdFn)dlsym(g_rccl, "ncclGetUniqueId");
    p_init = (CommInitRankFn)dlsym(g_rccl, "ncclCommInitRank");
    p_destroy = (CommDestroyFn)dlsym(g_rccl, "ncclCommDestroy");
    p_allgather = (AllGatherFn)dlsym(g_rccl, "ncclAllGather");
    p_allreduce = (AllReduceFn)dlsym(g_rccl, "ncclAllReduce");
    p_errstr = (GetErrorStringFn)dlsym(g_rccl, "ncclGetErrorString");
    if (!p_unique || !p_init || !p_destroy || !p_allgather || !p_allreduce) {
        set_error("librccl is missing a required symbol");
        return FIGH_ERR_COMM;
    }
    return FIGH_OK;
}

static int check(int rc, const char *what) {
    if (rc == 0) return FIGH_OK;
    set_error(std::string(what) + ": " + (p_errstr ? p_errstr(rc) : "rccl error"));
    return FIGH_ERR_COMM;
}

}  // namespace figh

using namespace figh;

extern "C" {

int figh_comm_available(void) {
    if (int rc = ensure_device()) return rc;
    return load_rccl();
}

int figh_comm_unique_id(void *h_id128) {
    FIGH_REQUIRE(h_id128, "h_id128 is NULL");
    if (int rc = load_rccl()) return rc;
    UniqueId id;
    if (int rc = check(p_unique(&id), "ncclGetUniqueId")) return rc;
    std::memcpy(h_id128, &id, sizeof(id));
    return FIGH_OK;
}

int figh_comm_init(int nranks, int rank, const void *h_id128) {
    FIGH_REQUIRE(h_id128 && nranks >= 1 && rank >= 0 && rank < nranks, "bad communicator arguments");
    FIGH_REQUIRE(!g_comm, "communicator already initialised");
    if (int rc = ensure_device()) return rc;
    if (load_rccl()) return FIGH_ERR_UNSUPPORTED;  // (FIGH_ERR_COMM is reserved for what the rendezvous itself reports)
    UniqueId id;
    std::memcpy(&id, h_id128, sizeof(id));
    if (int rc = check(p_init(&g_comm, nranks, id, rank), "ncclCommInitRank")) return rc;
    g_nranks = nranks;
    return FIGH_OK;
}

int figh_comm_destroy(void) {
    if (!g_comm) return FIGH_OK;
    (void)hipStreamSynchronize(stream());
    int rc = check(p_destroy(g_comm), "ncclCommDestroy");
    g_comm = nullptr;
    g_nranks = 0;
    return rc;
}

int figh_comm_allgather(const double *d_send, double *d_all, int64_t count_per_rank) {
    FIGH_REQUIRE(g_comm, "communicator not initialised");
    FIGH_REQUIRE(d_send && d_all && count_per_rank > 0, "bad all-gather arguments");
    ProfileScope scope("rccl_allgather");
    return check(p_allgather(d_send, d_all, (size_t)count_per_rank, kNcclFloat64, g_comm, stream()), "ncclAllGather");
}

int figh_comm_allreduce_sum(double *d_buf, int64_t count) {
    FIGH_REQUIRE(g_comm, "communicator not initialised");
    FIGH_REQUIRE(d_buf && count > 0, "bad all-reduce arguments");
    ProfileScope scope("rccl_allreduce");
    return check(p_allreduce(d_buf, d_buf, (size_t)count, kNcclFloat64, kNcclSum, g_comm, stream()), "ncclAllReduce");
}

}  // extern "C"
