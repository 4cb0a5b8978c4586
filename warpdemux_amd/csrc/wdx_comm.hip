// Multi-GPU exchange of the path: ONE sum all-reduce of the int64 call histogram over
// RCCL (xGMI within a node).  The reference has no analogue -- its workers bump Manager() counters
// (file_proc.py:1059-1066); here every rank counts its own shard and the totals meet once per job.
//
// librccl is dlopen'ed on first use: (1) the copy already mapped into the process (PyTorch bundles one and
// a second RCCL in one process would carry its own bootstrap/IPC state), (2) $ROCM's librccl.so.1 through
// the RUNPATH of this library.  Only the five entry points below are bound; the communicator type is opaque.
#include "wdx_ctx.h"

#include <dlfcn.h>
#include <string.h>

namespace wdx {

namespace {

// the slice of rccl.h this file needs (values are ABI constants of NCCL/RCCL)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[WDX_COMM_ID_BYTES];
} ncclUniqueId;
constexpr int kNcclSuccess = 0;
constexpr int kNcclInt64 = 4;
constexpr int kNcclSum = 0;

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;      // optional: what RCCL itself says the world is
    int (*CommUserRank)(const ncclComm_t, int *) = nullptr;   // optional
    bool ok = false;
    char why[256] = "";
};

Rccl &rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if (!R.handle) R.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // already in the process?
        for (const char *n : names)
            if (!R.handle) R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!R.handle) {
            snprintf(R.why, sizeof(R.why), "librccl not found: %s", dlerror());
            return;
        }
        R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.handle, "ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.handle, "ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.handle, "ncclCommDestroy");
        R.AllReduce = (decltype(R.AllReduce))dlsym(R.handle, "ncclAllReduce");
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.handle, "ncclGetErrorString");
        R.CommCount = (decltype(R.CommCount))dlsym(R.handle, "ncclCommCount");
        R.CommUserRank = (decltype(R.CommUserRank))dlsym(R.handle, "ncclCommUserRank");
        R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.AllReduce && R.GetErrorString;
        if (!R.ok) snprintf(R.why, sizeof(R.why), "librccl lacks an expected entry point");
    });
    return R;
}

int rccl_ready() {
    Rccl &R = rccl();
    if (!R.ok) {
        set_error("RCCL unavailable: %s", R.why);
        return WDX_ERR_NO_DEVICE;
    }
    return WDX_SUCCESS;
}

#define WDX_RCCL_TRY(expr)                                                                       \
    do {                                                                                         \
        int _r = (expr);                                                                         \
        if (_r != kNcclSuccess) {                                                                \
            set_error("%s failed: %s", #expr, rccl().GetErrorString(_r));                        \
            return WDX_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

void comm_destroy(wdx_ctx *ctx) {
    if (!ctx->comm) return;
    if (ctx->comm->comm && rccl().ok) (void)rccl().CommDestroy(ctx->comm->comm);
    delete ctx->comm;
    ctx->comm = nullptr;
}

}  // namespace wdx

using namespace wdx;

extern "C" {

int wdx_comm_available(void) { return rccl_ready(); }

int wdx_comm_info(wdx_ctx *ctx, int32_t *rank, int32_t *world, int32_t *rccl_count) {
    WDX_ENTER(ctx);
    std::lock_guard<std::mutex> g(ctx->mu);
    int32_t r = 0, w = 1, cnt = 0;
    if (ctx->comm) {
        r = ctx->comm->rank;
        w = ctx->comm->world;
        cnt = -1;
        int v = 0;
        if (rccl().CommCount && rccl().CommCount(ctx->comm->comm, &v) == kNcclSuccess) cnt = v;
        if (rccl().CommUserRank && rccl().CommUserRank(ctx->comm->comm, &v) == kNcclSuccess && v != r) {
            set_error("comm_info: RCCL reports rank %d, the context was bound as rank %d", v, r);
            return WDX_ERR_HIP;
        }
    }
    if (rank) *rank = r;
    if (world) *world = w;
    if (rccl_count) *rccl_count = cnt;
    return WDX_SUCCESS;
}

int wdx_comm_unique_id(void *id_out) {
    if (!id_out) {
        set_error("comm_unique_id: null output");
        return WDX_ERR_INVALID;
    }
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId id;
    WDX_RCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id_out, id.internal, WDX_COMM_ID_BYTES);
    return WDX_SUCCESS;
}

int wdx_comm_init(wdx_ctx *ctx, const void *id, int32_t rank, int32_t world) {
    WDX_ENTER(ctx);
    if (!id || world < 1 || rank < 0 || rank >= world) {
        set_error("comm_init: need an id and 0 <= rank < world");
        return WDX_ERR_INVALID;
    }
    if ((rc = rccl_ready())) return rc;
    std::lock_guard<std::mutex> g(ctx->mu);
    comm_destroy(ctx);
    ncclUniqueId uid;
    memcpy(uid.internal, id, WDX_COMM_ID_BYTES);
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    int r = rccl().CommInitRank(&c->comm, world, uid, rank);
    if (r != kNcclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl().GetErrorString(r));
        delete c;
        return WDX_ERR_HIP;
    }
    ctx->comm = c;
    return WDX_SUCCESS;
}

int wdx_comm_destroy(wdx_ctx *ctx) {
    WDX_ENTER(ctx);
    std::lock_guard<std::mutex> g(ctx->mu);
    comm_destroy(ctx);
    return WDX_SUCCESS;
}

int wdx_reduce_counts(wdx_ctx *ctx, int64_t *d_counts, int32_t n, void *stream) {
    WDX_ENTER(ctx);
    if (n < 0 || (n > 0 && !d_counts)) {
        set_error("reduce_counts: bad arguments");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->comm || n == 0) return WDX_SUCCESS;
    Timed t(ctx, WDX_K_REDUCE, (hipStream_t)stream);
    WDX_RCCL_TRY(rccl().AllReduce(d_counts, d_counts, (size_t)n, kNcclInt64, kNcclSum, ctx->comm->comm,
                                  (hipStream_t)stream));
    return WDX_SUCCESS;
}

int wdx_reduce_counts_host(wdx_ctx *ctx, int64_t *counts, int32_t n) {
    WDX_ENTER(ctx);
    if (n < 0 || (n > 0 && !counts)) {
        set_error("reduce_counts_host: bad arguments");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->comm || n == 0) return WDX_SUCCESS;
    hipStream_t s = ctx->stream;
    if ((rc = use_stream(ctx, s))) return rc;
    if ((rc = ctx->tmp0.ensure((size_t)n * 8))) return rc;
    WDX_HIP_TRY(hipMemcpyAsync(ctx->tmp0.p, counts, (size_t)n * 8, hipMemcpyHostToDevice, s));
    WDX_RCCL_TRY(rccl().AllReduce(ctx->tmp0.p, ctx->tmp0.p, (size_t)n, kNcclInt64, kNcclSum, ctx->comm->comm, s));
    WDX_HIP_TRY(hipMemcpyAsync(counts, ctx->tmp0.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    WDX_HIP_TRY(hipStreamSynchronize(s));
    return WDX_SUCCESS;
}

}  // extern "C"
