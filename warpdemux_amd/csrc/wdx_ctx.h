// The opaque context behind include/wdx.h's wdx_ctx (internal; shared by wdx_api.hip, wdx_comm.hip
// and wdx_live.hip).  Nothing here computes results.
#pragma once
#include "wdx_common.h"

#include <mutex>
#include <utility>
#include <vector>

namespace wdx {

struct Buffer {  // grow-only device workspace
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);
    void release();
};

struct PinnedBuffer {  // grow-only page-locked host staging area
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);
    void release();
};

constexpr int kNumTimed = 9;

// RAII: make the context's device current for the duration of one entry point and put the caller's
// device back afterwards (a host thread that also drives torch must not find its device switched).
struct DeviceGuard {
    int prev = -1;
    int rc = WDX_SUCCESS;
    explicit DeviceGuard(int device);
    ~DeviceGuard();
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

struct Comm;  // wdx_comm.hip: RCCL communicator + the dlopen'ed entry points

}  // namespace wdx

struct wdx_ctx {
    int device = 0;
    std::mutex mu;
    hipStream_t stream = nullptr;  // the context's own non-blocking stream: every host-buffer call runs on it
    hipStream_t last_stream = nullptr;  // stream of the latest enqueue that used the shared workspaces
    bool last_stream_valid = false;
    wdx::Knobs knobs;
    wdx::DtwRefs refs;
    wdx::Buffer refs_pad, refs_T, refs_nan;
    // host-buffer call workspaces
    wdx::Buffer in0, in1, in2, in3, out0, out1, out2, out3, tmp0, tmp1, tmp2, scratch, fp_ws, svm_buf, ref_buf;
    wdx::Buffer ref_ws;  // refinement branch: the fast kernels' hand-over records (fingerprint_refine_ws_bytes)
    wdx::Buffer fp_big;  // score curves of adapter windows beyond the exact kernel's LDS capacity (fingerprint_big_bytes)
    wdx::PinnedBuffer pin_in, pin_out;  // staging of small (live-tick sized) host-buffer calls
    std::vector<double> ref_query_host;  // the consensus query resident in ref_buf (wdx_fingerprint_refine_dev uploads on change)
    wdx::Buffer pk_idx;       // packed staging of a page-locked minibatch: window offsets / first columns / shifted bounds
    wdx::PinnedBuffer pk_host;  // ... and their host images (kept until the slot's copy has run)
    int64_t refs_gen = 0;  // bumped whenever the resident reference set (samples or window/penalty) changes
    wdx::SvmDev svm{};
    bool svm_set = false;
    // fused DTW + SVM path of wdx_demux_svm_dev: vector-major coefficients and the chunk tables (built with the model),
    // the resident references gathered into support-vector order (rebuilt when the reference set or the model changes)
    wdx::Buffer svm_fused, svm_refs;
    const double *svm_coefT = nullptr;
    const int32_t *svm_chunk_ref0 = nullptr, *svm_chunk_slot = nullptr;
    int svm_chunks = 0, svm_halves = 0;
    int64_t svm_refs_gen = -1, svm_model_gen = 0, svm_refs_model_gen = -1;
    wdx::Comm *comm = nullptr;
    // pipelined minibatches (wdx_demux_submit / wdx_demux_wait): up to WDX_MAX_SLOTS child contexts, each with its own stream and
    // workspaces, sharing this context's resident reference set; the fields below describe a child's batch in flight
    wdx_ctx *slots[WDX_MAX_SLOTS] = {};
    bool slot_busy = false, slot_waiting = false;
    uint32_t slot_want = 0;  // WDX_WANT_* of the batch in flight
    int64_t slot_n = 0, slot_K = 0, slot_nY = 0, slot_k = 0;
    size_t slot_off[9] = {};  // fpt, dwell, stats, prob, conf, dist, call, status, pred in the slot's page-locked block
    wdx::Buffer mb_dwell, mb_stats, mb_prob, mb_pred, mb_conf;  // device side of the optional minibatch outputs
    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[wdx::kNumTimed];
    std::vector<int64_t> pending_launches[wdx::kNumTimed];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double acc_ms[wdx::kNumTimed] = {};
    int64_t launches[wdx::kNumTimed] = {};
};

namespace wdx {

struct Timed {  // RAII: hipEvents around one kernel launch when timing is on
    wdx_ctx *c;
    int id;
    hipStream_t s;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    int64_t n_launches = 0;  // kernel launches bracketed by this event pair (0 -> counted as 1)
    MainEvents main;         // WDX_K_FINGERPRINT only: a second pair around the main fast-kernel launches alone
    Timed(wdx_ctx *c_, int id_, hipStream_t s_);
    ~Timed();
};

// Entry-point prologue: null check.  (The device is made current by a DeviceGuard in the caller.)
int check_ctx(wdx_ctx *ctx);
// The shared workspaces of a context are ordered by stream order only.  Call with the mutex held before
// enqueueing on `s`: when the previous user enqueued on a different stream, wait for that stream first.
int use_stream(wdx_ctx *ctx, hipStream_t s);
void comm_destroy(wdx_ctx *ctx);
// (wdx_api.hip) with the context's mutex held:
// (re)build the resident reference set from a HOST array (content-hashed: uploads only on change)
int set_refs_locked(wdx_ctx *ctx, const double *Y, int64_t nY, int64_t L, int32_t window, double penalty,
                    hipStream_t stream);
// DTW of device rows dX (nX, L) against the resident refs -> d_out (nX, nY) [+ argmin]
int dtw_dev_locked(wdx_ctx *ctx, const double *dX, int64_t nX, float *d_out, int32_t *d_argmin,
                   hipStream_t stream);

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

}  // namespace wdx

// Host-buffer entry points enqueue copies from/to the CALLER's (or the context's page-locked) memory and
// synchronise once at the end.  A failure half way must not return while such a copy may still be in flight
// (the caller frees or rewrites its buffers; PinnedBuffer::ensure may hipHostFree the staging block on the next
// call): every exit of the enqueue region drains the stream unless the normal final synchronisation already ran.
struct StreamDrain {
    hipStream_t s;
    bool armed = true;
    explicit StreamDrain(hipStream_t s_) : s(s_) {}
    void done() { armed = false; }
    ~StreamDrain() {
        if (armed) (void)hipStreamSynchronize(s);  // best effort: the error already recorded is the one reported
    }
};

#define WDX_ENTER(ctx)                                  \
    if (int _e = ::wdx::check_ctx(ctx)) return _e;      \
    ::wdx::DeviceGuard _guard((ctx)->device);           \
    if (_guard.rc) return _guard.rc;                    \
    int rc = WDX_SUCCESS;                               \
    (void)rc
