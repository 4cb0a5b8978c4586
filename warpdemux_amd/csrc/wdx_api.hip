// C ABI of libwdx_hip.so (include/wdx.h): context, workspaces, host<->device plumbing.
// The arithmetic lives in wdx_dtw.hip / wdx_fingerprint.hip; nothing here computes results.
#include "wdx_ctx.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace wdx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local int64_t g_slice_cap = 0;   // WDX_OPT_MAX_LAUNCH_SLICE of the call running on this thread (0 = none)

int64_t launch_slice_limit(int64_t builtin) { return g_slice_cap > 0 && g_slice_cap < builtin ? g_slice_cap : builtin; }
LaunchSliceScope::LaunchSliceScope(int64_t cap) : saved(g_slice_cap) { g_slice_cap = cap; }
LaunchSliceScope::~LaunchSliceScope() { g_slice_cap = saved; }

// content hash of the reference set (cache key): 64-bit words, multiply-xorshift mixing
static uint64_t fnv1a(const void *data, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
    const unsigned char *p = (const unsigned char *)data;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    for (; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

int Buffer::ensure(size_t need) {
    if (need <= bytes) return WDX_SUCCESS;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    size_t want = need + need / 4;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        e = hipMalloc(&p, need);
        want = need;
    }
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        p = nullptr;
        return WDX_ERR_HIP;
    }
    bytes = want;
    return WDX_SUCCESS;
}

void Buffer::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

int PinnedBuffer::ensure(size_t need) {
    if (need <= bytes) return WDX_SUCCESS;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
    const size_t want = need + need / 4;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        p = nullptr;
        return WDX_ERR_HIP;
    }
    bytes = want;
    return WDX_SUCCESS;
}

void PinnedBuffer::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
}

DeviceGuard::DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) {
            set_error("hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
            rc = WDX_ERR_HIP;
            prev = -1;
        }
    } else {
        prev = -1;  // nothing to restore
    }
}

DeviceGuard::~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
}

// one event pair from the context's pool, or two fresh events -- both or neither (a failed second create must not leak
// the first)
static std::pair<hipEvent_t, hipEvent_t> take_event_pair(wdx_ctx *c) {
    std::pair<hipEvent_t, hipEvent_t> q{nullptr, nullptr};
    if (!c->pool.empty()) {
        q = c->pool.back();
        c->pool.pop_back();
        return q;
    }
    if (hipEventCreate(&q.first) != hipSuccess) return {nullptr, nullptr};
    if (hipEventCreate(&q.second) != hipSuccess) {
        (void)hipEventDestroy(q.first);
        return {nullptr, nullptr};
    }
    return q;
}

Timed::Timed(wdx_ctx *c_, int id_, hipStream_t s_) : c(c_), id(id_), s(s_) {
    if (!c->timing) return;
    ev = take_event_pair(c);
    if (!ev.first) return;
    (void)hipEventRecord(ev.first, s);
    if (id == WDX_K_FINGERPRINT) {  // a second pair for the main fast-kernel launches (recorded by launch_fingerprint)
        const auto m = take_event_pair(c);
        main.first = m.first;
        main.second = m.second;
        const auto q = take_event_pair(c);   // ... and a third around clip_bounds_kernel
        main.c_first = q.first;
        main.c_second = q.second;
        // (the split main kernel asks for one more pair per tail-kernel launch)
        main.take = [](void *arg) { return take_event_pair(static_cast<wdx_ctx *>(arg)); };
        main.take_arg = c;
    }
}

Timed::~Timed() {
    if (!ev.first) return;
    (void)hipEventRecord(ev.second, s);
    c->pending[id].push_back(ev);
    c->pending_launches[id].push_back(n_launches > 0 ? n_launches : 1);
    if (main.first) {
        if (main.recorded) {
            c->pending[WDX_K_FINGERPRINT_MAIN].push_back({main.first, main.second});
            c->pending_launches[WDX_K_FINGERPRINT_MAIN].push_back(n_launches > 0 ? n_launches : 1);
        } else {
            c->pool.push_back({main.first, main.second});
        }
    }
    for (const auto &tp : main.tail) {
        c->pending[WDX_K_FINGERPRINT_TAIL].push_back(tp);
        c->pending_launches[WDX_K_FINGERPRINT_TAIL].push_back(1);
    }
    if (main.c_first) {
        if (main.c_recorded) {
            c->pending[WDX_K_FINGERPRINT_CLIP].push_back({main.c_first, main.c_second});
            c->pending_launches[WDX_K_FINGERPRINT_CLIP].push_back(1);
        } else {
            c->pool.push_back({main.c_first, main.c_second});
        }
    }
}

int check_ctx(wdx_ctx *ctx) {
    if (!ctx) {
        set_error("null context");
        return WDX_ERR_INVALID;
    }
    return WDX_SUCCESS;
}

int use_stream(wdx_ctx *ctx, hipStream_t s) {
    if (ctx->last_stream_valid && ctx->last_stream != s) WDX_HIP_TRY(hipStreamSynchronize(ctx->last_stream));
    ctx->last_stream = s;
    ctx->last_stream_valid = true;
    return WDX_SUCCESS;
}

}  // namespace wdx

using namespace wdx;

namespace wdx {

// (re)build the resident reference set from a HOST array
int set_refs_locked(wdx_ctx *ctx, const double *Y, int64_t nY, int64_t L, int32_t window,
                    double penalty, hipStream_t stream) {
    if (nY < 0 || L <= 0 || (nY > 0 && !Y)) {
        set_error("set_refs: need nY >= 0, L > 0 and a non-null Y");
        return WDX_ERR_INVALID;
    }
    if (penalty != penalty) {
        set_error("set_refs: penalty is NaN");
        return WDX_ERR_INVALID;
    }
    const int w_eff = (window <= 0 || window > L) ? (int)L : window;
    uint64_t h = fnv1a(Y, (size_t)(nY * L) * sizeof(double));
    h = fnv1a(&nY, sizeof(nY), h);
    h = fnv1a(&L, sizeof(L), h);
    DtwRefs &R = ctx->refs;
    if (R.window != 0 && R.content_hash == h && R.nY == nY && R.L == L) {
        if (R.window != w_eff || R.penalty != penalty) ++ctx->refs_gen;
        R.window = w_eff;  // same samples: only the scalars may have changed
        R.penalty = penalty;
        return WDX_SUCCESS;
    }
    ++ctx->refs_gen;
    // a pipelined minibatch (wdx_demux_submit) may still be reading the set that is about to be rebuilt
    for (wdx_ctx *S : ctx->slots)
        if (S) WDX_HIP_TRY(hipStreamSynchronize(S->stream));
    // not set until every buffer below is rebuilt: a failure in the middle must not leave a stale hash over
    // freed or half-written buffers (a later call then reports WDX_ERR_NO_REFS instead of reading them)
    R.window = 0;
    R.content_hash = 0;
    const int halo = kMaxRegWindow - 1;
    const int64_t Lpad = L + 2 * halo;
    const int64_t ldT = round_up(nY > 0 ? nY : 1, 64);
    int rc;
    if ((rc = ctx->refs_pad.ensure((size_t)(nY > 0 ? nY : 1) * Lpad * sizeof(double)))) return rc;
    if ((rc = ctx->refs_T.ensure((size_t)L * ldT * sizeof(double)))) return rc;
    if ((rc = ctx->refs_nan.ensure((size_t)ldT))) return rc;
    if ((rc = ctx->tmp0.ensure((size_t)(nY > 0 ? nY : 1) * L * sizeof(double)))) return rc;
    if (nY > 0) {
        WDX_HIP_TRY(hipMemcpyAsync(ctx->tmp0.p, Y, (size_t)(nY * L) * sizeof(double),
                                   hipMemcpyHostToDevice, stream));
        WDX_HIP_TRY(hipMemsetAsync(ctx->refs_T.p, 0, (size_t)L * ldT * sizeof(double), stream));
        if ((rc = launch_pad_rows((const double *)ctx->tmp0.p, nY, L, (double *)ctx->refs_pad.p,
                                  Lpad, halo, (uint8_t *)ctx->refs_nan.p, stream)))
            return rc;
        if ((rc = launch_transpose((const double *)ctx->tmp0.p, nY, L, (double *)ctx->refs_T.p, ldT,
                                   nullptr, stream)))
            return rc;
        WDX_HIP_TRY(hipStreamSynchronize(stream));  // tmp0 is reused by later calls
    }
    R.pad = (double *)ctx->refs_pad.p;
    R.T = (double *)ctx->refs_T.p;
    R.has_nan = (uint8_t *)ctx->refs_nan.p;
    R.nY = nY;
    R.L = L;
    R.Lpad = Lpad;
    R.ldT = ldT;
    R.halo = halo;
    R.window = w_eff;
    R.penalty = penalty;
    R.content_hash = h;
    return WDX_SUCCESS;
}

// Batches from this size on go through the DTW kernel's row-major form (no transposed copy of the fingerprints:
// at 10 M reads the copy is 17.6 GB of traffic and 6 ms); below it the launch is latency-bound and the
// read-minor copy (a few MB, two tiny kernels) buys coalesced row loads: live ticks and minibatches measured
// 0.1-0.3 ms faster that way.
constexpr int64_t kRowMajorMinReads = 8192;

// DTW of device rows dX (nX, L) against the resident refs -> d_out (nX, nY) [+ argmin]
int dtw_dev_locked(wdx_ctx *ctx, const double *dX, int64_t nX, float *d_out, int32_t *d_argmin,
                   hipStream_t stream) {
    DtwRefs &R = ctx->refs;
    if (R.window == 0) {
        set_error("no reference set: call wdx_set_refs first");
        return WDX_ERR_NO_REFS;
    }
    if (nX == 0 || R.nY == 0) return WDX_SUCCESS;
    int rc;
    const int64_t L = R.L;
    const int64_t sb = dtw_scratch_bytes(L, R.window);
    if (sb && (rc = ctx->scratch.ensure((size_t)sb))) return rc;
    // small problems: one launch of the anti-diagonal wavefront kernel straight from the row-major
    // inputs (no transpose), then the argmin
    if (dtw_wavefront_eligible(nX, R.nY, L, R.window, ctx->knobs)) {
        {
            Timed t(ctx, WDX_K_DTW, stream);
            if ((rc = launch_dtw_wavefront(dX, nX, R.pad, R.Lpad, R.halo, R.nY, L, R.window, R.penalty,
                                           d_out, stream)))
                return rc;
        }
        if (d_argmin) return launch_argmin(d_out, nX, R.nY, d_argmin, stream);
        return WDX_SUCCESS;
    }
    // lanes = reads unless there are too few of them to fill a wave and there are more refs
    const bool lanes_are_reads = nX >= 64 || nX >= R.nY;
    if (lanes_are_reads && !sb && nX >= kRowMajorMinReads) {
        // the kernel reads the row-major fingerprints as they are (a lane owns a row): no transposed copy
        Timed t(ctx, WDX_K_DTW, stream);
        return launch_dtw(dX, 1, nX, nullptr, R.pad, R.Lpad, R.halo, R.nY, R.has_nan, L, R.window, R.penalty, d_out,
                          R.nY, 1, d_argmin, nullptr, 0, stream, ctx->knobs, true);
    }
    if (lanes_are_reads) {
        const int64_t ld = round_up(nX, 64);
        if ((rc = ctx->tmp1.ensure((size_t)L * ld * sizeof(double)))) return rc;
        if ((rc = ctx->tmp2.ensure((size_t)ld))) return rc;
        {
            Timed t(ctx, WDX_K_TRANSPOSE, stream);
            if ((rc = launch_transpose(dX, nX, L, (double *)ctx->tmp1.p, ld, (uint8_t *)ctx->tmp2.p,
                                       stream)))
                return rc;
        }
        Timed t(ctx, WDX_K_DTW, stream);
        if (sb && d_argmin) {
            if ((rc = launch_dtw((const double *)ctx->tmp1.p, ld, nX, (const uint8_t *)ctx->tmp2.p,
                                 R.pad, R.Lpad, R.halo, R.nY, R.has_nan, L, R.window, R.penalty,
                                 d_out, R.nY, 1, nullptr, ctx->scratch.p, (int64_t)ctx->scratch.bytes,
                                 stream, ctx->knobs)))
                return rc;
            return launch_argmin(d_out, nX, R.nY, d_argmin, stream);
        }
        return launch_dtw((const double *)ctx->tmp1.p, ld, nX, (const uint8_t *)ctx->tmp2.p, R.pad,
                          R.Lpad, R.halo, R.nY, R.has_nan, L, R.window, R.penalty, d_out, R.nY, 1,
                          d_argmin, ctx->scratch.p, (int64_t)ctx->scratch.bytes, stream, ctx->knobs);
    }
    // few reads, many refs (live / per-read calls): lanes = refs, the read is the uniform operand
    const int halo = kMaxRegWindow - 1;
    const int64_t Lpad = L + 2 * halo;
    if ((rc = ctx->tmp1.ensure((size_t)nX * Lpad * sizeof(double)))) return rc;
    if ((rc = ctx->tmp2.ensure((size_t)round_up(nX, 64)))) return rc;
    {
        Timed t(ctx, WDX_K_TRANSPOSE, stream);
        if ((rc = launch_pad_rows(dX, nX, L, (double *)ctx->tmp1.p, Lpad, halo,
                                  (uint8_t *)ctx->tmp2.p, stream)))
            return rc;
    }
    {
        Timed t(ctx, WDX_K_DTW, stream);
        if ((rc = launch_dtw(R.T, R.ldT, R.nY, R.has_nan, (const double *)ctx->tmp1.p, Lpad, halo,
                             nX, (const uint8_t *)ctx->tmp2.p, L, R.window, R.penalty, d_out, 1,
                             R.nY, nullptr, ctx->scratch.p, (int64_t)ctx->scratch.bytes, stream, ctx->knobs)))
            return rc;
    }
    if (d_argmin) return launch_argmin(d_out, nX, R.nY, d_argmin, stream);
    return WDX_SUCCESS;
}

}  // namespace wdx

extern "C" {

int wdx_abi_version(void) { return WDX_ABI_VERSION; }

const char *wdx_last_error(void) { return g_err; }

int wdx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return WDX_ERR_NO_DEVICE;
    }
    return n;
}

int wdx_ctx_create(int device, wdx_ctx **out) {
    if (!out) {
        set_error("null out pointer");
        return WDX_ERR_INVALID;
    }
    *out = nullptr;
    int n = wdx_device_count();
    if (n < 0) return n;
    if (n == 0) {
        set_error("no HIP device visible");
        return WDX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (0..%d)", device, n - 1);
        return WDX_ERR_INVALID;
    }
    DeviceGuard guard(device);
    if (guard.rc) return guard.rc;
    hipDeviceProp_t prop;
    WDX_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 only", device,
                  prop.gcnArchName);
        return WDX_ERR_NO_DEVICE;
    }
    wdx_ctx *c = new wdx_ctx();
    c->device = device;
    // the context's own stream: host-buffer calls of different contexts (one per thread)
    // overlap instead of queueing on the NULL stream
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
        delete c;
        return WDX_ERR_HIP;
    }
    *out = c;
    return WDX_SUCCESS;
}

void wdx_ctx_destroy(wdx_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    (void)hipDeviceSynchronize();
    for (wdx_ctx *&S : ctx->slots) {
        if (S) {
            S->refs = wdx::DtwRefs{};  // (borrowed pointers)
            wdx_ctx_destroy(S);
        }
        S = nullptr;
    }
    comm_destroy(ctx);
    for (Buffer *b : {&ctx->refs_pad, &ctx->refs_T, &ctx->refs_nan, &ctx->in0, &ctx->in1, &ctx->in2,
                      &ctx->in3, &ctx->out0, &ctx->out1, &ctx->out2, &ctx->out3, &ctx->tmp0,
                      &ctx->tmp1, &ctx->tmp2, &ctx->scratch, &ctx->fp_ws, &ctx->svm_buf, &ctx->ref_buf, &ctx->fp_big, &ctx->ref_ws, &ctx->pk_idx, &ctx->svm_fused, &ctx->svm_refs,
                      &ctx->mb_dwell, &ctx->mb_stats, &ctx->mb_prob, &ctx->mb_pred, &ctx->mb_conf})
        b->release();
    ctx->pin_in.release();
    ctx->pin_out.release();
    ctx->pk_host.release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (int k = 0; k < kNumTimed; ++k)
        for (auto &e : ctx->pending[k]) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    for (auto &e : ctx->pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete ctx;
}

int wdx_ctx_set_option(wdx_ctx *ctx, int32_t option, int64_t value) {
    WDX_ENTER(ctx);
    std::lock_guard<std::mutex> g(ctx->mu);
    switch (option) {
        case WDX_OPT_EXACT_PATH: ctx->knobs.exact_path = value != 0; break;
        case WDX_OPT_NO_WAVEFRONT_DTW: ctx->knobs.no_wavefront = value != 0; break;
        case WDX_OPT_NO_SHORT_DTW: ctx->knobs.no_short_dtw = value != 0; break;
        case WDX_OPT_SVM_SCALAR: ctx->knobs.svm_scalar = value != 0; break;
        case WDX_OPT_DEBUG_OCCUPANCY: ctx->knobs.debug_occ = value != 0; break;
        case WDX_OPT_FAST_PEAK_CAP: ctx->knobs.fast_peak_cap = (int)value; break;
        case WDX_OPT_FAST_EXACT_SCORES: ctx->knobs.fast_exact_scores = value != 0; break;
        case WDX_OPT_FAST_MAIN_CAP: ctx->knobs.fast_main_cap = (int)value; break;
        case WDX_OPT_FAST_CHAIN_MIN_READS: ctx->knobs.fast_chain_min = (int)value; break;
        case WDX_OPT_EXACT_NO_PEAK_LIST: ctx->knobs.exact_no_list = value != 0; break;
        case WDX_OPT_NO_PEAK_FILTER: ctx->knobs.no_peak_filter = value != 0; break;
        case WDX_OPT_NO_WAVE_CLIP_LONG: ctx->knobs.no_wave_clip_long = value != 0; break;
        case WDX_OPT_NO_CLIP_REUSE: ctx->knobs.no_clip_reuse = value != 0; break;
        case WDX_OPT_NO_SPLIT_TAIL: ctx->knobs.no_split = value != 0; break;
        case WDX_OPT_DTW_UNFUSED: ctx->knobs.dtw_unfused = (value >= 0 && value <= 3) ? (int)value : 0; break;
        case WDX_OPT_MAX_LAUNCH_SLICE: ctx->knobs.max_launch_slice = value > 0 ? value : 0; break;
        default:
            set_error("unknown option %d", (int)option);
            return WDX_ERR_INVALID;
    }
    return WDX_SUCCESS;
}

int wdx_ctx_stream(wdx_ctx *ctx, void **stream) {
    WDX_ENTER(ctx);
    if (!stream) {
        set_error("ctx_stream: null output");
        return WDX_ERR_INVALID;
    }
    *stream = (void *)ctx->stream;
    return WDX_SUCCESS;
}

int wdx_ctx_synchronize(wdx_ctx *ctx, void *stream) {
    WDX_ENTER(ctx);
    // NULL names the legacy NULL stream, as in every *_dev entry point (engine.py hands over torch's default
    // stream, whose handle is 0).  The context's own non-blocking stream -- the one the host-buffer calls run
    // on -- is waited for as well: those calls synchronise before they return, so this costs nothing and a
    // caller who only knows the context cannot be handed a half-finished result.
    if (stream) {
        WDX_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    } else {
        WDX_HIP_TRY(hipStreamSynchronize(nullptr));
        WDX_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return WDX_SUCCESS;
}

int wdx_refs_generation(wdx_ctx *ctx, int64_t *generation) {
    WDX_ENTER(ctx);
    if (!generation) {
        set_error("refs_generation: null output");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    *generation = ctx->refs_gen;
    return WDX_SUCCESS;
}

int wdx_set_refs(wdx_ctx *ctx, const double *Y, int64_t nY, int64_t L, int32_t window,
                 double penalty) {
    WDX_ENTER(ctx);
    std::lock_guard<std::mutex> g(ctx->mu);
    if ((rc = use_stream(ctx, ctx->stream))) return rc;
    for (wdx_ctx *S : ctx->slots)   // (also when only window / penalty change: a slot copied them at its submit)
        if (S) WDX_HIP_TRY(hipStreamSynchronize(S->stream));
    return set_refs_locked(ctx, Y, nY, L, window, penalty, ctx->stream);
}

int wdx_dtw_matrix_dev(wdx_ctx *ctx, const double *dX, int64_t nX, float *d_out, int32_t *d_argmin,
                       void *stream) {
    WDX_ENTER(ctx);
    if (nX < 0 || (nX > 0 && (!dX || !d_out))) {
        set_error("dtw_matrix_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    if ((rc = use_stream(ctx, (hipStream_t)stream))) return rc;
    return dtw_dev_locked(ctx, dX, nX, d_out, d_argmin, (hipStream_t)stream);
}

int wdx_dtw_matrix(wdx_ctx *ctx, const double *X, int64_t nX, const double *Y, int64_t nY,
                   int64_t L, int32_t window, double penalty, float *out, int32_t *argmin) {
    WDX_ENTER(ctx);
    if (nX < 0 || nY < 0 || L <= 0 || (nX > 0 && !X) || (nY > 0 && !Y) ||
        (nX > 0 && nY > 0 && !out)) {
        set_error("dtw_matrix: need nX,nY >= 0, L > 0 and non-null buffers");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    hipStream_t s = ctx->stream;
    if ((rc = use_stream(ctx, s))) return rc;
    if ((rc = set_refs_locked(ctx, Y, nY, L, window, penalty, s))) return rc;
    if (nX == 0 || nY == 0) {
        if (argmin)
            for (int64_t i = 0; i < nX; ++i) argmin[i] = 0;
        return WDX_SUCCESS;
    }
    const size_t xb = (size_t)(nX * L) * sizeof(double), ob = (size_t)(nX * nY) * sizeof(float);
    if ((rc = ctx->in0.ensure(xb))) return rc;
    if ((rc = ctx->out0.ensure(ob))) return rc;
    if (argmin && (rc = ctx->out1.ensure((size_t)nX * sizeof(int32_t)))) return rc;
    StreamDrain drain(s);
    WDX_HIP_TRY(hipMemcpyAsync(ctx->in0.p, X, xb, hipMemcpyHostToDevice, s));
    if ((rc = dtw_dev_locked(ctx, (const double *)ctx->in0.p, nX, (float *)ctx->out0.p,
                             argmin ? (int32_t *)ctx->out1.p : nullptr, s)))
        return rc;
    WDX_HIP_TRY(hipMemcpyAsync(out, ctx->out0.p, ob, hipMemcpyDeviceToHost, s));
    if (argmin)
        WDX_HIP_TRY(hipMemcpyAsync(argmin, ctx->out1.p, (size_t)nX * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, s));
    WDX_HIP_TRY(hipStreamSynchronize(s));
    drain.done();
    return WDX_SUCCESS;
}

int wdx_fingerprint_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off,
                        const int32_t *d_row_len, int64_t stride, int64_t max_len, int64_t n_reads,
                        const int32_t *d_a_start, const int32_t *d_a_end, const uint8_t *d_ok,
                        const wdx_seg_params *p, double *d_fpt, int64_t *d_dwell, double *d_stats,
                        int32_t *d_status, void *stream) {
    WDX_ENTER(ctx);
    if (n_reads < 0 || !p || (n_reads > 0 && (!d_sig || !d_a_start || !d_a_end || !d_status))) {
        set_error("fingerprint_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    if ((rc = use_stream(ctx, (hipStream_t)stream))) return rc;
    if ((rc = ctx->fp_ws.ensure((size_t)fingerprint_workspace_bytes(n_reads)))) return rc;
    if ((rc = ctx->fp_big.ensure((size_t)fingerprint_big_bytes(max_len)))) return rc;
    Timed t(ctx, WDX_K_FINGERPRINT, (hipStream_t)stream);
    return launch_fingerprint(d_sig, d_row_off, d_row_len, stride, max_len, n_reads, d_a_start,
                              d_a_end, d_ok, *p, d_fpt, d_dwell, d_stats, d_status,
                              (hipStream_t)stream, ctx->fp_ws.p, ctx->knobs, &t.n_launches, nullptr, 0, 0, nullptr,
                              &t.main, (double *)ctx->fp_big.p);
}

int wdx_fingerprint_refine_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off,
                               const int32_t *d_row_len, int64_t stride, int64_t max_len, int64_t n_reads,
                               const int32_t *d_a_start, const int32_t *d_a_end, const uint8_t *d_ok,
                               const wdx_seg_params *p_in, const wdx_refine_params *rp, double *d_fpt,
                               int64_t *d_dwell, double *d_stats, int32_t *d_refine_idx, int32_t *d_status,
                               void *stream) {
    WDX_ENTER(ctx);
    if (n_reads < 0 || !p_in || !rp || !rp->query ||
        (n_reads > 0 && (!d_sig || !d_a_start || !d_a_end || !d_status || !d_refine_idx))) {
        set_error("fingerprint_refine_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    if (rp->n_query < 1) {
        set_error("consensus refinement: empty query");
        return WDX_ERR_INVALID;
    }
    if (n_reads == 0) return WDX_SUCCESS;
    wdx_seg_params pv = *p_in;
    pv.barcode_num_events = rp->barcode_keep_events;  // K of the outputs
    std::lock_guard<std::mutex> g(ctx->mu);
    hipStream_t s = (hipStream_t)stream;
    if ((rc = use_stream(ctx, s))) return rc;
    if ((rc = ctx->fp_ws.ensure((size_t)fingerprint_workspace_bytes(n_reads)))) return rc;
    if ((rc = ctx->fp_big.ensure((size_t)fingerprint_big_bytes(max_len)))) return rc;
    const size_t qb = ((size_t)rp->n_query * 8 + 15) / 16 * 16;
    {
        const void *before = ctx->ref_buf.p;
        if ((rc = ctx->ref_buf.ensure(qb))) return rc;
        if (ctx->ref_buf.p != before) ctx->ref_query_host.clear();  // (a new block: nothing is resident in it)
    }
    const size_t wb = (size_t)fingerprint_refine_ws_bytes(n_reads);
    if ((rc = ctx->ref_ws.ensure(wb ? wb : 8))) return rc;
    // the consensus comes from the host (84 doubles) and is kept resident: uploaded only when its content changes, and
    // then synchronously after the stream has drained (no reliance on how the runtime stages pageable copies)
    if (ctx->ref_query_host.size() != (size_t)rp->n_query ||
        memcmp(ctx->ref_query_host.data(), rp->query, (size_t)rp->n_query * 8) != 0) {
        WDX_HIP_TRY(hipStreamSynchronize(s));
        WDX_HIP_TRY(hipMemcpy(ctx->ref_buf.p, rp->query, (size_t)rp->n_query * 8, hipMemcpyHostToDevice));
        ctx->ref_query_host.assign(rp->query, rp->query + rp->n_query);
    }
    WDX_HIP_TRY(hipMemsetAsync(d_refine_idx, 0xff, (size_t)n_reads * 12, s));
    // (only the state word of every hand-over record: 4 of its 1632 bytes)
    WDX_HIP_TRY(hipMemset2DAsync(ctx->ref_ws.p, (size_t)fingerprint_refine_ws_bytes(1), 0, 4, (size_t)n_reads, s));
    RefineDev *rf = nullptr;
    struct RfGuard {
        RefineDev *&r;
        ~RfGuard() { free_refine_dev(r); }
    } rf_guard{rf};
    if ((rc = fill_refine_dev(*rp, (const double *)ctx->ref_buf.p, d_refine_idx, &rf))) return rc;
    set_refine_ws(rf, ctx->ref_ws.p);
    Timed t(ctx, WDX_K_FINGERPRINT, s);
    return launch_fingerprint(d_sig, d_row_off, d_row_len, stride, max_len, n_reads, d_a_start, d_a_end, d_ok, pv, d_fpt,
                              d_dwell, d_stats, d_status, s, ctx->fp_ws.p, ctx->knobs, &t.n_launches, nullptr, 0, 0, rf,
                              nullptr, (double *)ctx->fp_big.p);
}

int wdx_fingerprint_profile_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off,
                                int64_t stride, int64_t max_len, int64_t n_reads,
                                const int32_t *d_a_start, const int32_t *d_a_end,
                                const wdx_seg_params *p, int32_t *d_status, long long *d_prof,
                                int64_t prof_reads, int32_t fast_path, int32_t stop_phase, void *stream) {
    WDX_ENTER(ctx);
    if (!p || !d_prof || !d_status) {
        set_error("fingerprint_profile_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    if ((rc = use_stream(ctx, (hipStream_t)stream))) return rc;
    if ((rc = ctx->fp_ws.ensure((size_t)fingerprint_workspace_bytes(n_reads)))) return rc;
    rc = launch_fingerprint(d_sig, d_row_off, nullptr, stride, max_len, n_reads, d_a_start, d_a_end,
                            nullptr, *p, nullptr, nullptr, nullptr, d_status, (hipStream_t)stream,
                            fast_path ? ctx->fp_ws.p : nullptr, ctx->knobs, nullptr, d_prof, prof_reads,
                            fast_path == 2 ? -2 : stop_phase);   // (-2: the split pair's diagnostic build)
    if (rc == WDX_SUCCESS && fast_path && prof_reads > 0 && stop_phase == 0) {
        // slot 15 of read 0 <- number of reads the fast kernel handed to the slow path
        WDX_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        unsigned cnt = 0;
        WDX_HIP_TRY(hipMemcpy(&cnt, ctx->fp_ws.p, 4, hipMemcpyDeviceToHost));
        long long v = cnt;
        WDX_HIP_TRY(hipMemcpy(d_prof + 15, &v, 8, hipMemcpyHostToDevice));  // row 0, slot 15 of 32
    }
    return rc;
}

static int fingerprint_batch_impl(wdx_ctx *ctx, const float *sig, int64_t n_reads, int64_t stride,
                                  const int32_t *a_start, const int32_t *a_end, const uint8_t *ok,
                                  const wdx_seg_params *p_in, const wdx_refine_params *rp, double *fpt, int64_t *dwell,
                                  double *stats, int32_t *refine_idx, int32_t *status) {
    WDX_ENTER(ctx);
    if (!p_in || (rp && (!rp->query || !refine_idx))) {
        set_error("fingerprint_batch: bad arguments");
        return WDX_ERR_INVALID;
    }
    wdx_seg_params pv = *p_in;
    if (rp) pv.barcode_num_events = rp->barcode_keep_events;  // K of the outputs
    const wdx_seg_params *p = &pv;
    if (n_reads < 0 || stride < 0 || !p ||
        (n_reads > 0 && (!sig || !a_start || !a_end || !fpt || !dwell || !stats || !status))) {
        set_error("fingerprint_batch: bad arguments");
        return WDX_ERR_INVALID;
    }
    if (n_reads == 0) return WDX_SUCCESS;
    std::lock_guard<std::mutex> g(ctx->mu);
    hipStream_t s = ctx->stream;
    if ((rc = use_stream(ctx, s))) return rc;
    const int64_t K = p->barcode_num_events;
    if (K < 1) {
        set_error("barcode_num_events must be >= 1");
        return WDX_ERR_INVALID;
    }
    // exact bound on the adapter window over this batch (extract_adapter, sig_proc.py:388-389)
    int64_t max_len = 0, col0 = stride, col1 = 0;  // columns [col0, col1) hold every adapter window of the batch
    for (int64_t r = 0; r < n_reads; ++r) {
        if (ok && !ok[r]) continue;
        int64_t st = (int64_t)a_start[r] - p->padding, en = (int64_t)a_end[r] + p->padding;
        if (st < 0) st = 0;
        if (en > stride) en = stride;
        if (en - st > max_len) max_len = en - st;
        if (en > st) {
            if (st < col0) col0 = st;
            if (en > col1) col1 = en;
        }
    }
    const size_t sb = (size_t)(n_reads * stride) * sizeof(float);
    if ((rc = ctx->in0.ensure(sb ? sb : 4))) return rc;
    if ((rc = ctx->in1.ensure((size_t)n_reads * 4))) return rc;
    if ((rc = ctx->in2.ensure((size_t)n_reads * 4))) return rc;
    if ((rc = ctx->in3.ensure((size_t)n_reads))) return rc;
    if ((rc = ctx->out0.ensure((size_t)(n_reads * K) * 8))) return rc;
    if ((rc = ctx->out1.ensure((size_t)(n_reads * K) * 8))) return rc;
    if ((rc = ctx->out2.ensure((size_t)n_reads * 6 * 8))) return rc;
    if ((rc = ctx->out3.ensure((size_t)n_reads * 4))) return rc;
    if ((rc = ctx->fp_ws.ensure((size_t)fingerprint_workspace_bytes(n_reads)))) return rc;
    if ((rc = ctx->fp_big.ensure((size_t)fingerprint_big_bytes(max_len)))) return rc;
    RefineDev *rf = nullptr;
    struct RfGuard {
        RefineDev *&r;
        ~RfGuard() { free_refine_dev(r); }
    } rf_guard{rf};
    StreamDrain drain(s);
    if (rp) {
        // [query doubles | idx int32 (n,3)] on the device; idx starts as -1 (reads that fail before the match)
        const size_t qb = ((size_t)rp->n_query * 8 + 15) / 16 * 16;
        if (rp->n_query < 1) {
            set_error("consensus refinement: empty query");
            return WDX_ERR_INVALID;
        }
        if ((rc = ctx->ref_buf.ensure(qb + (size_t)n_reads * 12))) return rc;
        ctx->ref_query_host.clear();  // (this call's query replaces whatever wdx_fingerprint_refine_dev kept resident)
        WDX_HIP_TRY(hipMemcpyAsync(ctx->ref_buf.p, rp->query, (size_t)rp->n_query * 8, hipMemcpyHostToDevice, s));
        WDX_HIP_TRY(hipMemsetAsync((unsigned char *)ctx->ref_buf.p + qb, 0xff, (size_t)n_reads * 12, s));
        if ((rc = fill_refine_dev(*rp, (const double *)ctx->ref_buf.p, (int32_t *)((unsigned char *)ctx->ref_buf.p + qb),
                                  &rf)))
            return rc;
        // hand-over records of the fast kernels (state 0 = untouched)
        const size_t wb = (size_t)fingerprint_refine_ws_bytes(n_reads);
        if ((rc = ctx->ref_ws.ensure(wb ? wb : 8))) return rc;
        WDX_HIP_TRY(hipMemsetAsync(ctx->ref_ws.p, 0, wb, s));
        set_refine_ws(rf, ctx->ref_ws.p);
    }
    // only the columns that hold adapter windows travel (the rows are NaN-padded to sig_preload_size,
    // file_proc.py:244-260; the kernels never read outside [start, stop))
    if (col1 > col0)
        WDX_HIP_TRY(hipMemcpy2DAsync((float *)ctx->in0.p + col0, (size_t)stride * sizeof(float), sig + col0,
                                     (size_t)stride * sizeof(float), (size_t)(col1 - col0) * sizeof(float),
                                     (size_t)n_reads, hipMemcpyHostToDevice, s));
    WDX_HIP_TRY(hipMemcpyAsync(ctx->in1.p, a_start, (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
    WDX_HIP_TRY(hipMemcpyAsync(ctx->in2.p, a_end, (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
    if (ok) WDX_HIP_TRY(hipMemcpyAsync(ctx->in3.p, ok, (size_t)n_reads, hipMemcpyHostToDevice, s));
    {
        Timed t(ctx, WDX_K_FINGERPRINT, s);
        if ((rc = launch_fingerprint((const float *)ctx->in0.p, nullptr, nullptr, stride, max_len,
                                     n_reads, (const int32_t *)ctx->in1.p,
                                     (const int32_t *)ctx->in2.p,
                                     ok ? (const uint8_t *)ctx->in3.p : nullptr, *p,
                                     (double *)ctx->out0.p, (int64_t *)ctx->out1.p,
                                     (double *)ctx->out2.p, (int32_t *)ctx->out3.p, s, ctx->fp_ws.p,
                                     ctx->knobs, &t.n_launches, nullptr, 0, 0, rf, nullptr, (double *)ctx->fp_big.p)))
            return rc;
    }
    WDX_HIP_TRY(hipMemcpyAsync(fpt, ctx->out0.p, (size_t)(n_reads * K) * 8, hipMemcpyDeviceToHost, s));
    WDX_HIP_TRY(hipMemcpyAsync(dwell, ctx->out1.p, (size_t)(n_reads * K) * 8, hipMemcpyDeviceToHost, s));
    WDX_HIP_TRY(hipMemcpyAsync(stats, ctx->out2.p, (size_t)n_reads * 48, hipMemcpyDeviceToHost, s));
    WDX_HIP_TRY(hipMemcpyAsync(status, ctx->out3.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s));
    if (rp) {
        const size_t qb = ((size_t)rp->n_query * 8 + 15) / 16 * 16;
        WDX_HIP_TRY(hipMemcpyAsync(refine_idx, (unsigned char *)ctx->ref_buf.p + qb, (size_t)n_reads * 12,
                                   hipMemcpyDeviceToHost, s));
    }
    WDX_HIP_TRY(hipStreamSynchronize(s));
    drain.done();
    return WDX_SUCCESS;
}

int wdx_fingerprint_batch(wdx_ctx *ctx, const float *sig, int64_t n_reads, int64_t stride,
                          const int32_t *a_start, const int32_t *a_end, const uint8_t *ok,
                          const wdx_seg_params *p, double *fpt, int64_t *dwell, double *stats,
                          int32_t *status) {
    return fingerprint_batch_impl(ctx, sig, n_reads, stride, a_start, a_end, ok, p, nullptr, fpt, dwell, stats, nullptr,
                                  status);
}

int wdx_fingerprint_refine_batch(wdx_ctx *ctx, const float *sig, int64_t n_reads, int64_t stride,
                                 const int32_t *a_start, const int32_t *a_end, const uint8_t *ok,
                                 const wdx_seg_params *p, const wdx_refine_params *rp, double *fpt, int64_t *dwell,
                                 double *stats, int32_t *refine_idx, int32_t *status) {
    if (!rp) {
        set_error("fingerprint_refine_batch: null refinement parameters");
        return WDX_ERR_INVALID;
    }
    return fingerprint_batch_impl(ctx, sig, n_reads, stride, a_start, a_end, ok, p, rp, fpt, dwell, stats, refine_idx,
                                  status);
}

int64_t wdx_demux_workspace_bytes(int64_t n_reads, int32_t K) {
    if (n_reads < 0 || K < 1) return 0;
    // [fpt (n,K) f64][small batches only: fptT (K,ld) f64 + nan flags ld][slow-path lists]
    const int64_t ld = round_up(n_reads > 0 ? n_reads : 1, 64);
    const int64_t small = n_reads < kRowMajorMinReads ? (int64_t)K * ld * 8 + ld : 0;
    return n_reads * K * 8 + small + 512 + fingerprint_workspace_bytes(n_reads);
}

int wdx_demux_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off,
                  const int32_t *d_row_len, int64_t stride, int64_t max_len, int64_t n_reads,
                  const int32_t *d_a_start, const int32_t *d_a_end, const uint8_t *d_ok,
                  const wdx_seg_params *p, double *d_fpt, int64_t *d_dwell, double *d_stats,
                  int32_t *d_status, float *d_dist, int32_t *d_call, int64_t *d_counts, void *d_work,
                  void *stream) {
    WDX_ENTER(ctx);
    if (n_reads < 0 || !p ||
        (n_reads > 0 && (!d_sig || !d_a_start || !d_a_end || !d_status || !d_dist || !d_call || !d_work))) {
        set_error("demux_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    DtwRefs &R = ctx->refs;
    if (R.window == 0) {
        set_error("no reference set: call wdx_set_refs first");
        return WDX_ERR_NO_REFS;
    }
    const int64_t K = p->barcode_num_events;
    if (K != R.L) {
        set_error("barcode_num_events (%lld) != reference length (%lld)", (long long)K,
                  (long long)R.L);
        return WDX_ERR_INVALID;
    }
    if (dtw_scratch_bytes(R.L, R.window)) {
        set_error("demux_dev needs window <= %d", kMaxRegWindow);
        return WDX_ERR_UNSUPPORTED;
    }
    if (n_reads == 0) return WDX_SUCCESS;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = use_stream(ctx, s))) return rc;
    if ((rc = ctx->fp_big.ensure((size_t)fingerprint_big_bytes(max_len)))) return rc;
    unsigned char *w = (unsigned char *)d_work;
    double *fpt = d_fpt ? d_fpt : (double *)w;
    const bool rowmajor = n_reads >= kRowMajorMinReads;
    const int64_t ld = round_up(n_reads, 64);
    double *fptT = (double *)(w + ((n_reads * K * 8 + 255) / 256) * 256);
    uint8_t *flags = (uint8_t *)(fptT + (rowmajor ? 0 : K * ld));
    void *fp_ws = (unsigned char *)flags + (rowmajor ? 0 : ((ld + 255) / 256) * 256);
    {
        Timed t(ctx, WDX_K_FINGERPRINT, s);
        if ((rc = launch_fingerprint(d_sig, d_row_off, d_row_len, stride, max_len, n_reads, d_a_start,
                                     d_a_end, d_ok, *p, fpt, d_dwell, d_stats, d_status, s, fp_ws,
                                     ctx->knobs, &t.n_launches, nullptr, 0, 0, nullptr, &t.main, (double *)ctx->fp_big.p)))
            return rc;
    }
    if (rowmajor) {
        // failed reads carry NaN fingerprints; the DTW kernel reads the row-major rows in place and flags them
        Timed t(ctx, WDX_K_DTW, s);
        if ((rc = launch_dtw(fpt, 1, n_reads, nullptr, R.pad, R.Lpad, R.halo, R.nY, R.has_nan, R.L,
                             R.window, R.penalty, d_dist, R.nY, 1, d_call, nullptr, 0, s, ctx->knobs, true)))
            return rc;
    } else {
        {
            Timed t(ctx, WDX_K_TRANSPOSE, s);
            if ((rc = launch_transpose(fpt, n_reads, K, fptT, ld, flags, s))) return rc;
        }
        Timed t(ctx, WDX_K_DTW, s);
        if ((rc = launch_dtw(fptT, ld, n_reads, flags, R.pad, R.Lpad, R.halo, R.nY, R.has_nan, R.L,
                             R.window, R.penalty, d_dist, R.nY, 1, d_call, nullptr, 0, s, ctx->knobs)))
            return rc;
    }
    Timed t(ctx, WDX_K_COUNT, s);
    return launch_count_calls(d_call, d_status, n_reads, R.nY, d_counts, s);
}

// Body of the fused host-buffer call.  `B` owns the stream and every workspace that is touched (the context itself
// for wdx_demux_batch; one of its pipeline slots for wdx_demux_submit[_ex]), `R` is the resident reference set
// (read-only device memory of the parent context), `svm` the parent's resident model when the SVM tail is asked for.
// Everything is ENQUEUED on B->stream -- copies in, the kernel chain, copies out to the host destinations (caller arrays
// or the slot's page-locked block); the caller synchronises.  Only the columns that hold adapter windows travel (the
// rows are NaN-padded to sig_preload_size, file_proc.py:244-260; the kernels never read outside [start, stop)).
struct MbHostOut {  // host destinations of one minibatch; null = not wanted (status always)
    int32_t *status = nullptr, *call = nullptr;
    float *dist = nullptr;
    double *fpt = nullptr;
    int64_t *dwell = nullptr;
    double *stats = nullptr, *prob = nullptr;
    int32_t *pred = nullptr;
    double *conf = nullptr;
};

static int demux_batch_enqueue(wdx_ctx *B, const DtwRefs &R, const wdx_minibatch_in &in, const wdx_seg_params *p,
                               const MbHostOut &H, const SvmDev *svm) {
    int rc = WDX_SUCCESS;
    hipStream_t s = B->stream;
    const float *sig = in.sig;
    const int64_t n_reads = in.n_reads, stride = in.stride;
    const int32_t *a_start = in.a_start, *a_end = in.a_end;
    const uint8_t *ok = in.ok;
    const int64_t K = p->barcode_num_events;
    const bool packed_in = in.row_off != nullptr;
    int64_t max_len = 0, col0 = stride, col1 = 0;  // columns [col0, col1) hold every adapter window of the batch
    int64_t win_total = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        if (ok && !ok[r]) continue;
        const int64_t rl = packed_in ? (int64_t)in.row_len[r] : stride;
        int64_t st = (int64_t)a_start[r] - p->padding, en = (int64_t)a_end[r] + p->padding;
        if (st < 0) st = 0;
        if (en > rl) en = rl;
        if (en - st > max_len) max_len = en - st;
        if (en > st) {
            win_total += en - st;
            if (st < col0) col0 = st;
            if (en > col1) col1 = en;
        }
    }
    const size_t sb = (size_t)(packed_in ? in.row_off[n_reads] : n_reads * stride) * sizeof(float);
    const size_t db = (size_t)(n_reads * (R.nY > 0 ? R.nY : 1)) * sizeof(float);
    if ((rc = B->in0.ensure(sb ? sb : 4))) return rc;
    if ((rc = B->in1.ensure((size_t)n_reads * 4))) return rc;
    if ((rc = B->in2.ensure((size_t)n_reads * 4))) return rc;
    if ((rc = B->in3.ensure((size_t)n_reads))) return rc;
    if ((rc = B->out0.ensure((size_t)(n_reads * K) * 8))) return rc;
    if ((rc = B->out1.ensure(db))) return rc;
    if ((rc = B->out2.ensure((size_t)n_reads * 4))) return rc;
    if ((rc = B->out3.ensure((size_t)n_reads * 4))) return rc;
    if (H.dwell && (rc = B->mb_dwell.ensure((size_t)(n_reads * K) * 8))) return rc;
    if (H.stats && (rc = B->mb_stats.ensure((size_t)n_reads * 48))) return rc;
    if (svm) {
        if ((rc = B->mb_prob.ensure((size_t)n_reads * svm->k * 8))) return rc;
        if ((rc = B->mb_pred.ensure((size_t)n_reads * 4))) return rc;
        if ((rc = B->mb_conf.ensure((size_t)n_reads * 8))) return rc;
    }
    if ((rc = B->fp_ws.ensure((size_t)fingerprint_workspace_bytes(n_reads)))) return rc;
    if ((rc = B->fp_big.ensure((size_t)fingerprint_big_bytes(max_len)))) return rc;
    int64_t *d_dwell = H.dwell ? (int64_t *)B->mb_dwell.p : nullptr;
    double *d_stats = H.stats ? (double *)B->mb_stats.p : nullptr;
    // (Letting the FINGERPRINT kernel read a page-locked minibatch in place over the bus was measured and lost: 1.80 M
    // reads/s against 2.46 M with a DMA copy, which runs at 49 GB/s.)
    // Three ways in.  (i) The 2-D DMA copy of the column range that holds every adapter window of the batch -- all a
    // pageable array allows, and the best there is when every read's adapter starts at the same sample.  (ii) Rows that
    // carry whole reads have their adapters at different places (sig_proc.py:382-391: adapter_start varies per read) and
    // the column union is most of the row: when the minibatch is page-locked (wdx_host_alloc) and the windows are less
    // than 0.85 of the union, a copy kernel reads ONLY the windows over the bus, back to back into a packed device
    // buffer (pack_windows_kernel: pure copy, every load in flight), and the kernels run on the packed layout -- row r =
    // the original row's samples [st_r, en_r), adapter bounds shifted by st_r: the same window, bit for bit.  (iii) Rows the
    // CALLER packed (wdx_minibatch_in.row_off; the feeder's workers): one flat copy of exactly the windows.
    const float *sig_dev = nullptr;  // the minibatch as the device sees it, when it is page-locked
    if (!packed_in && col1 > col0 && (double)win_total < 0.85 * (double)((col1 - col0) * n_reads)) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, sig) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer)
            sig_dev = (const float *)at.devicePointer;
        else
            (void)hipGetLastError();
    }
    if (packed_in) {
        // device images: off int64[n+1] | len int32[n] | a_start int32[n] | a_end int32[n], straight from the caller's arrays
        const size_t ib = (size_t)(n_reads + 1) * 8 + (size_t)n_reads * 12;
        if ((rc = B->pk_idx.ensure(ib))) return rc;
        int64_t *d_off = (int64_t *)B->pk_idx.p;
        int32_t *d_len = (int32_t *)(d_off + n_reads + 1), *d_as = d_len + n_reads, *d_ae = d_as + n_reads;
        if (sb) WDX_HIP_TRY(hipMemcpyAsync(B->in0.p, sig, sb, hipMemcpyHostToDevice, s));
        WDX_HIP_TRY(hipMemcpyAsync(d_off, in.row_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s));
        WDX_HIP_TRY(hipMemcpyAsync(d_len, in.row_len, (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
        WDX_HIP_TRY(hipMemcpyAsync(d_as, a_start, (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
        WDX_HIP_TRY(hipMemcpyAsync(d_ae, a_end, (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
        if (ok) WDX_HIP_TRY(hipMemcpyAsync(B->in3.p, ok, (size_t)n_reads, hipMemcpyHostToDevice, s));
        Timed t(B, WDX_K_FINGERPRINT, s);
        if ((rc = launch_fingerprint((const float *)B->in0.p, d_off, d_len, 0, max_len, n_reads, d_as, d_ae,
                                     ok ? (const uint8_t *)B->in3.p : nullptr, *p, (double *)B->out0.p, d_dwell, d_stats,
                                     (int32_t *)B->out3.p, s, B->fp_ws.p, B->knobs, &t.n_launches, nullptr, 0, 0, nullptr,
                                     &t.main, (double *)B->fp_big.p)))
            return rc;
    } else if (sig_dev) {
        // host images (page-locked, owned by the slot until its copy has run): off int64[n+1] | st int32[n] | len
        // int32[n] | a_start' int32[n] | a_end' int32[n]
        const size_t ib = (size_t)(n_reads + 1) * 8 + (size_t)n_reads * 16;
        if ((rc = B->pk_host.ensure(ib))) return rc;
        if ((rc = B->pk_idx.ensure(ib))) return rc;
        int64_t *h_off = (int64_t *)B->pk_host.p;
        int32_t *h_st = (int32_t *)(h_off + n_reads + 1), *h_len = h_st + n_reads, *h_as = h_len + n_reads, *h_ae = h_as + n_reads;
        int64_t acc = 0;
        for (int64_t r = 0; r < n_reads; ++r) {
            int64_t st = (int64_t)a_start[r] - p->padding, en = (int64_t)a_end[r] + p->padding;
            if (st < 0) st = 0;
            if (en > stride) en = stride;
            if (en < st || (ok && !ok[r])) en = st;
            st &= ~(int64_t)3;  // (up to three samples ahead of the window come along: 16-byte aligned bus reads)
            h_off[r] = acc;
            h_st[r] = (int32_t)st;
            h_len[r] = (int32_t)(en - st);
            h_as[r] = a_start[r] - (int32_t)st;
            h_ae[r] = a_end[r] - (int32_t)st;
            acc += ((en - st) + 3) & ~(int64_t)3;  // (rows start on 16-byte boundaries)
        }
        h_off[n_reads] = acc;
        if ((rc = B->in0.ensure((size_t)(acc ? acc : 1) * sizeof(float)))) return rc;
        WDX_HIP_TRY(hipMemcpyAsync(B->pk_idx.p, B->pk_host.p, ib, hipMemcpyHostToDevice, s));
        const int64_t *d_off = (const int64_t *)B->pk_idx.p;
        const int32_t *d_st = (const int32_t *)(d_off + n_reads + 1), *d_len = d_st + n_reads, *d_as = d_len + n_reads,
                      *d_ae = d_as + n_reads;
        if ((rc = launch_pack_windows(sig_dev, stride, n_reads, d_off, d_st, d_len, (float *)B->in0.p, s))) return rc;
        if (ok) WDX_HIP_TRY(hipMemcpyAsync(B->in3.p, ok, (size_t)n_reads, hipMemcpyHostToDevice, s));
        Timed t(B, WDX_K_FINGERPRINT, s);
        if ((rc = launch_fingerprint((const float *)B->in0.p, d_off, d_len, 0, max_len, n_reads, d_as, d_ae,
                                     ok ? (const uint8_t *)B->in3.p : nullptr, *p, (double *)B->out0.p, d_dwell, d_stats,
                                     (int32_t *)B->out3.p, s, B->fp_ws.p, B->knobs, &t.n_launches, nullptr, 0, 0, nullptr,
                                     &t.main, (double *)B->fp_big.p)))
            return rc;
    } else {
        const float *d_sig = (const float *)B->in0.p;
        if (col1 > col0)
            WDX_HIP_TRY(hipMemcpy2DAsync((float *)B->in0.p + col0, (size_t)stride * sizeof(float), sig + col0,
                                         (size_t)stride * sizeof(float), (size_t)(col1 - col0) * sizeof(float),
                                         (size_t)n_reads, hipMemcpyHostToDevice, s));
        WDX_HIP_TRY(hipMemcpyAsync(B->in1.p, a_start, (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
        WDX_HIP_TRY(hipMemcpyAsync(B->in2.p, a_end, (size_t)n_reads * 4, hipMemcpyHostToDevice, s));
        if (ok) WDX_HIP_TRY(hipMemcpyAsync(B->in3.p, ok, (size_t)n_reads, hipMemcpyHostToDevice, s));
        Timed t(B, WDX_K_FINGERPRINT, s);
        if ((rc = launch_fingerprint(d_sig, nullptr, nullptr, stride, max_len, n_reads,
                                     (const int32_t *)B->in1.p, (const int32_t *)B->in2.p,
                                     ok ? (const uint8_t *)B->in3.p : nullptr, *p, (double *)B->out0.p,
                                     d_dwell, d_stats, (int32_t *)B->out3.p, s, B->fp_ws.p, B->knobs, &t.n_launches,
                                     nullptr, 0, 0, nullptr, &t.main, (double *)B->fp_big.p)))
            return rc;
    }
    if (R.nY > 0) {
        if ((rc = dtw_dev_locked(B, (const double *)B->out0.p, n_reads, (float *)B->out1.p,
                                 (int32_t *)B->out2.p, s)))
            return rc;
        if ((rc = launch_count_calls((int32_t *)B->out2.p, (const int32_t *)B->out3.p, n_reads, R.nY,
                                     nullptr, s)))
            return rc;
        if (H.call) WDX_HIP_TRY(hipMemcpyAsync(H.call, B->out2.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s));
        if (H.dist) WDX_HIP_TRY(hipMemcpyAsync(H.dist, B->out1.p, db, hipMemcpyDeviceToHost, s));
        if (svm) {
            // the classifier tail on the distance rows that are on the device anyway (models/dtw_svm.py:90-93, models/utils.py:45-61);
            // failed reads: pred -1, NaN probabilities (the reference never shows them to the model)
            {
                Timed t(B, WDX_K_SVM, s);
                if ((rc = launch_svm_predict(*svm, (const float *)B->out1.p, n_reads, (double *)B->mb_prob.p,
                                             (int32_t *)B->mb_pred.p, (double *)B->mb_conf.p, s, B->knobs)))
                    return rc;
            }
            if ((rc = launch_svm_mask_failed((const int32_t *)B->out3.p, n_reads, svm->k, (double *)B->mb_prob.p,
                                             (int32_t *)B->mb_pred.p, (double *)B->mb_conf.p, s)))
                return rc;
            if (H.prob) WDX_HIP_TRY(hipMemcpyAsync(H.prob, B->mb_prob.p, (size_t)n_reads * svm->k * 8, hipMemcpyDeviceToHost, s));
            if (H.pred) WDX_HIP_TRY(hipMemcpyAsync(H.pred, B->mb_pred.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s));
            if (H.conf) WDX_HIP_TRY(hipMemcpyAsync(H.conf, B->mb_conf.p, (size_t)n_reads * 8, hipMemcpyDeviceToHost, s));
        }
    }
    WDX_HIP_TRY(hipMemcpyAsync(H.status, B->out3.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, s));
    if (H.fpt) WDX_HIP_TRY(hipMemcpyAsync(H.fpt, B->out0.p, (size_t)(n_reads * K) * 8, hipMemcpyDeviceToHost, s));
    if (H.dwell) WDX_HIP_TRY(hipMemcpyAsync(H.dwell, d_dwell, (size_t)(n_reads * K) * 8, hipMemcpyDeviceToHost, s));
    if (H.stats) WDX_HIP_TRY(hipMemcpyAsync(H.stats, d_stats, (size_t)n_reads * 48, hipMemcpyDeviceToHost, s));
    return WDX_SUCCESS;
}

static int demux_check_args(wdx_ctx *ctx, const char *who, int64_t n_reads, int64_t stride, const float *sig,
                            const int32_t *a_start, const int32_t *a_end, const wdx_seg_params *p, int64_t n_refs) {
    if (n_reads < 0 || stride < 0 || !p || (n_reads > 0 && (!sig || !a_start || !a_end))) {
        set_error("%s: bad arguments", who);
        return WDX_ERR_INVALID;
    }
    DtwRefs &R = ctx->refs;
    if (R.window == 0) {
        set_error("no reference set: call wdx_set_refs first");
        return WDX_ERR_NO_REFS;
    }
    if (p->barcode_num_events != R.L) {
        set_error("barcode_num_events (%lld) != reference length (%lld)", (long long)p->barcode_num_events,
                  (long long)R.L);
        return WDX_ERR_INVALID;
    }
    if (n_refs != R.nY) {
        set_error("%s: the caller sized `dist` for %lld references but %lld are resident", who, (long long)n_refs,
                  (long long)R.nY);
        return WDX_ERR_INVALID;
    }
    return WDX_SUCCESS;
}

int wdx_demux_batch(wdx_ctx *ctx, const float *sig, int64_t n_reads, int64_t stride,
                    const int32_t *a_start, const int32_t *a_end, const uint8_t *ok,
                    const wdx_seg_params *p, int64_t n_refs, double *fpt, float *dist, int32_t *call,
                    int32_t *status) {
    WDX_ENTER(ctx);
    if (n_reads > 0 && (!call || !status)) {
        set_error("demux_batch: bad arguments");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    if ((rc = demux_check_args(ctx, "demux_batch", n_reads, stride, sig, a_start, a_end, p, n_refs))) return rc;
    if (n_reads == 0) return WDX_SUCCESS;
    hipStream_t s = ctx->stream;
    if ((rc = use_stream(ctx, s))) return rc;
    StreamDrain drain(s);
    const wdx_minibatch_in in{sig, n_reads, stride, nullptr, nullptr, a_start, a_end, ok};
    MbHostOut H;
    H.status = status;
    H.call = call;
    H.dist = dist;
    H.fpt = fpt;
    if ((rc = demux_batch_enqueue(ctx, ctx->refs, in, p, H, nullptr))) return rc;
    WDX_HIP_TRY(hipStreamSynchronize(s));
    drain.done();
    if (ctx->refs.nY == 0)
        for (int64_t r = 0; r < n_reads; ++r) call[r] = -1;
    return WDX_SUCCESS;
}

// ---- pipelined minibatches: two slots per context, submit / wait -------------------------------------------------
// The reference's workers (file_proc.py:380-454, 1197-1243) alternate "fill the next minibatch" with "process this
// one".  A slot is a child context (its own non-blocking stream and workspaces) that shares the parent's resident
// reference set: minibatch k+1's host->device copy runs while minibatch k's kernels and device->host copy are still
// in flight, and the caller's thread is free to fill the next buffer in between.
static int slot_get(wdx_ctx *ctx, int32_t slot, wdx_ctx **out) {
    if (slot < 0 || slot >= WDX_MAX_SLOTS) {
        set_error("slot must be in [0, %d)", WDX_MAX_SLOTS);
        return WDX_ERR_INVALID;
    }
    if (!ctx->slots[slot]) {
        wdx_ctx *c = nullptr;
        if (int rc = wdx_ctx_create(ctx->device, &c)) return rc;
        ctx->slots[slot] = c;
    }
    *out = ctx->slots[slot];
    return WDX_SUCCESS;
}

int wdx_host_alloc_on(int device, size_t bytes, void **out) {
    DeviceGuard guard(device);   // page-lock under the device that will read the buffer: no stray context on device 0
    if (guard.rc) return guard.rc;
    return wdx_host_alloc(bytes, out);
}

int wdx_host_alloc(size_t bytes, void **out) {
    if (!out) {
        set_error("host_alloc: null output");
        return WDX_ERR_INVALID;
    }
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) {
        set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        *out = nullptr;
        return e == hipErrorNoDevice ? WDX_ERR_NO_DEVICE : WDX_ERR_HIP;
    }
    return WDX_SUCCESS;
}

int wdx_host_register(void *p, size_t bytes) {
    if (!p || bytes == 0) {
        set_error("host_register: bad arguments");
        return WDX_ERR_INVALID;
    }
    WDX_HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    return WDX_SUCCESS;
}

int wdx_host_unregister(void *p) {
    if (!p) return WDX_SUCCESS;
    WDX_HIP_TRY(hipHostUnregister(p));
    return WDX_SUCCESS;
}

int wdx_host_free(void *p) {
    if (p) WDX_HIP_TRY(hipHostFree(p));
    return WDX_SUCCESS;
}

int wdx_demux_submit_ex(wdx_ctx *ctx, int32_t slot, const wdx_minibatch_in *in, const wdx_seg_params *p, int64_t n_refs,
                        uint32_t want) {
    WDX_ENTER(ctx);
    if (!in) {
        set_error("demux_submit: null minibatch");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    const int64_t n_reads = in->n_reads;
    if ((rc = demux_check_args(ctx, "demux_submit", n_reads, in->row_off ? 0 : in->stride, in->sig, in->a_start, in->a_end, p,
                               n_refs)))
        return rc;
    if (in->row_off) {  // packed rows: offsets ascending on 16-byte boundaries, every row inside its slice
        if (n_reads > 0 && !in->row_len) {
            set_error("demux_submit: packed rows need row_len");
            return WDX_ERR_INVALID;
        }
        for (int64_t r = 0; r < n_reads; ++r) {
            const int64_t o0 = in->row_off[r], o1 = in->row_off[r + 1];
            if (o0 < 0 || (o0 & 3) || o1 < o0 || in->row_len[r] < 0 || (int64_t)in->row_len[r] > o1 - o0) {
                set_error("demux_submit: packed row %lld: offsets must ascend in multiples of 4 and hold row_len samples", (long long)r);
                return WDX_ERR_INVALID;
            }
        }
    }
    const bool want_svm = (want & WDX_WANT_SVM) != 0;
    if (want_svm) {
        if (!ctx->svm_set) {
            set_error("demux_submit: WDX_WANT_SVM needs wdx_svm_set_model first");
            return WDX_ERR_NO_REFS;
        }
        if (ctx->refs.nY != ctx->svm.n_train) {
            set_error("reference set has %lld rows but the SVM was trained on %d", (long long)ctx->refs.nY, ctx->svm.n_train);
            return WDX_ERR_INVALID;
        }
    }
    wdx_ctx *S = nullptr;
    if ((rc = slot_get(ctx, slot, &S))) return rc;
    if (S->slot_busy) {
        set_error("demux_submit: slot %d still holds a minibatch (wdx_demux_wait it first)", (int)slot);
        return WDX_ERR_INVALID;
    }
    // the parent's own stream may still be building the reference set this slot is about to read
    if ((rc = use_stream(ctx, ctx->stream))) return rc;
    WDX_HIP_TRY(hipStreamSynchronize(ctx->stream));
    S->knobs = ctx->knobs;
    S->timing = ctx->timing;  // (timed like the parent's own launches; wdx_kernel_time sums the slots' events)
    S->refs = ctx->refs;  // device pointers of the parent's resident set (read-only; wdx_set_refs drains the slots)
    const DtwRefs &R = ctx->refs;
    const int64_t K = p->barcode_num_events;
    const int64_t k = want_svm ? ctx->svm.k : 0;
    // page-locked output block of the slot, 8-byte aligned pieces:
    // [fpt f64 n*K][dwell i64 n*K][stats f64 n*6][prob f64 n*k][conf f64 n][dist f32 n*nY][call i32 n][status i32 n][pred i32 n]
    const size_t n = (size_t)n_reads;
    const size_t bytes[9] = {(want & WDX_WANT_FPT) ? n * K * 8 : 0,
                             (want & WDX_WANT_DWELL) ? n * K * 8 : 0,
                             (want & WDX_WANT_STATS) ? n * 48 : 0,
                             want_svm ? n * (size_t)k * 8 : 0,
                             want_svm ? n * 8 : 0,
                             ((want & WDX_WANT_DIST) && R.nY > 0) ? n * (size_t)R.nY * 4 : 0,
                             n * 4,
                             n * 4,
                             want_svm ? n * 4 : 0};
    size_t off = 0;
    for (int q = 0; q < 9; ++q) {
        S->slot_off[q] = off;
        off += (bytes[q] + 7) / 8 * 8;
    }
    if ((rc = S->pin_out.ensure(off + 8))) return rc;
    unsigned char *ho = (unsigned char *)S->pin_out.p;
    S->slot_n = n_reads;
    S->slot_K = K;
    S->slot_nY = R.nY;
    S->slot_k = k;
    S->slot_want = want & ~(bytes[5] ? 0u : WDX_WANT_DIST);
    if (n_reads == 0) {
        S->slot_busy = true;
        return WDX_SUCCESS;
    }
    MbHostOut H;
    H.fpt = bytes[0] ? (double *)(ho + S->slot_off[0]) : nullptr;
    H.dwell = bytes[1] ? (int64_t *)(ho + S->slot_off[1]) : nullptr;
    H.stats = bytes[2] ? (double *)(ho + S->slot_off[2]) : nullptr;
    H.prob = bytes[3] ? (double *)(ho + S->slot_off[3]) : nullptr;
    H.conf = bytes[4] ? (double *)(ho + S->slot_off[4]) : nullptr;
    H.dist = bytes[5] ? (float *)(ho + S->slot_off[5]) : nullptr;
    H.call = (int32_t *)(ho + S->slot_off[6]);
    H.status = (int32_t *)(ho + S->slot_off[7]);
    H.pred = bytes[8] ? (int32_t *)(ho + S->slot_off[8]) : nullptr;
    StreamDrain drain(S->stream);
    if ((rc = demux_batch_enqueue(S, R, *in, p, H, want_svm ? &ctx->svm : nullptr))) return rc;
    drain.done();  // in flight on purpose: wdx_demux_wait synchronises
    S->slot_busy = true;
    return WDX_SUCCESS;
}

int wdx_demux_submit(wdx_ctx *ctx, int32_t slot, const float *sig, int64_t n_reads, int64_t stride,
                     const int32_t *a_start, const int32_t *a_end, const uint8_t *ok, const wdx_seg_params *p,
                     int64_t n_refs, int32_t want_fpt, int32_t want_dist) {
    const wdx_minibatch_in in{sig, n_reads, stride, nullptr, nullptr, a_start, a_end, ok};
    return wdx_demux_submit_ex(ctx, slot, &in, p, n_refs, (want_fpt ? WDX_WANT_FPT : 0u) | (want_dist ? WDX_WANT_DIST : 0u));
}

int wdx_demux_wait_ex(wdx_ctx *ctx, int32_t slot, const wdx_minibatch_out *out) {
    WDX_ENTER(ctx);
    if (!out) {
        set_error("demux_wait: null output block");
        return WDX_ERR_INVALID;
    }
    wdx_ctx *S = nullptr;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        if (slot < 0 || slot >= WDX_MAX_SLOTS || !ctx->slots[slot] || !ctx->slots[slot]->slot_busy) {
            set_error("demux_wait: nothing was submitted on slot %d", (int)slot);
            return WDX_ERR_INVALID;
        }
        S = ctx->slots[slot];
        // argument errors are reported BEFORE the wait and leave the minibatch in the slot: the caller can wait again
        // with the right arguments; a second thread waiting on the same slot is refused
        if (S->slot_waiting) {
            set_error("demux_wait: another thread is waiting on slot %d", (int)slot);
            return WDX_ERR_INVALID;
        }
        if (S->slot_n > 0 && (!out->call || !out->status)) {
            set_error("demux_wait: call and status are required");
            return WDX_ERR_INVALID;
        }
        const uint32_t w = S->slot_want;
        if (S->slot_n > 0 && ((out->fpt && !(w & WDX_WANT_FPT)) || (out->dist && !(w & WDX_WANT_DIST) && S->slot_nY > 0) ||
                              (out->dwell && !(w & WDX_WANT_DWELL)) || (out->stats && !(w & WDX_WANT_STATS)) ||
                              ((out->prob || out->pred || out->conf) && !(w & WDX_WANT_SVM)))) {
            set_error("demux_wait: an output that was not requested at wdx_demux_submit");
            return WDX_ERR_INVALID;
        }
        S->slot_waiting = true;
    }
    // (the wait itself runs outside the parent's mutex: the other slots can be submitted meanwhile)
    hipError_t e = S->slot_n > 0 ? hipStreamSynchronize(S->stream) : hipSuccess;
    std::lock_guard<std::mutex> g(ctx->mu);
    S->slot_waiting = false;
    S->slot_busy = false;
    if (e != hipSuccess) {
        set_error("demux_wait: hipStreamSynchronize failed: %s", hipGetErrorString(e));
        return WDX_ERR_HIP;
    }
    const int64_t n = S->slot_n;
    if (n == 0) return WDX_SUCCESS;
    const unsigned char *ho = (const unsigned char *)S->pin_out.p;
    const size_t nn = (size_t)n;
    memcpy(out->status, ho + S->slot_off[7], nn * 4);
    if (S->slot_nY > 0) memcpy(out->call, ho + S->slot_off[6], nn * 4);
    else for (int64_t r = 0; r < n; ++r) out->call[r] = -1;
    if (out->fpt) memcpy(out->fpt, ho + S->slot_off[0], nn * S->slot_K * 8);
    if (out->dwell) memcpy(out->dwell, ho + S->slot_off[1], nn * S->slot_K * 8);
    if (out->stats) memcpy(out->stats, ho + S->slot_off[2], nn * 48);
    if (out->prob) memcpy(out->prob, ho + S->slot_off[3], nn * (size_t)S->slot_k * 8);
    if (out->conf) memcpy(out->conf, ho + S->slot_off[4], nn * 8);
    if (out->dist && S->slot_nY > 0) memcpy(out->dist, ho + S->slot_off[5], nn * S->slot_nY * 4);
    if (out->pred) memcpy(out->pred, ho + S->slot_off[8], nn * 4);
    return WDX_SUCCESS;
}

int wdx_demux_wait(wdx_ctx *ctx, int32_t slot, double *fpt, float *dist, int32_t *call, int32_t *status) {
    wdx_minibatch_out out{};
    out.status = status;
    out.call = call;
    out.dist = dist;
    out.fpt = fpt;
    return wdx_demux_wait_ex(ctx, slot, &out);
}

int wdx_svm_set_model(wdx_ctx *ctx, const wdx_svm_model *m) {
    WDX_ENTER(ctx);
    if (!m || m->n_classes < 2 || m->n_classes > 16 || m->n_sv < 1 || m->n_train < 1 || !m->n_support ||
        !m->support || !m->dual_coef || !m->rho || !m->probA || !m->probB || m->pwr_dist < 1) {
        set_error("svm_set_model: need 2..16 classes, support vectors, coefficients and Platt parameters");
        return WDX_ERR_INVALID;
    }
    const int k = m->n_classes, nsv = m->n_sv, np = k * (k - 1) / 2;
    int64_t tot = 0;
    std::vector<int32_t> start(k);
    for (int c = 0; c < k; ++c) {
        if (m->n_support[c] < 0) {
            set_error("svm_set_model: negative n_support");
            return WDX_ERR_INVALID;
        }
        start[c] = (int32_t)tot;
        tot += m->n_support[c];
    }
    if (tot != nsv) {
        set_error("svm_set_model: sum(n_support) != n_sv");
        return WDX_ERR_INVALID;
    }
    for (int s_ = 0; s_ < nsv; ++s_)
        if (m->support[s_] < 0 || m->support[s_] >= m->n_train) {
            set_error("svm_set_model: support index out of range");
            return WDX_ERR_INVALID;
        }
    std::lock_guard<std::mutex> g(ctx->mu);
    if ((rc = use_stream(ctx, ctx->stream))) return rc;
    WDX_HIP_TRY(hipStreamSynchronize(ctx->stream));  // no kernel may still be reading the previous model
    // one device block: [doubles: dual_coef | rho | probA | probB | thresholds][int32: n_support | start | support | label_map]
    const size_t nd = (size_t)(k - 1) * nsv + 3 * (size_t)np + (size_t)k;
    const size_t ni = 3 * (size_t)k + (size_t)nsv;
    ctx->svm_set = false;  // not set until the upload below has succeeded
    if ((rc = ctx->svm_buf.ensure(nd * 8 + ni * 4))) return rc;
    std::vector<unsigned char> h(nd * 8 + ni * 4);
    double *hd = reinterpret_cast<double *>(h.data());
    int32_t *hi = reinterpret_cast<int32_t *>(h.data() + nd * 8);
    size_t o = 0;
    memcpy(hd + o, m->dual_coef, (size_t)(k - 1) * nsv * 8); o += (size_t)(k - 1) * nsv;
    memcpy(hd + o, m->rho, (size_t)np * 8); o += np;
    memcpy(hd + o, m->probA, (size_t)np * 8); o += np;
    memcpy(hd + o, m->probB, (size_t)np * 8); o += np;
    if (m->thresholds) memcpy(hd + o, m->thresholds, (size_t)k * 8);
    memcpy(hi, m->n_support, (size_t)k * 4);
    memcpy(hi + k, start.data(), (size_t)k * 4);
    memcpy(hi + 2 * k, m->support, (size_t)nsv * 4);
    if (m->label_map) memcpy(hi + 2 * k + nsv, m->label_map, (size_t)k * 4);
    WDX_HIP_TRY(hipMemcpy(ctx->svm_buf.p, h.data(), h.size(), hipMemcpyHostToDevice));
    const double *dd = reinterpret_cast<const double *>(ctx->svm_buf.p);
    const int32_t *di = reinterpret_cast<const int32_t *>(reinterpret_cast<const unsigned char *>(ctx->svm_buf.p) + nd * 8);
    SvmDev &S = ctx->svm;
    S.dual_coef = dd;
    S.rho = dd + (size_t)(k - 1) * nsv;
    S.probA = S.rho + np;
    S.probB = S.probA + np;
    S.thresholds = m->thresholds ? S.probB + np : nullptr;
    S.n_support = di;
    S.start = di + k;
    S.support = di + 2 * k;
    S.label_map = m->label_map ? di + 2 * k + nsv : nullptr;
    S.k = k;
    S.n_sv = nsv;
    S.n_train = m->n_train;
    S.pwr = m->pwr_dist;
    S.ngamma = (float)(-m->gamma);
    // for the fused DTW + SVM path (wdx_demux_svm_dev): coefficients vector-major, two chunks per class
    {
        const int H = 2, nch = k * H;
        const size_t cb = (size_t)nsv * (k - 1) * 8, ib = (size_t)(2 * nch + 1) * 4;
        if ((rc = ctx->svm_fused.ensure(cb + ib))) return rc;
        std::vector<unsigned char> hf(cb + ib);
        double *ct = reinterpret_cast<double *>(hf.data());
        for (int s_ = 0; s_ < nsv; ++s_)
            for (int q = 0; q < k - 1; ++q) ct[(size_t)s_ * (k - 1) + q] = m->dual_coef[(size_t)q * nsv + s_];
        int32_t *ref0 = reinterpret_cast<int32_t *>(hf.data() + cb), *slot = ref0 + nch + 1;
        for (int c = 0; c < k; ++c) {
            const int half = (m->n_support[c] + 1) / 2;
            ref0[2 * c] = start[c];
            ref0[2 * c + 1] = start[c] + half;
            slot[2 * c] = 2 * c;
            slot[2 * c + 1] = 2 * c + 1;
        }
        ref0[nch] = nsv;
        WDX_HIP_TRY(hipMemcpy(ctx->svm_fused.p, hf.data(), hf.size(), hipMemcpyHostToDevice));
        ctx->svm_coefT = reinterpret_cast<const double *>(ctx->svm_fused.p);
        ctx->svm_chunk_ref0 = reinterpret_cast<const int32_t *>(reinterpret_cast<const unsigned char *>(ctx->svm_fused.p) + cb);
        ctx->svm_chunk_slot = ctx->svm_chunk_ref0 + nch + 1;
        ctx->svm_chunks = nch;
        ctx->svm_halves = H;
        ++ctx->svm_model_gen;
    }
    ctx->svm_set = true;
    return WDX_SUCCESS;
}

int wdx_svm_predict_dev(wdx_ctx *ctx, const float *d_dist, int64_t n, double *d_prob, int32_t *d_pred,
                        double *d_conf, void *stream) {
    WDX_ENTER(ctx);
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->svm_set) {
        set_error("no SVM model: call wdx_svm_set_model first");
        return WDX_ERR_NO_REFS;
    }
    if (n < 0 || (n > 0 && !d_dist)) {
        set_error("svm_predict_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    if ((rc = use_stream(ctx, (hipStream_t)stream))) return rc;
    Timed t(ctx, WDX_K_SVM, (hipStream_t)stream);
    return launch_svm_predict(ctx->svm, d_dist, n, d_prob, d_pred, d_conf, (hipStream_t)stream, ctx->knobs);
}

int wdx_demux_svm_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off, const int32_t *d_row_len, int64_t stride,
                      int64_t max_len, int64_t n_reads, const int32_t *d_a_start, const int32_t *d_a_end,
                      const uint8_t *d_ok, const wdx_seg_params *p, double *d_fpt, int32_t *d_status, float *d_dist,
                      double *d_prob, int32_t *d_pred, double *d_conf, void *d_work, int64_t block_rows, void *stream) {
    WDX_ENTER(ctx);
    if (n_reads < 0 || !p || block_rows < 0 || (n_reads > 0 && (!d_sig || !d_a_start || !d_a_end || !d_status || !d_work))) {
        set_error("demux_svm_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    DtwRefs &R = ctx->refs;
    if (R.window == 0 || !ctx->svm_set) {
        set_error("demux_svm_dev needs wdx_set_refs and wdx_svm_set_model first");
        return WDX_ERR_NO_REFS;
    }
    if (R.nY != ctx->svm.n_train) {
        set_error("reference set has %lld rows but the SVM was trained on %d", (long long)R.nY, ctx->svm.n_train);
        return WDX_ERR_INVALID;
    }
    const int64_t K = p->barcode_num_events;
    if (K != R.L) {
        set_error("barcode_num_events (%lld) != reference length (%lld)", (long long)K, (long long)R.L);
        return WDX_ERR_INVALID;
    }
    if (n_reads == 0) return WDX_SUCCESS;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = use_stream(ctx, s))) return rc;
    if ((rc = ctx->fp_big.ensure((size_t)fingerprint_big_bytes(max_len)))) return rc;
    const int k = ctx->svm.k;
    // rows per block: the (rows, nY) float32 distances of a block stay in the memory-side cache (<= 96 MiB)
    int64_t rows = block_rows > 0 ? block_rows : (((int64_t)96 << 20) / (4 * R.nY)) / 64 * 64;
    if (rows < 2048) rows = 2048;
    if (rows > n_reads) rows = n_reads;
    const bool fused = !d_dist && R.L == 25 && R.window == 15 && !ctx->knobs.no_short_dtw && !ctx->knobs.svm_scalar && k >= 2 &&
                       k <= 16 && ctx->svm_chunks > 0;
    if (!d_dist && !fused && (rc = ctx->out0.ensure((size_t)(rows * R.nY) * 4))) return rc;
    unsigned char *w = (unsigned char *)d_work;
    double *fpt = d_fpt ? d_fpt : (double *)w;
    void *fp_ws = w + ((n_reads * K * 8 + 255) / 256) * 256;   // (fingerprint workspace behind the fingerprints)
    {
        Timed t(ctx, WDX_K_FINGERPRINT, s);
        if ((rc = launch_fingerprint(d_sig, d_row_off, d_row_len, stride, max_len, n_reads, d_a_start, d_a_end, d_ok, *p,
                                     fpt, nullptr, nullptr, d_status, s, fp_ws, ctx->knobs, &t.n_launches, nullptr, 0, 0,
                                     nullptr, &t.main, (double *)ctx->fp_big.p)))
            return rc;
    }
    // Fused form (the distances are not asked for, the shipped shape): dtw_short_svm_kernel over the references in
    // support-vector order leaves the decision sums P[slot][q][read] -- 16 (k - 1) k bytes per read instead of 4 nY -- and
    // the tail only adds them up, takes the sigmoids and runs the coupling.  No distance matrix, no row blocks.
    if (fused) {
        const SvmDev &M = ctx->svm;
        if (ctx->svm_refs_gen != ctx->refs_gen || ctx->svm_refs_model_gen != ctx->svm_model_gen) {
            const size_t rb = (size_t)M.n_sv * R.Lpad * 8;
            if ((rc = ctx->svm_refs.ensure(rb + (size_t)M.n_sv))) return rc;
            if ((rc = launch_gather_rows(R.pad, R.has_nan, M.support, M.n_sv, R.Lpad, (double *)ctx->svm_refs.p,
                                         (uint8_t *)ctx->svm_refs.p + rb, s)))
                return rc;
            ctx->svm_refs_gen = ctx->refs_gen;
            ctx->svm_refs_model_gen = ctx->svm_model_gen;
        }
        const size_t rb = (size_t)M.n_sv * R.Lpad * 8;
        if ((rc = ctx->out0.ensure((size_t)ctx->svm_chunks * (k - 1) * (size_t)n_reads * 8))) return rc;
        {
            Timed t(ctx, WDX_K_DTW, s);
            if ((rc = launch_dtw_svm_partial(fpt, n_reads, (const double *)ctx->svm_refs.p, R.Lpad, R.halo,
                                             (const uint8_t *)ctx->svm_refs.p + rb, R.L, R.window, R.penalty, ctx->svm_coefT,
                                             ctx->svm_chunk_ref0, ctx->svm_chunk_slot, ctx->svm_chunks, k - 1, M.pwr, M.ngamma,
                                             (double *)ctx->out0.p, s, ctx->knobs.dtw_unfused)))
                return rc;
        }
        {
            Timed t(ctx, WDX_K_SVM, s);
            if ((rc = launch_svm_finish(M, (const double *)ctx->out0.p, ctx->svm_halves, n_reads, d_prob, d_pred, d_conf, s)))
                return rc;
        }
        return launch_svm_mask_failed(d_status, n_reads, k, d_prob, d_pred, d_conf, s);
    }
    for (int64_t r0 = 0; r0 < n_reads; r0 += rows) {
        const int64_t m = std::min(rows, n_reads - r0);
        float *dblk = d_dist ? d_dist + r0 * R.nY : (float *)ctx->out0.p;
        if ((rc = dtw_dev_locked(ctx, fpt + r0 * K, m, dblk, nullptr, s))) return rc;
        Timed t(ctx, WDX_K_SVM, s);
        if ((rc = launch_svm_predict(ctx->svm, dblk, m, d_prob ? d_prob + r0 * k : nullptr, d_pred ? d_pred + r0 : nullptr,
                                     d_conf ? d_conf + r0 : nullptr, s, ctx->knobs)))
            return rc;
    }
    return launch_svm_mask_failed(d_status, n_reads, k, d_prob, d_pred, d_conf, s);
}

int wdx_dtw_svm_predict(wdx_ctx *ctx, const double *X, int64_t n, double *prob, int32_t *pred, double *conf) {
    WDX_ENTER(ctx);
    std::lock_guard<std::mutex> g(ctx->mu);
    DtwRefs &R = ctx->refs;
    if (R.window == 0 || !ctx->svm_set) {
        set_error("dtw_svm_predict needs wdx_set_refs and wdx_svm_set_model first");
        return WDX_ERR_NO_REFS;
    }
    if (R.nY != ctx->svm.n_train) {
        set_error("reference set has %lld rows but the SVM was trained on %d", (long long)R.nY, ctx->svm.n_train);
        return WDX_ERR_INVALID;
    }
    if (n < 0 || (n > 0 && !X)) {
        set_error("dtw_svm_predict: bad arguments");
        return WDX_ERR_INVALID;
    }
    if (n == 0) return WDX_SUCCESS;
    hipStream_t s = ctx->stream;
    if ((rc = use_stream(ctx, s))) return rc;
    const int k = ctx->svm.k;
    // rows per pass: the (rows, nY) float32 distance block stays <= 1 GiB and never leaves HBM
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, ((int64_t)1 << 30) / (4 * (int64_t)R.nY)));
    if ((rc = ctx->in0.ensure((size_t)(chunk * R.L) * 8))) return rc;
    if ((rc = ctx->out0.ensure((size_t)(chunk * R.nY) * 4))) return rc;
    if ((rc = ctx->out1.ensure((size_t)chunk * k * 8))) return rc;
    if ((rc = ctx->out2.ensure((size_t)chunk * 4))) return rc;
    if ((rc = ctx->out3.ensure((size_t)chunk * 8))) return rc;
    StreamDrain drain(s);
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t m = std::min(chunk, n - r0);
        WDX_HIP_TRY(hipMemcpyAsync(ctx->in0.p, X + r0 * R.L, (size_t)(m * R.L) * 8, hipMemcpyHostToDevice, s));
        if ((rc = dtw_dev_locked(ctx, (const double *)ctx->in0.p, m, (float *)ctx->out0.p, nullptr, s))) return rc;
        {
            Timed t(ctx, WDX_K_SVM, s);
            if ((rc = launch_svm_predict(ctx->svm, (const float *)ctx->out0.p, m, (double *)ctx->out1.p,
                                         (int32_t *)ctx->out2.p, (double *)ctx->out3.p, s, ctx->knobs)))
                return rc;
        }
        if (prob) WDX_HIP_TRY(hipMemcpyAsync(prob + r0 * k, ctx->out1.p, (size_t)m * k * 8, hipMemcpyDeviceToHost, s));
        if (pred) WDX_HIP_TRY(hipMemcpyAsync(pred + r0, ctx->out2.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
        if (conf) WDX_HIP_TRY(hipMemcpyAsync(conf + r0, ctx->out3.p, (size_t)m * 8, hipMemcpyDeviceToHost, s));
    }
    WDX_HIP_TRY(hipStreamSynchronize(s));
    drain.done();
    return WDX_SUCCESS;
}

int wdx_kernel_timing(wdx_ctx *ctx, int enable) {
    WDX_ENTER(ctx);
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->timing = enable != 0;
    for (wdx_ctx *S : ctx->slots)   // pipelined minibatches are timed on their slot's events
        if (S) S->timing = ctx->timing;
    return WDX_SUCCESS;
}

int wdx_kernel_time(wdx_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches) {
    WDX_ENTER(ctx);
    if (kernel_id < 0 || kernel_id >= kNumTimed) {
        set_error("kernel id out of range");
        return WDX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    double tot = 0.0;
    int64_t nl = 0;
    // the context's own launches and those of its pipeline slots (wdx_demux_submit enqueues on a slot's stream)
    wdx_ctx *all[1 + WDX_MAX_SLOTS] = {ctx};
    for (int k = 0; k < WDX_MAX_SLOTS; ++k) all[1 + k] = ctx->slots[k];
    for (wdx_ctx *c : all) {
        if (!c) continue;
        for (size_t i = 0; i < c->pending[kernel_id].size(); ++i) {
            auto &e = c->pending[kernel_id][i];
            WDX_HIP_TRY(hipEventSynchronize(e.second));
            float ms = 0;
            WDX_HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
            c->acc_ms[kernel_id] += ms;
            c->launches[kernel_id] += c->pending_launches[kernel_id][i];
            c->pool.push_back(e);
        }
        c->pending[kernel_id].clear();
        c->pending_launches[kernel_id].clear();
        tot += c->acc_ms[kernel_id];
        nl += c->launches[kernel_id];
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = nl;
    return WDX_SUCCESS;
}

int wdx_kernel_time_reset(wdx_ctx *ctx) {
    WDX_ENTER(ctx);
    std::lock_guard<std::mutex> g(ctx->mu);
    wdx_ctx *all[1 + WDX_MAX_SLOTS] = {ctx};
    for (int k = 0; k < WDX_MAX_SLOTS; ++k) all[1 + k] = ctx->slots[k];
    for (wdx_ctx *c : all) {
        if (!c) continue;
        for (int k = 0; k < kNumTimed; ++k) {
            for (auto &e : c->pending[k]) {
                (void)hipEventSynchronize(e.second);
                c->pool.push_back(e);
            }
            c->pending[k].clear();
            c->pending_launches[k].clear();
            c->acc_ms[k] = 0;
            c->launches[k] = 0;
        }
    }
    return WDX_SUCCESS;
}

int wdx_selftest_score_dev(wdx_ctx *ctx, const double *d_dm, const double *d_vs, int64_t n, double *d_fast,
                           double *d_ref, void *stream) {
    WDX_ENTER(ctx);
    if (n < 0 || (n > 0 && (!d_dm || !d_vs || !d_fast || !d_ref))) {
        set_error("selftest_score_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    return launch_score_selftest(d_dm, d_vs, n, d_fast, d_ref, (hipStream_t)stream);
}

int wdx_selftest_clip_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off, int64_t stride, int64_t n_reads,
                          const int32_t *d_a_start, const int32_t *d_a_end, const wdx_seg_params *p, int32_t cap,
                          void *d_rec, void *stream) {
    WDX_ENTER(ctx);
    if (n_reads < 0 || !p || (n_reads > 0 && (!d_sig || !d_a_start || !d_a_end || !d_rec)) ||
        (cap != 4096 && cap != 5120 && cap != 6144 && cap != 8192 && cap != 13312)) {
        set_error("selftest_clip_dev: bad arguments");
        return WDX_ERR_INVALID;
    }
    if (n_reads == 0) return WDX_SUCCESS;
    return launch_clip_bounds_selftest(d_sig, d_row_off, stride, n_reads, d_a_start, d_a_end, *p, cap, d_rec,
                                       (hipStream_t)stream);
}

int wdx_calib_read_dev(wdx_ctx *ctx, const float *d_p, int64_t n, float *d_out, void *stream) {
    WDX_ENTER(ctx);
    return launch_calib_read(d_p, n, d_out, (hipStream_t)stream);
}

int wdx_synth_lengths_dev(wdx_ctx *ctx, uint64_t seed, int64_t first_read, int64_t n_reads,
                          int32_t n_barcodes, const int32_t *d_dwell_table, int64_t *d_len,
                          void *stream) {
    WDX_ENTER(ctx);
    return launch_synth_lengths(seed, first_read, n_reads, n_barcodes, d_dwell_table, d_len,
                                (hipStream_t)stream);
}

int wdx_synth_fill_dev(wdx_ctx *ctx, uint64_t seed, int64_t first_read, int64_t n_reads,
                       int32_t n_barcodes, int32_t n_bc_events, float noise_scale, int32_t spikes,
                       const int32_t *d_dwell_table, const float *d_lead, const float *d_bc,
                       const int64_t *d_off, float *d_sig, int32_t *d_barcode, void *stream) {
    WDX_ENTER(ctx);
    return launch_synth_fill(seed, first_read, n_reads, n_barcodes, n_bc_events, noise_scale, spikes,
                             d_dwell_table, d_lead, d_bc, d_off, d_sig, d_barcode,
                             (hipStream_t)stream);
}

}  // extern "C"
