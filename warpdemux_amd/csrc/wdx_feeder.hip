// Many worker processes, one GPU-facing process: the shared-memory minibatch ring behind include/wdx.h's wdx_feeder_*.
//
// The reference runs `-j 8..16` forked workers (file_proc.py:1197-1243), each of which would drive the GPU from its own
// process.  Sixteen HIP processes on one device collapse (0.86 M reads/s where four reach 2.17 M; a cross-process gate that
// lets only four of them work at a time made it worse -- the device time-slices the processes' queues whether they have
// work or not).  The cure is ONE process that owns the context and keeps up to eight minibatches in flight
// (wdx_demux_submit_ex / wdx_demux_wait_ex) for everybody: the workers copy their minibatch into a slot of a ring in
// shared memory, which the feeder has page-locked, and sleep on the slot until the results are there.  Nothing on the
// workers' side touches HIP -- a worker needs no context and initialises no runtime.
//
// What a worker gets back is what the reference's worker needs from its minibatch (file_proc.py:380-454): the
// ReadResults' arrays (fingerprint, dwell times, six statistics, status: WDX_WANT_FPT / _DWELL / _STATS), the
// nearest-reference call and distance rows, and the DTW_SVM prediction (WDX_WANT_SVM: prob / pred / conf) -- one pass
// over the rows for all of it.  wdx_feeder_predict is model.predict() on fingerprints the worker already holds.
//
// A worker copies only what the kernels read: samples [a_start - padding, a_end + padding) of each row, back to back
// (PACKED rows, wdx_minibatch_in.row_off) -- 18.7 MB instead of 40 MB per 1000-read minibatch on both of the worker's
// copies' sides (its own memcpy and the feeder's DMA).
//
// Slot life cycle.  One futex word per slot: (owner pid << 8) | phase, FREE -> FILLING (a worker owns it) -> READY ->
// INFLIGHT (submitted) -> DONE -> FREE.  Workers claim FREE slots by compare-and-swap -- which records the owner in the
// same atomic -- and sleep on `free_seq` when there is none; the feeder sleeps on `seq` when nothing is READY or in
// flight.  Shared futexes (no FUTEX_PRIVATE_FLAG): the words live in memory mapped by several processes.
//   * a worker that dies while it holds a slot (OOM kill, pool.terminate()): the feeder's idle loop and every claimant
//     that finds the ring full give FILLING / DONE slots whose owner is gone back to the ring;
//   * a feeder that dies without saying so (SIGKILL, an abort inside the runtime) is noticed although it stays a zombie
//     until its parent reaps it: its heartbeat (a thread beside the serve loop stamps CLOCK_MONOTONIC every 20 ms) goes
//     stale, /proc/<pid>/stat says 'Z', or kill(pid, 0) says ESRCH -- whichever a worker sees first;
//   * wdx_feeder_stop drains: INFLIGHT slots are finished and handed over, READY slots that were never submitted are
//     answered WDX_ERR_NO_DEVICE, and only then does the server leave (server_pid = 0).  A waiting worker gives up on
//     "server gone", never on `stop` alone, and never frees a slot the server may still write to; new claims are
//     refused from `stop` on.  A generation counter per slot guards against stale hand-overs.
#include "wdx_ctx.h"

#include <errno.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <signal.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <thread>

namespace wdx {

constexpr uint32_t kFeederMagic = 0x57444632u;  // "WDF2"
// Ring slots and in-flight slots are different things: a ring slot is held by its worker while the worker COPIES its
// minibatch in (milliseconds) and again while it copies the results out, the context keeps at most WDX_MAX_SLOTS of
// them in flight on the device.  With as many ring slots as in-flight slots the ring is what sixteen workers queue for.
constexpr int kMaxRingSlots = WDX_FEEDER_MAX_RING_SLOTS;
enum : uint32_t { kFree = 0, kFilling = 1, kReady = 2, kInflight = 3, kDone = 4 };
enum : uint32_t { kModeRows = 0, kModePredict = 1 };
constexpr long long kHeartbeatStaleNs = 3000000000ll;   // a serving feeder stamps every 20 ms

static inline uint32_t phase_of(uint32_t v) { return v & 0xffu; }
static inline int32_t owner_of(uint32_t v) { return (int32_t)(v >> 8); }
static inline uint32_t word_of(int32_t pid, uint32_t phase) { return ((uint32_t)pid << 8) | phase; }

struct FeederSlot {
    uint32_t state;    // futex word: (owner pid << 8) | phase
    int32_t rc;        // WDX_* of the slot's last minibatch
    int64_t n_reads;
    uint32_t has_ok, want, mode, gen, gen_done, pad_;
    char err[224];     // wdx_last_error() of the feeder for rc != 0
};
static_assert(sizeof(FeederSlot) == 264, "slot record");

struct FeederRing {
    uint32_t magic, n_slots;
    int64_t max_reads, max_stride, n_refs;
    int32_t n_events, n_classes;
    wdx_seg_params params;   // what every minibatch is fingerprinted with (workers read `padding` to pack their rows)
    // byte offsets of the data regions (each holds n_slots consecutive per-slot pieces)
    uint64_t off_sig, off_roff, off_rlen, off_as, off_ae, off_ok, off_dist, off_call, off_status, off_fpt, off_dwell, off_stats,
        off_prob, off_pred, off_conf, bytes;
    uint64_t sig_floats;     // per-slot capacity of the sample region
    uint32_t seq;        // futex: bumped by a worker that made a slot READY
    uint32_t free_seq;   // futex: bumped by whoever made a slot FREE
    uint32_t stop;       // 1: wdx_feeder_stop was called (or the server is leaving)
    int32_t server_pid;  // the feeder process while it serves, else 0
    uint64_t served;     // minibatches handed back (statistics)
    uint64_t reclaimed;  // slots taken back from dead workers (statistics)
    int64_t heartbeat_ns;  // CLOCK_MONOTONIC of the server's latest sign of life
    FeederSlot slot[kMaxRingSlots];
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static long futex(uint32_t *addr, int op, uint32_t val, const struct timespec *to) {
    return syscall(SYS_futex, addr, op, val, to, nullptr, 0);
}
static void futex_wait_ms(uint32_t *addr, uint32_t val, long ms) {
    struct timespec ts{ms / 1000, (ms % 1000) * 1000000L};
    (void)futex(addr, FUTEX_WAIT, val, &ts);   // (EAGAIN: the word changed already; EINTR / ETIMEDOUT: the caller loops)
}
static void futex_wake_all(uint32_t *addr) { (void)futex(addr, FUTEX_WAKE, 0x7fffffff, nullptr); }

static inline uint32_t ld(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void st(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

static long long now_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

// Is process `pid` gone -- exited, or dead and only waiting to be reaped?  kill(pid, 0) keeps answering 0 for a zombie
// (a parent blocked in pool.map never reaps the feeder), so /proc/<pid>/stat's state letter is read as well.
static bool pid_gone(int32_t pid) {
    if (pid <= 0) return true;
    if (kill(pid, 0) != 0 && errno == ESRCH) return true;
    char path[48], buf[256];
    snprintf(path, sizeof(path), "/proc/%d/stat", (int)pid);
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;   // (no /proc here, or reaped this instant: kill() says so on the next poll; the heartbeat covers the server)
    const ssize_t n = read(fd, buf, sizeof(buf) - 1);
    close(fd);
    if (n <= 0) return false;
    buf[n] = 0;
    const char *q = strrchr(buf, ')');   // "pid (comm) S ..." -- comm may contain anything, the last ')' ends it
    return q && q[1] == ' ' && (q[2] == 'Z' || q[2] == 'X');
}

static int ring_check(const FeederRing *R) {
    if (!R || R->magic != kFeederMagic || R->n_slots < 1 || R->n_slots > (uint32_t)kMaxRingSlots) {
        set_error("feeder: not an initialised ring (wdx_feeder_ring_init)");
        return WDX_ERR_INVALID;
    }
    return WDX_SUCCESS;
}

// the server is serving (or about to); false once it has left after a stop, or died
static bool server_up(const FeederRing *R) {
    const int32_t pid = __atomic_load_n(&R->server_pid, __ATOMIC_ACQUIRE);
    if (pid <= 0) return false;
    if (pid_gone(pid)) return false;
    const long long hb = __atomic_load_n(&R->heartbeat_ns, __ATOMIC_ACQUIRE);
    return now_ns() - hb < kHeartbeatStaleNs;
}

// Slots a dead worker left in FILLING / DONE go back to the ring (READY / INFLIGHT ones pass through the server first).
static int reclaim_dead_owners(FeederRing *R) {
    int n = 0;
    for (uint32_t s = 0; s < R->n_slots; ++s) {
        uint32_t v = ld(&R->slot[s].state);
        const uint32_t ph = phase_of(v);
        if ((ph == kFilling || ph == kDone) && pid_gone(owner_of(v)) &&
            __atomic_compare_exchange_n(&R->slot[s].state, &v, (uint32_t)kFree, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
            ++n;
    }
    if (n) {
        __atomic_fetch_add(&R->reclaimed, (uint64_t)n, __ATOMIC_ACQ_REL);
        __atomic_fetch_add(&R->free_seq, 1u, __ATOMIC_ACQ_REL);
        futex_wake_all(&R->free_seq);
    }
    return n;
}

struct Geo {
    size_t sig, roff, rlen, i32, ok, dist, fpt, stats, prob, f64;
};
static bool geometry(const wdx_feeder_geometry *g, Geo &G, size_t &sig_floats) {
    if (!g || g->n_slots < 1 || g->n_slots > kMaxRingSlots || g->max_reads < 1 || g->max_stride < 1 || g->n_refs < 0 ||
        g->n_events < 0 || g->n_classes < 0 || g->n_classes > 16)
        return false;
    const size_t mr = (size_t)g->max_reads;
    // a packed row starts on a 16-byte boundary: up to 3 + 3 floats of slack per row; fingerprints for
    // wdx_feeder_predict (max_reads x n_events doubles) use the same region
    sig_floats = mr * (size_t)((g->max_stride + 3) / 4 * 4 + 4);
    if ((size_t)g->n_events * 2 * mr > sig_floats) sig_floats = (size_t)g->n_events * 2 * mr;
    G.sig = align_up(sig_floats * 4, 4096);
    G.roff = align_up((mr + 1) * 8, 4096);
    G.rlen = G.i32 = align_up(mr * 4, 4096);
    G.ok = align_up(mr, 4096);
    G.dist = align_up(mr * (size_t)(g->n_refs ? g->n_refs : 1) * 4, 4096);
    G.fpt = g->n_events ? align_up(mr * (size_t)g->n_events * 8, 4096) : 0;
    G.stats = g->n_events ? align_up(mr * 48, 4096) : 0;
    G.prob = g->n_classes ? align_up(mr * (size_t)g->n_classes * 8, 4096) : 0;
    G.f64 = g->n_classes ? align_up(mr * 8, 4096) : 0;
    return true;
}

static int claim_slot(FeederRing *R, int &slot);

}  // namespace wdx

using namespace wdx;

extern "C" {

size_t wdx_feeder_ring_bytes(const wdx_feeder_geometry *g) {
    Geo G;
    size_t sf;
    if (!geometry(g, G, sf)) return 0;
    const size_t per_slot = G.sig + G.roff + G.rlen + 2 * G.i32 + G.ok + G.dist + 2 * G.i32 + 2 * G.fpt + G.stats + G.prob +
                            (g->n_classes ? G.i32 : 0) + G.f64;
    return align_up(sizeof(FeederRing), 4096) + (size_t)g->n_slots * per_slot;
}

int wdx_feeder_ring_init(void *mem, size_t bytes, const wdx_feeder_geometry *g, const wdx_seg_params *p) {
    const size_t need = wdx_feeder_ring_bytes(g);
    if (!mem || !p || need == 0 || bytes < need || ((uintptr_t)mem & 4095u)) {
        set_error("feeder_ring_init: need parameters and a page-aligned block of %zu bytes for this geometry", need);
        return WDX_ERR_INVALID;
    }
    if (g->n_events && g->n_events != p->barcode_num_events) {
        set_error("feeder_ring_init: n_events (%d) must equal barcode_num_events (%d)", (int)g->n_events, (int)p->barcode_num_events);
        return WDX_ERR_INVALID;
    }
    Geo G;
    size_t sf;
    (void)geometry(g, G, sf);
    FeederRing *R = (FeederRing *)mem;
    memset(R, 0, sizeof(FeederRing));
    R->n_slots = (uint32_t)g->n_slots;
    R->max_reads = g->max_reads;
    R->max_stride = g->max_stride;
    R->n_refs = g->n_refs;
    R->n_events = g->n_events;
    R->n_classes = g->n_classes;
    R->params = *p;
    R->sig_floats = sf;
    const size_t ns = (size_t)g->n_slots;
    size_t o = align_up(sizeof(FeederRing), 4096);
    R->off_sig = o;    o += ns * G.sig;
    R->off_roff = o;   o += ns * G.roff;
    R->off_rlen = o;   o += ns * G.rlen;
    R->off_as = o;     o += ns * G.i32;
    R->off_ae = o;     o += ns * G.i32;
    R->off_ok = o;     o += ns * G.ok;
    R->off_dist = o;   o += ns * G.dist;
    R->off_call = o;   o += ns * G.i32;
    R->off_status = o; o += ns * G.i32;
    R->off_fpt = o;    o += ns * G.fpt;
    R->off_dwell = o;  o += ns * G.fpt;
    R->off_stats = o;  o += ns * G.stats;
    R->off_prob = o;   o += ns * G.prob;
    R->off_pred = o;   o += g->n_classes ? ns * G.i32 : 0;
    R->off_conf = o;   o += ns * G.f64;
    R->bytes = o;
    __atomic_store_n(&R->magic, kFeederMagic, __ATOMIC_RELEASE);
    return WDX_SUCCESS;
}

int wdx_feeder_stop(void *ring) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    st(&R->stop, 1u);
    __atomic_fetch_add(&R->seq, 1u, __ATOMIC_ACQ_REL);
    futex_wake_all(&R->seq);
    __atomic_fetch_add(&R->free_seq, 1u, __ATOMIC_ACQ_REL);
    futex_wake_all(&R->free_seq);
    // (workers asleep on their slot keep sleeping: the server hands their minibatch over, or answers it, before it leaves)
    return WDX_SUCCESS;
}

int wdx_feeder_alive(void *ring) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    return (!ld(&R->stop) && server_up(R)) ? 1 : 0;
}

int wdx_feeder_served(void *ring, int64_t *minibatches) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    if (minibatches) *minibatches = (int64_t)__atomic_load_n(&R->served, __ATOMIC_ACQUIRE);
    return WDX_SUCCESS;
}

int wdx_feeder_stats(void *ring, int64_t *served, int64_t *reclaimed, int32_t *free_slots) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    if (served) *served = (int64_t)__atomic_load_n(&R->served, __ATOMIC_ACQUIRE);
    if (reclaimed) *reclaimed = (int64_t)__atomic_load_n(&R->reclaimed, __ATOMIC_ACQUIRE);
    if (free_slots) {
        int32_t n = 0;
        for (uint32_t s = 0; s < R->n_slots; ++s) n += phase_of(ld(&R->slot[s].state)) == kFree;
        *free_slots = n;
    }
    return WDX_SUCCESS;
}

// Test hooks (no GPU, no context): what == 1 claims a ring slot for the calling process and leaves it FILLING (returns the
// slot index) -- a worker that then dies is what the reclaim logic exists for; what == 2 makes the calling process pose as
// the serving feeder (server_pid + one heartbeat stamp) without serving -- when it dies unreaped, wdx_feeder_alive and the
// workers must notice.  Never on the product path.
int wdx_feeder_selftest(void *ring, int32_t what) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    if (what == 1) {
        int s = -1;
        if (int rc = claim_slot(R, s)) return rc;
        return s;
    }
    if (what == 2) {
        __atomic_store_n(&R->heartbeat_ns, (int64_t)now_ns(), __ATOMIC_RELEASE);
        __atomic_store_n(&R->server_pid, (int32_t)getpid(), __ATOMIC_RELEASE);
        return WDX_SUCCESS;
    }
    set_error("feeder_selftest: unknown hook");
    return WDX_ERR_INVALID;
}

// The GPU-facing process: serves the ring until wdx_feeder_stop.  The resident reference set of `ctx` (wdx_set_refs) is
// what every minibatch is classified against; a resident SVM (wdx_svm_set_model) serves WDX_WANT_SVM / wdx_feeder_predict.
int wdx_feeder_serve(wdx_ctx *ctx, void *ring) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = ring_check(R)) return rc;
    unsigned char *base = (unsigned char *)ring;
    {
        DeviceGuard guard(ctx->device);
        if (guard.rc) return guard.rc;
        // page-lock the data regions: a minibatch is then copied by DMA at the bus rate and wdx_demux_submit returns at once
        hipError_t e = hipHostRegister(ring, (size_t)R->bytes, hipHostRegisterMapped | hipHostRegisterPortable);
        if (e != hipSuccess) {
            set_error("feeder_serve: hipHostRegister of the ring failed: %s", hipGetErrorString(e));
            return WDX_ERR_HIP;
        }
    }
    // sign of life for the workers: independent of the serve loop, which may sit in a stream synchronisation
    std::atomic<bool> beat_on{true};
    __atomic_store_n(&R->heartbeat_ns, (int64_t)now_ns(), __ATOMIC_RELEASE);
    std::thread beat([&]() {
        while (beat_on.load(std::memory_order_acquire)) {
            __atomic_store_n(&R->heartbeat_ns, (int64_t)now_ns(), __ATOMIC_RELEASE);
            struct timespec ts{0, 20000000L};
            nanosleep(&ts, nullptr);
        }
    });
    __atomic_store_n(&R->server_pid, (int32_t)getpid(), __ATOMIC_RELEASE);
    const pid_t parent = getppid();   // (the process that created the ring: when it is gone, nobody will stop us)
    const size_t mr = (size_t)R->max_reads, nY = (size_t)R->n_refs, K = (size_t)R->n_events, kc = (size_t)R->n_classes;
    auto sig_of = [&](uint32_t s) { return (float *)(base + R->off_sig) + (size_t)s * align_up((size_t)R->sig_floats * 4, 4096) / 4; };
    auto roff_of = [&](uint32_t s) { return (int64_t *)(base + R->off_roff + (size_t)s * align_up((mr + 1) * 8, 4096)); };
    auto i32_of = [&](uint64_t off, uint32_t s) { return (int32_t *)(base + off + (size_t)s * align_up(mr * 4, 4096)); };
    auto ok_of = [&](uint32_t s) { return (uint8_t *)(base + R->off_ok + (size_t)s * align_up(mr, 4096)); };
    auto dist_of = [&](uint32_t s) { return (float *)(base + R->off_dist + (size_t)s * align_up(mr * (nY ? nY : 1) * 4, 4096)); };
    auto fptlike_of = [&](uint64_t off, uint32_t s) { return base + off + (size_t)s * align_up(mr * K * 8, 4096); };
    auto stats_of = [&](uint32_t s) { return (double *)(base + R->off_stats + (size_t)s * align_up(mr * 48, 4096)); };
    auto prob_of = [&](uint32_t s) { return (double *)(base + R->off_prob + (size_t)s * align_up(mr * kc * 8, 4096)); };
    auto conf_of = [&](uint32_t s) { return (double *)(base + R->off_conf + (size_t)s * align_up(mr * 8, 4096)); };
    // ring slots in flight, oldest first, each on one of the context's WDX_MAX_SLOTS submit / wait slots
    struct Fly { int ring, cslot; } fifo[WDX_MAX_SLOTS];
    int head = 0, count = 0;
    bool cbusy[WDX_MAX_SLOTS] = {};
    int rc_fatal = WDX_SUCCESS;
    uint32_t scan_from = 0;   // (READY slots are taken round-robin: no worker is starved by the slots in front of its own)
    auto hand_over = [&](uint32_t s, int rc) {
        FeederSlot &S = R->slot[s];
        S.rc = rc;
        if (rc != WDX_SUCCESS) {
            strncpy(S.err, wdx_last_error(), sizeof(S.err) - 1);
            S.err[sizeof(S.err) - 1] = 0;
        }
        S.gen_done = S.gen;
        __atomic_fetch_add(&R->served, 1ull, __ATOMIC_ACQ_REL);
        st(&S.state, word_of(owner_of(ld(&S.state)), kDone));
        futex_wake_all(&S.state);
    };
    auto finish_oldest = [&]() -> int {
        const Fly f = fifo[head];
        head = (head + 1) % WDX_MAX_SLOTS;
        --count;
        const uint32_t s = (uint32_t)f.ring;
        const uint32_t want = R->slot[s].want;
        wdx_minibatch_out out{};
        out.status = i32_of(R->off_status, s);
        out.call = i32_of(R->off_call, s);
        if ((want & WDX_WANT_DIST) && nY) out.dist = dist_of(s);
        if (want & WDX_WANT_FPT) out.fpt = (double *)fptlike_of(R->off_fpt, s);
        if (want & WDX_WANT_DWELL) out.dwell = (int64_t *)fptlike_of(R->off_dwell, s);
        if (want & WDX_WANT_STATS) out.stats = stats_of(s);
        if (want & WDX_WANT_SVM) {
            out.prob = prob_of(s);
            out.pred = i32_of(R->off_pred, s);
            out.conf = conf_of(s);
        }
        int rc = wdx_demux_wait_ex(ctx, f.cslot, &out);
        if (rc == WDX_ERR_INVALID) {
            // by contract the context slot still holds the minibatch: take it out with the two outputs every batch has,
            // and keep the slot marked busy if even that is refused (it is lost to this serve loop, not reused)
            char keep[224];
            strncpy(keep, wdx_last_error(), sizeof(keep) - 1);
            keep[sizeof(keep) - 1] = 0;
            wdx_minibatch_out two{};
            two.status = out.status;
            two.call = out.call;
            if (wdx_demux_wait_ex(ctx, f.cslot, &two) != WDX_ERR_INVALID) cbusy[f.cslot] = false;
            set_error("%s", keep);
        } else {
            cbusy[f.cslot] = false;
        }
        hand_over(s, rc);
        return rc;
    };
    int idle_rounds = 0;
    while (!ld(&R->stop)) {
        if (getppid() != parent) break;   // orphaned: the parent died without wdx_feeder_stop
        const uint32_t seen = ld(&R->seq);
        bool progressed = false;
        for (uint32_t q = 0; q < R->n_slots && count < WDX_MAX_SLOTS; ++q) {
            const uint32_t s = (scan_from + q) % R->n_slots;
            FeederSlot &S = R->slot[s];
            const uint32_t v = ld(&S.state);
            if (phase_of(v) != kReady) continue;
            progressed = true;
            scan_from = (s + 1) % R->n_slots;
            if (S.mode == kModePredict) {
                // model.predict on fingerprints the worker holds (models/dtw_svm.py:54-98): a synchronous call on the
                // context's own stream -- milliseconds, beside the slots in flight
                st(&S.state, word_of(owner_of(v), kInflight));
                const int rc = wdx_dtw_svm_predict(ctx, (const double *)sig_of(s), S.n_reads, prob_of(s), i32_of(R->off_pred, s),
                                                   conf_of(s));
                hand_over(s, rc);
                continue;
            }
            int cs = 0;
            while (cbusy[cs] && cs < WDX_MAX_SLOTS - 1) ++cs;
            if (cbusy[cs]) break;   // (every context slot lost to refused waits: nothing can be submitted)
            wdx_minibatch_in in{};
            in.sig = sig_of(s);
            in.n_reads = S.n_reads;
            in.stride = 0;
            in.row_off = roff_of(s);
            in.row_len = i32_of(R->off_rlen, s);
            in.a_start = i32_of(R->off_as, s);
            in.a_end = i32_of(R->off_ae, s);
            in.ok = S.has_ok ? ok_of(s) : nullptr;
            const int rc = wdx_demux_submit_ex(ctx, cs, &in, &R->params, (int64_t)nY, S.want);
            if (rc == WDX_SUCCESS) {
                st(&S.state, word_of(owner_of(v), kInflight));
                cbusy[cs] = true;
                fifo[(head + count++) % WDX_MAX_SLOTS] = Fly{(int)s, cs};
            } else {   // (an argument error of this minibatch: its worker gets the code and the message)
                hand_over(s, rc);
            }
        }
        if (count > 0) {
            progressed = true;
            if (finish_oldest() == WDX_ERR_HIP) {   // the device is gone: nothing more can be served
                rc_fatal = WDX_ERR_HIP;
                break;
            }
        }
        if (!progressed) {
            if (++idle_rounds >= 25) {   // ~50 ms of idling: look for slots whose worker died
                idle_rounds = 0;
                (void)reclaim_dead_owners(R);
            }
            futex_wait_ms(&R->seq, seen, 2);
        } else if (++idle_rounds >= 2000) {
            idle_rounds = 0;
            (void)reclaim_dead_owners(R);
        }
    }
    // From here on no new claims are accepted.  Drain: what is in flight is finished and handed over; what was READY
    // but never submitted is answered, so that no worker sleeps on a slot that will never change; then the server
    // leaves (server_pid = 0), which is what the waiting workers give up on.
    st(&R->stop, 1u);
    while (count > 0) (void)finish_oldest();
    for (uint32_t s = 0; s < R->n_slots; ++s) {
        if (phase_of(ld(&R->slot[s].state)) == kReady) {
            set_error("feeder: stopped before this minibatch was submitted");
            hand_over(s, WDX_ERR_NO_DEVICE);
        }
    }
    __atomic_store_n(&R->server_pid, 0, __ATOMIC_RELEASE);
    beat_on.store(false, std::memory_order_release);
    beat.join();
    for (uint32_t s = 0; s < R->n_slots; ++s) futex_wake_all(&R->slot[s].state);
    __atomic_fetch_add(&R->free_seq, 1u, __ATOMIC_ACQ_REL);
    futex_wake_all(&R->free_seq);
    {
        DeviceGuard guard(ctx->device);
        (void)hipHostUnregister(ring);
    }
    if (rc_fatal) set_error("feeder_serve: the device failed while a minibatch was in flight");
    return rc_fatal;
}

}  // extern "C"

namespace wdx {

// claim a FREE slot for this process; WDX_ERR_NO_DEVICE once the feeder has stopped or died
static int claim_slot(FeederRing *R, int &slot) {
    const int32_t me = (int32_t)getpid();
    int rounds = 0;
    for (;;) {
        if (ld(&R->stop)) {
            set_error("feeder: the feeder has stopped");
            return WDX_ERR_NO_DEVICE;
        }
        const uint32_t seen = ld(&R->free_seq);
        const uint32_t first = (uint32_t)me % R->n_slots;
        for (uint32_t k = 0; k < R->n_slots; ++k) {
            const uint32_t c = (first + k) % R->n_slots;
            uint32_t expect = kFree;
            if (__atomic_compare_exchange_n(&R->slot[c].state, &expect, word_of(me, kFilling), false, __ATOMIC_ACQ_REL,
                                            __ATOMIC_ACQUIRE)) {
                slot = (int)c;
                return WDX_SUCCESS;
            }
        }
        // (a server that has not come up yet -- server_pid still 0, no stop -- is waited for: the parent starts it first)
        if (__atomic_load_n(&R->server_pid, __ATOMIC_ACQUIRE) != 0 && !server_up(R)) {
            set_error("feeder: the feeder process died");
            return WDX_ERR_NO_DEVICE;
        }
        if ((++rounds & 3) == 0) (void)reclaim_dead_owners(R);   // the ring is full: is a dead worker sitting on a slot?
        futex_wait_ms(&R->free_seq, seen, 50);
    }
}

static void release_slot(FeederRing *R, int s) {
    st(&R->slot[s].state, (uint32_t)kFree);
    __atomic_fetch_add(&R->free_seq, 1u, __ATOMIC_ACQ_REL);
    futex_wake_all(&R->free_seq);
}

// READY -> sleep until DONE.  Gives up when the server is really gone: it has left after a stop (its drain answers every
// slot first), or it died.  A slot the server may still be writing to is never freed here.
static int publish_and_wait(FeederRing *R, int s, uint32_t my_gen) {
    FeederSlot &S = R->slot[s];
    const int32_t me = (int32_t)getpid();
    st(&S.state, word_of(me, kReady));
    __atomic_fetch_add(&R->seq, 1u, __ATOMIC_ACQ_REL);
    futex_wake_all(&R->seq);
    for (;;) {
        const uint32_t v = ld(&S.state);
        if (phase_of(v) == kDone) break;
        const int32_t pid = __atomic_load_n(&R->server_pid, __ATOMIC_ACQUIRE);
        const bool left = pid == 0 && ld(&R->stop);           // the drain is over and did not answer this slot
        const bool died = pid != 0 && !server_up(R);
        if ((left || died) && phase_of(ld(&S.state)) != kDone) {
            set_error(died ? "feeder: the feeder process died while the minibatch was in its hands"
                           : "feeder: the feeder stopped before the minibatch was served");
            // READY was never picked up and the server is gone for good: the slot is ours to give back.  INFLIGHT under a
            // dead server is left as it is (nothing serves this ring any more).
            if (phase_of(ld(&S.state)) == kReady) release_slot(R, s);
            return WDX_ERR_NO_DEVICE;
        }
        futex_wait_ms(&S.state, v, 100);
    }
    if (S.gen_done != my_gen) {   // (cannot happen while slots are only freed by their owners; a stale hand-over is not a result)
        set_error("feeder: slot %d was handed over for generation %u, not %u", s, S.gen_done, my_gen);
        release_slot(R, s);
        return WDX_ERR_INVALID;
    }
    return WDX_SUCCESS;
}

}  // namespace wdx

extern "C" {

// A worker process: one minibatch through the feeder -- no context, NO HIP call.  Blocks until the results are there.
int wdx_feeder_run(void *ring, const wdx_feeder_job *job) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    if (!job) {
        set_error("feeder_run: null job");
        return WDX_ERR_INVALID;
    }
    const int64_t n_reads = job->n_reads, stride = job->stride;
    const uint32_t want = job->want;
    if (n_reads < 0 || stride < 0 || (n_reads > 0 && (!job->sig || !job->a_start || !job->a_end || !job->status))) {
        set_error("feeder_run: bad arguments");
        return WDX_ERR_INVALID;
    }
    if (n_reads > R->max_reads) {
        set_error("feeder_run: %lld reads do not fit the ring's %lld-read slots", (long long)n_reads, (long long)R->max_reads);
        return WDX_ERR_INVALID;
    }
    if ((want & ~(WDX_WANT_FPT | WDX_WANT_DIST | WDX_WANT_DWELL | WDX_WANT_STATS | WDX_WANT_SVM)) ||
        ((want & (WDX_WANT_FPT | WDX_WANT_DWELL | WDX_WANT_STATS)) && R->n_events == 0) || ((want & WDX_WANT_SVM) && R->n_classes == 0)) {
        set_error("feeder_run: the ring was laid out without room for an output that is asked for (n_events %d, n_classes %d)",
                  (int)R->n_events, (int)R->n_classes);
        return WDX_ERR_INVALID;
    }
    if (n_reads > 0 && (((want & WDX_WANT_FPT) && !job->fpt) || ((want & WDX_WANT_DWELL) && !job->dwell) ||
                        ((want & WDX_WANT_STATS) && !job->stats) || ((want & WDX_WANT_DIST) && R->n_refs > 0 && !job->dist) ||
                        ((want & WDX_WANT_SVM) && (!job->prob || !job->pred || !job->conf)))) {
        set_error("feeder_run: an output that is asked for has no destination");
        return WDX_ERR_INVALID;
    }
    if (n_reads == 0) return WDX_SUCCESS;
    // the windows, packed: row r = samples [st & ~3, en) of the caller's row, st = a_start - padding, en = a_end + padding
    // clamped to the row (extract_adapter, sig_proc.py:388-389); every row starts on a 16-byte boundary
    const int64_t pad = R->params.padding;
    int64_t need = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        int64_t s0 = (int64_t)job->a_start[r] - pad, e0 = (int64_t)job->a_end[r] + pad;
        if (s0 < 0) s0 = 0;
        if (e0 > stride) e0 = stride;
        if (e0 < s0 || (job->ok && !job->ok[r])) e0 = s0;
        s0 &= ~(int64_t)3;
        need += ((e0 - s0) + 3) & ~(int64_t)3;
    }
    if ((uint64_t)need > R->sig_floats) {
        set_error("feeder_run: the minibatch's adapter windows (%lld samples) do not fit a ring slot (%llu)", (long long)need,
                  (unsigned long long)R->sig_floats);
        return WDX_ERR_INVALID;
    }
    int s = -1;
    if (int rc = claim_slot(R, s)) return rc;
    unsigned char *base = (unsigned char *)ring;
    FeederSlot &S = R->slot[s];
    const size_t mr = (size_t)R->max_reads, nY = (size_t)R->n_refs, K = (size_t)R->n_events, kc = (size_t)R->n_classes;
    float *dsig = (float *)(base + R->off_sig + (size_t)s * align_up((size_t)R->sig_floats * 4, 4096));
    int64_t *roff = (int64_t *)(base + R->off_roff + (size_t)s * align_up((mr + 1) * 8, 4096));
    auto i32_of = [&](uint64_t off) { return (int32_t *)(base + off + (size_t)s * align_up(mr * 4, 4096)); };
    int32_t *rlen = i32_of(R->off_rlen), *ras = i32_of(R->off_as), *rae = i32_of(R->off_ae);
    int64_t acc = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        int64_t s0 = (int64_t)job->a_start[r] - pad, e0 = (int64_t)job->a_end[r] + pad;
        if (s0 < 0) s0 = 0;
        if (e0 > stride) e0 = stride;
        if (e0 < s0 || (job->ok && !job->ok[r])) e0 = s0;
        s0 &= ~(int64_t)3;
        roff[r] = acc;
        rlen[r] = (int32_t)(e0 - s0);
        ras[r] = job->a_start[r] - (int32_t)s0;
        rae[r] = job->a_end[r] - (int32_t)s0;
        if (e0 > s0) memcpy(dsig + acc, job->sig + r * stride + s0, (size_t)(e0 - s0) * 4);
        acc += ((e0 - s0) + 3) & ~(int64_t)3;
    }
    roff[n_reads] = acc;
    if (job->ok) memcpy(base + R->off_ok + (size_t)s * align_up(mr, 4096), job->ok, (size_t)n_reads);
    S.n_reads = n_reads;
    S.has_ok = job->ok ? 1u : 0u;
    S.want = want;
    S.mode = kModeRows;
    S.rc = WDX_SUCCESS;
    const uint32_t my_gen = ++S.gen;
    if (int rc = publish_and_wait(R, s, my_gen)) return rc;
    int rc = S.rc;
    if (rc == WDX_SUCCESS) {
        const size_t n = (size_t)n_reads;
        memcpy(job->status, i32_of(R->off_status), n * 4);
        if (job->call) memcpy(job->call, i32_of(R->off_call), n * 4);
        if ((want & WDX_WANT_DIST) && nY)
            memcpy(job->dist, base + R->off_dist + (size_t)s * align_up(mr * nY * 4, 4096), n * nY * 4);
        if (want & WDX_WANT_FPT) memcpy(job->fpt, base + R->off_fpt + (size_t)s * align_up(mr * K * 8, 4096), n * K * 8);
        if (want & WDX_WANT_DWELL) memcpy(job->dwell, base + R->off_dwell + (size_t)s * align_up(mr * K * 8, 4096), n * K * 8);
        if (want & WDX_WANT_STATS) memcpy(job->stats, base + R->off_stats + (size_t)s * align_up(mr * 48, 4096), n * 48);
        if (want & WDX_WANT_SVM) {
            memcpy(job->prob, base + R->off_prob + (size_t)s * align_up(mr * kc * 8, 4096), n * kc * 8);
            memcpy(job->pred, i32_of(R->off_pred), n * 4);
            memcpy(job->conf, base + R->off_conf + (size_t)s * align_up(mr * 8, 4096), n * 8);
        }
    } else {
        set_error("feeder: %s", S.err);
    }
    release_slot(R, s);
    return rc;
}

// wdx_demux_batch's arguments and outputs (bit-identical results) through the feeder
int wdx_feeder_demux(void *ring, const float *sig, int64_t n_reads, int64_t stride, const int32_t *a_start,
                     const int32_t *a_end, const uint8_t *ok, int64_t n_refs, float *dist, int32_t *call, int32_t *status) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    if (n_refs != R->n_refs) {
        set_error("feeder_demux: the caller sized `dist` for %lld references but the ring serves %lld", (long long)n_refs,
                  (long long)R->n_refs);
        return WDX_ERR_INVALID;
    }
    if (n_reads > 0 && !call) {
        set_error("feeder_demux: bad arguments");
        return WDX_ERR_INVALID;
    }
    wdx_feeder_job job{};
    job.sig = sig;
    job.n_reads = n_reads;
    job.stride = stride;
    job.a_start = a_start;
    job.a_end = a_end;
    job.ok = ok;
    job.want = dist ? WDX_WANT_DIST : 0u;
    job.status = status;
    job.call = call;
    job.dist = dist;
    return wdx_feeder_run(ring, &job);
}

// DTW_SVM.predict (models/dtw_svm.py:54-98) on fingerprints the worker holds: X (n, n_events) float64 -> prob / pred / conf
int wdx_feeder_predict(void *ring, const double *X, int64_t n, double *prob, int32_t *pred, double *conf) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    if (n < 0 || (n > 0 && (!X || !prob || !pred || !conf))) {
        set_error("feeder_predict: bad arguments");
        return WDX_ERR_INVALID;
    }
    if (R->n_classes == 0 || R->n_events == 0) {
        set_error("feeder_predict: the ring was laid out without a model (n_classes / n_events)");
        return WDX_ERR_INVALID;
    }
    unsigned char *base = (unsigned char *)ring;
    const size_t mr = (size_t)R->max_reads, K = (size_t)R->n_events, kc = (size_t)R->n_classes;
    for (int64_t r0 = 0; r0 < n; r0 += R->max_reads) {   // (model.predict takes any number of rows: slot-sized pieces)
        const int64_t m = n - r0 < R->max_reads ? n - r0 : R->max_reads;
        int s = -1;
        if (int rc = claim_slot(R, s)) return rc;
        FeederSlot &S = R->slot[s];
        memcpy(base + R->off_sig + (size_t)s * align_up((size_t)R->sig_floats * 4, 4096), X + r0 * (int64_t)K, (size_t)m * K * 8);
        S.n_reads = m;
        S.has_ok = 0u;
        S.want = WDX_WANT_SVM;
        S.mode = kModePredict;
        S.rc = WDX_SUCCESS;
        const uint32_t my_gen = ++S.gen;
        if (int rc = publish_and_wait(R, s, my_gen)) return rc;
        const int rc = S.rc;
        if (rc == WDX_SUCCESS) {
            memcpy(prob + r0 * (int64_t)kc, base + R->off_prob + (size_t)s * align_up(mr * kc * 8, 4096), (size_t)m * kc * 8);
            memcpy(pred + r0, base + R->off_pred + (size_t)s * align_up(mr * 4, 4096), (size_t)m * 4);
            memcpy(conf + r0, base + R->off_conf + (size_t)s * align_up(mr * 8, 4096), (size_t)m * 8);
        } else {
            set_error("feeder: %s", S.err);
        }
        release_slot(R, s);
        if (rc) return rc;
    }
    return WDX_SUCCESS;
}

}  // extern "C"
