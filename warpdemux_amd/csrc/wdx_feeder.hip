// Many worker processes, one GPU-facing process: the shared-memory minibatch ring behind include/wdx.h's wdx_feeder_*.
//
// The reference runs `-j 8..16` forked workers (file_proc.py:1197-1243), each of which would drive the GPU from its own
// process.  Sixteen HIP processes on one device collapse (0.86 M reads/s where four reach 2.17 M; a cross-process gate that
// lets only four of them work at a time made it worse -- the device time-slices the processes' queues whether they have
// work or not).  The cure is ONE process that owns the context and keeps up to eight minibatches in flight
// (wdx_demux_submit / wdx_demux_wait) for everybody: the workers copy their minibatch into a slot of a ring in shared
// memory, which the feeder has page-locked, and sleep on the slot until the results are there.  Nothing on the workers'
// side touches HIP -- a worker needs no context and initialises no runtime.
//
// Slot life cycle (one futex word per slot): FREE -> FILLING (a worker owns it) -> READY -> INFLIGHT (submitted) -> DONE
// -> FREE.  Workers claim FREE slots by compare-and-swap and sleep on `free_seq` when there is none; the feeder sleeps on
// `seq` when nothing is READY or in flight.  Shared futexes (no FUTEX_PRIVATE_FLAG): the words live in memory mapped by
// several processes.
#include "wdx_ctx.h"

#include <errno.h>
#include <linux/futex.h>
#include <signal.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

namespace wdx {

constexpr uint32_t kFeederMagic = 0x57444658u;  // "WDFX"
// Ring slots and in-flight slots are different things: a ring slot is held by its worker while the worker COPIES its
// minibatch in (40 MB: milliseconds) and again while it copies the results out, the context keeps at most WDX_MAX_SLOTS of
// them in flight on the device.  With as many ring slots as in-flight slots the ring is what sixteen workers queue for.
constexpr int kMaxRingSlots = WDX_FEEDER_MAX_RING_SLOTS;
enum : uint32_t { kFree = 0, kFilling = 1, kReady = 2, kInflight = 3, kDone = 4 };

struct FeederSlot {
    uint32_t state;    // futex word
    int32_t rc;        // WDX_* of the slot's last minibatch
    int64_t n_reads, stride;
    uint32_t has_ok, pad_;
    char err[232];     // wdx_last_error() of the feeder for rc != 0
};
static_assert(sizeof(FeederSlot) == 264, "slot record");

struct FeederRing {
    uint32_t magic, n_slots;
    int64_t max_reads, max_stride, n_refs;
    uint64_t off_sig, off_as, off_ae, off_ok, off_dist, off_call, off_status, bytes;   // byte offsets of the data regions
    uint32_t seq;        // futex: bumped by a worker that made a slot READY
    uint32_t free_seq;   // futex: bumped by a worker that made a slot FREE
    uint32_t stop;       // 1: wdx_feeder_stop was called
    int32_t server_pid;  // the feeder process while it serves, else 0
    uint64_t served;     // minibatches handed back (statistics)
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

static int ring_check(const FeederRing *R) {
    if (!R || R->magic != kFeederMagic || R->n_slots < 1 || R->n_slots > (uint32_t)kMaxRingSlots) {
        set_error("feeder: not an initialised ring (wdx_feeder_ring_init)");
        return WDX_ERR_INVALID;
    }
    return WDX_SUCCESS;
}

}  // namespace wdx

using namespace wdx;

extern "C" {

size_t wdx_feeder_ring_bytes(int32_t n_slots, int64_t max_reads, int64_t max_stride, int64_t n_refs) {
    if (n_slots < 1 || n_slots > kMaxRingSlots || max_reads < 1 || max_stride < 1 || n_refs < 0) return 0;
    size_t b = align_up(sizeof(FeederRing), 4096);
    b += align_up((size_t)n_slots * (size_t)max_reads * (size_t)max_stride * 4, 4096);   // signals
    b += 2 * align_up((size_t)n_slots * (size_t)max_reads * 4, 4096);                    // adapter_start, adapter_end
    b += align_up((size_t)n_slots * (size_t)max_reads, 4096);                            // success flags
    b += align_up((size_t)n_slots * (size_t)max_reads * (size_t)(n_refs ? n_refs : 1) * 4, 4096);  // distances
    b += 2 * align_up((size_t)n_slots * (size_t)max_reads * 4, 4096);                    // call, status
    return b;
}

int wdx_feeder_ring_init(void *mem, size_t bytes, int32_t n_slots, int64_t max_reads, int64_t max_stride, int64_t n_refs) {
    const size_t need = wdx_feeder_ring_bytes(n_slots, max_reads, max_stride, n_refs);
    if (!mem || need == 0 || bytes < need || ((uintptr_t)mem & 4095u)) {
        set_error("feeder_ring_init: need a page-aligned block of %zu bytes for this geometry", need);
        return WDX_ERR_INVALID;
    }
    FeederRing *R = (FeederRing *)mem;
    memset(R, 0, sizeof(FeederRing));
    R->n_slots = (uint32_t)n_slots;
    R->max_reads = max_reads;
    R->max_stride = max_stride;
    R->n_refs = n_refs;
    size_t o = align_up(sizeof(FeederRing), 4096);
    R->off_sig = o;    o += align_up((size_t)n_slots * (size_t)max_reads * (size_t)max_stride * 4, 4096);
    R->off_as = o;     o += align_up((size_t)n_slots * (size_t)max_reads * 4, 4096);
    R->off_ae = o;     o += align_up((size_t)n_slots * (size_t)max_reads * 4, 4096);
    R->off_ok = o;     o += align_up((size_t)n_slots * (size_t)max_reads, 4096);
    R->off_dist = o;   o += align_up((size_t)n_slots * (size_t)max_reads * (size_t)(n_refs ? n_refs : 1) * 4, 4096);
    R->off_call = o;   o += align_up((size_t)n_slots * (size_t)max_reads * 4, 4096);
    R->off_status = o; o += align_up((size_t)n_slots * (size_t)max_reads * 4, 4096);
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
    for (uint32_t s = 0; s < R->n_slots; ++s) futex_wake_all(&R->slot[s].state);
    return WDX_SUCCESS;
}

int wdx_feeder_alive(void *ring) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    const int32_t pid = __atomic_load_n(&R->server_pid, __ATOMIC_ACQUIRE);
    return (!ld(&R->stop) && pid > 0 && !(kill(pid, 0) != 0 && errno == ESRCH)) ? 1 : 0;
}

int wdx_feeder_served(void *ring, int64_t *minibatches) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    if (minibatches) *minibatches = (int64_t)__atomic_load_n(&R->served, __ATOMIC_ACQUIRE);
    return WDX_SUCCESS;
}

// The GPU-facing process: serves the ring until wdx_feeder_stop.  The resident reference set of `ctx` (wdx_set_refs) is
// what every minibatch is classified against.
int wdx_feeder_serve(wdx_ctx *ctx, void *ring, const wdx_seg_params *p) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = check_ctx(ctx)) return rc;
    if (int rc = ring_check(R)) return rc;
    if (!p) {
        set_error("feeder_serve: null parameters");
        return WDX_ERR_INVALID;
    }
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
    __atomic_store_n(&R->server_pid, (int32_t)getpid(), __ATOMIC_RELEASE);
    const pid_t parent = getppid();   // (the process that created the ring: when it is gone, nobody will stop us)
    const int64_t mr = R->max_reads, nY = R->n_refs;
    // ring slots in flight, oldest first, each on one of the context's WDX_MAX_SLOTS submit / wait slots
    struct Fly { int ring, cslot; } fifo[WDX_MAX_SLOTS];
    int head = 0, count = 0;
    bool cbusy[WDX_MAX_SLOTS] = {};
    int rc_fatal = WDX_SUCCESS;
    uint32_t scan_from = 0;   // (READY slots are taken round-robin: no worker is starved by the slots in front of its own)
    auto finish_oldest = [&]() -> int {
        const Fly f = fifo[head];
        head = (head + 1) % WDX_MAX_SLOTS;
        --count;
        FeederSlot &S = R->slot[f.ring];
        const int rc = wdx_demux_wait(ctx, f.cslot, nullptr, (float *)(base + R->off_dist) + (size_t)f.ring * mr * (nY ? nY : 1),
                                      (int32_t *)(base + R->off_call) + (size_t)f.ring * mr,
                                      (int32_t *)(base + R->off_status) + (size_t)f.ring * mr);
        cbusy[f.cslot] = false;
        S.rc = rc;
        if (rc != WDX_SUCCESS) {
            strncpy(S.err, wdx_last_error(), sizeof(S.err) - 1);
            S.err[sizeof(S.err) - 1] = 0;
        }
        __atomic_fetch_add(&R->served, 1ull, __ATOMIC_ACQ_REL);
        st(&S.state, kDone);
        futex_wake_all(&S.state);
        return rc;
    };
    while (!ld(&R->stop)) {
        if (getppid() != parent) break;   // orphaned: the parent died without wdx_feeder_stop
        const uint32_t seen = ld(&R->seq);
        bool progressed = false;
        for (uint32_t q = 0; q < R->n_slots && count < WDX_MAX_SLOTS; ++q) {
            const uint32_t s = (scan_from + q) % R->n_slots;
            FeederSlot &S = R->slot[s];
            if (ld(&S.state) != kReady) continue;
            int cs = 0;
            while (cbusy[cs]) ++cs;   // (count < WDX_MAX_SLOTS: one is free)
            const int rc = wdx_demux_submit(ctx, cs, (const float *)(base + R->off_sig) + (size_t)s * mr * R->max_stride,
                                            S.n_reads, S.stride, (const int32_t *)(base + R->off_as) + (size_t)s * mr,
                                            (const int32_t *)(base + R->off_ae) + (size_t)s * mr,
                                            S.has_ok ? (const uint8_t *)(base + R->off_ok) + (size_t)s * mr : nullptr, p, nY, 0, 1);
            progressed = true;
            scan_from = (s + 1) % R->n_slots;
            if (rc == WDX_SUCCESS) {
                st(&S.state, kInflight);
                cbusy[cs] = true;
                fifo[(head + count++) % WDX_MAX_SLOTS] = Fly{(int)s, cs};
            } else {   // (an argument error of this minibatch: its worker gets the code and the message)
                S.rc = rc;
                strncpy(S.err, wdx_last_error(), sizeof(S.err) - 1);
                S.err[sizeof(S.err) - 1] = 0;
                st(&S.state, kDone);
                futex_wake_all(&S.state);
            }
        }
        if (count > 0) {
            progressed = true;
            if (finish_oldest() == WDX_ERR_HIP) {   // the device is gone: nothing more can be served
                rc_fatal = WDX_ERR_HIP;
                break;
            }
        }
        if (!progressed) futex_wait_ms(&R->seq, seen, 2);
    }
    // drain what is still in flight so that no worker sleeps on a slot that will never change
    while (count > 0) (void)finish_oldest();
    st(&R->stop, 1u);
    __atomic_store_n(&R->server_pid, 0, __ATOMIC_RELEASE);
    for (uint32_t s = 0; s < R->n_slots; ++s) futex_wake_all(&R->slot[s].state);
    futex_wake_all(&R->free_seq);
    {
        DeviceGuard guard(ctx->device);
        (void)hipHostUnregister(ring);
    }
    if (rc_fatal) set_error("feeder_serve: the device failed while a minibatch was in flight");
    return rc_fatal;
}

// A worker process: one minibatch through the feeder -- the drop-in for wdx_demux_batch (same outputs, bit for bit)
// that needs no context and makes no HIP call.  Blocks until the results are there.
int wdx_feeder_demux(void *ring, const float *sig, int64_t n_reads, int64_t stride, const int32_t *a_start,
                     const int32_t *a_end, const uint8_t *ok, int64_t n_refs, float *dist, int32_t *call, int32_t *status) {
    FeederRing *R = (FeederRing *)ring;
    if (int rc = ring_check(R)) return rc;
    if (n_reads < 0 || stride < 0 || (n_reads > 0 && (!sig || !a_start || !a_end || !call || !status))) {
        set_error("feeder_demux: bad arguments");
        return WDX_ERR_INVALID;
    }
    if (n_reads > R->max_reads || n_reads * stride > R->max_reads * R->max_stride) {
        set_error("feeder_demux: a (%lld, %lld) minibatch does not fit the ring's (%lld, %lld) slots", (long long)n_reads,
                  (long long)stride, (long long)R->max_reads, (long long)R->max_stride);
        return WDX_ERR_INVALID;
    }
    if (n_refs != R->n_refs) {
        set_error("feeder_demux: the caller sized `dist` for %lld references but the ring serves %lld", (long long)n_refs,
                  (long long)R->n_refs);
        return WDX_ERR_INVALID;
    }
    if (n_reads == 0) return WDX_SUCCESS;
    auto feeder_gone = [&]() -> bool {
        if (ld(&R->stop)) return true;
        const int32_t pid = __atomic_load_n(&R->server_pid, __ATOMIC_ACQUIRE);
        return pid > 0 && kill(pid, 0) != 0 && errno == ESRCH;   // (it died without saying so)
    };
    // claim a slot
    int s = -1;
    for (;;) {
        const uint32_t seen = ld(&R->free_seq);
        const uint32_t first = (uint32_t)getpid() % R->n_slots;
        for (uint32_t k = 0; k < R->n_slots && s < 0; ++k) {
            const uint32_t c = (first + k) % R->n_slots;
            uint32_t expect = kFree;
            if (__atomic_compare_exchange_n(&R->slot[c].state, &expect, kFilling, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) s = (int)c;
        }
        if (s >= 0) break;
        if (feeder_gone()) {
            set_error("feeder_demux: the feeder has stopped");
            return WDX_ERR_NO_DEVICE;
        }
        futex_wait_ms(&R->free_seq, seen, 50);
    }
    unsigned char *base = (unsigned char *)ring;
    FeederSlot &S = R->slot[s];
    const int64_t mr = R->max_reads, nY = R->n_refs;
    memcpy((float *)(base + R->off_sig) + (size_t)s * mr * R->max_stride, sig, (size_t)n_reads * (size_t)stride * 4);
    memcpy((int32_t *)(base + R->off_as) + (size_t)s * mr, a_start, (size_t)n_reads * 4);
    memcpy((int32_t *)(base + R->off_ae) + (size_t)s * mr, a_end, (size_t)n_reads * 4);
    if (ok) memcpy(base + R->off_ok + (size_t)s * mr, ok, (size_t)n_reads);
    S.n_reads = n_reads;
    S.stride = stride;
    S.has_ok = ok ? 1u : 0u;
    S.rc = WDX_SUCCESS;
    st(&S.state, kReady);
    __atomic_fetch_add(&R->seq, 1u, __ATOMIC_ACQ_REL);
    futex_wake_all(&R->seq);
    // sleep until the feeder hands it back
    int rc = WDX_SUCCESS;
    for (;;) {
        const uint32_t v = ld(&S.state);
        if (v == kDone) break;
        if (feeder_gone() && ld(&S.state) != kDone) {
            set_error("feeder_demux: the feeder stopped while the minibatch was in its hands");
            rc = WDX_ERR_NO_DEVICE;
            break;
        }
        futex_wait_ms(&S.state, v, 100);
    }
    if (rc == WDX_SUCCESS) {
        rc = S.rc;
        if (rc == WDX_SUCCESS) {
            memcpy(status, (int32_t *)(base + R->off_status) + (size_t)s * mr, (size_t)n_reads * 4);
            memcpy(call, (int32_t *)(base + R->off_call) + (size_t)s * mr, (size_t)n_reads * 4);
            if (dist && nY > 0) memcpy(dist, (float *)(base + R->off_dist) + (size_t)s * mr * nY, (size_t)n_reads * (size_t)nY * 4);
        } else {
            set_error("feeder: %s", S.err);
        }
    }
    st(&S.state, kFree);
    __atomic_fetch_add(&R->free_seq, 1u, __ATOMIC_ACQ_REL);
    futex_wake_all(&R->free_seq);
    return rc;
}

}  // extern "C"
