// Live path (BASELINE config 5): all reads of one 100 ms chunk round in ONE call.
// Replaces the per-read loops of live_balancing/worker.py:26-96 (segmentation_worker: extract_adapter, MAD
// clip, segment_signal, normalize, keep the last K events) and :99-131 (classification_worker:
// model.predict(fpt, nproc=1)).  Ragged host rows in, results out; staged through page-locked buffers on the
// context's own stream: one host->device copy, the kernel chain, one device->host copy, one synchronisation.
#include "wdx_ctx.h"

#include <string.h>

#include <algorithm>

using namespace wdx;

extern "C" {

int wdx_live_tick(wdx_ctx *ctx, const float *const *rows, const int32_t *row_len, int64_t n_reads,
                  const int32_t *a_start, const int32_t *a_end, const uint8_t *ok, const wdx_seg_params *p,
                  int64_t n_refs, int32_t use_svm, double *fpt, float *dist, int32_t *call, int32_t *status,
                  double *prob, int32_t *pred, double *conf) {
    WDX_ENTER(ctx);
    if (n_reads < 0 || !p || (n_reads > 0 && (!rows || !row_len || !a_start || !a_end || !status))) {
        set_error("live_tick: bad arguments");
        return WDX_ERR_INVALID;
    }
    if (n_reads == 0) return WDX_SUCCESS;
    std::lock_guard<std::mutex> g(ctx->mu);
    DtwRefs &R = ctx->refs;
    if (R.window == 0) {
        set_error("no reference set: call wdx_set_refs first");
        return WDX_ERR_NO_REFS;
    }
    const int64_t K = p->barcode_num_events;
    if (K != R.L) {
        set_error("barcode_num_events (%lld) != reference length (%lld)", (long long)K, (long long)R.L);
        return WDX_ERR_INVALID;
    }
    if (n_refs != R.nY) {
        set_error("live_tick: the caller sized `dist` for %lld references but %lld are resident", (long long)n_refs,
                  (long long)R.nY);
        return WDX_ERR_INVALID;
    }
    if (use_svm && (!ctx->svm_set || ctx->svm.n_train != R.nY)) {
        set_error("live_tick: use_svm needs wdx_svm_set_model with a model trained on the resident reference set");
        return WDX_ERR_NO_REFS;
    }
    if (p->padding < 0) {
        set_error("padding must be >= 0");
        return WDX_ERR_INVALID;
    }
    hipStream_t s = ctx->stream;
    if ((rc = use_stream(ctx, s))) return rc;

    // ---- adapter windows (extract_adapter, sig_proc.py:382-391), packed back to back ----------------------
    // staging block: [int64 off[n+1]] [int32 zero[n]] [int32 len[n]] [uint8 ok[n] padded] [float samples...]
    int64_t total = 0, max_len = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        if (row_len[r] < 0 || (row_len[r] > 0 && !rows[r])) {
            set_error("live_tick: row %lld is null or has a negative length", (long long)r);
            return WDX_ERR_INVALID;
        }
        if (ok && !ok[r]) continue;
        int64_t st = std::max<int64_t>(0, (int64_t)a_start[r] - p->padding);
        int64_t en = std::min<int64_t>(row_len[r], (int64_t)a_end[r] + p->padding);
        if (en > st) {
            total += en - st;
            max_len = std::max(max_len, en - st);
        }
    }
    const size_t o_off = 0, o_zero = (size_t)(n_reads + 1) * 8, o_len = o_zero + (size_t)n_reads * 4,
                 o_ok = o_len + (size_t)n_reads * 4, o_sig = (o_ok + (size_t)n_reads + 15) / 16 * 16;
    const size_t in_bytes = o_sig + (size_t)total * 4;
    if ((rc = ctx->pin_in.ensure(in_bytes))) return rc;
    if ((rc = ctx->in0.ensure(in_bytes))) return rc;
    unsigned char *hin = (unsigned char *)ctx->pin_in.p;
    int64_t *h_off = (int64_t *)(hin + o_off);
    int32_t *h_zero = (int32_t *)(hin + o_zero), *h_len = (int32_t *)(hin + o_len);
    uint8_t *h_ok = hin + o_ok;
    float *h_sig = (float *)(hin + o_sig);
    int64_t pos = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        h_off[r] = pos;
        h_zero[r] = 0;
        int64_t n = 0;
        const bool good = !ok || ok[r];
        if (good) {
            int64_t st = std::max<int64_t>(0, (int64_t)a_start[r] - p->padding);
            int64_t en = std::min<int64_t>(row_len[r], (int64_t)a_end[r] + p->padding);
            if (en > st) {
                n = en - st;
                memcpy(h_sig + pos, rows[r] + st, (size_t)n * 4);
            }
        }
        // the packed row IS the window: a_start' = 0, a_end' = len makes [max(0, 0-pad), min(len, len+pad)) = the row
        h_len[r] = (int32_t)n;
        h_ok[r] = good ? 1 : 0;
        pos += n;
    }
    h_off[n_reads] = pos;

    const int k = use_svm ? ctx->svm.k : 0;
    const size_t b_status = (size_t)n_reads * 4, b_call = (size_t)n_reads * 4, b_dist = (size_t)(n_reads * R.nY) * 4,
                 b_fpt = (size_t)(n_reads * K) * 8, b_prob = (size_t)n_reads * k * 8, b_pred = (size_t)n_reads * 4,
                 b_conf = (size_t)n_reads * 8;
    // device/pinned output block (8-byte aligned pieces first)
    const size_t q_fpt = 0, q_prob = q_fpt + b_fpt, q_conf = q_prob + b_prob, q_dist = q_conf + b_conf,
                 q_status = (q_dist + b_dist + 7) / 8 * 8, q_call = q_status + b_status, q_pred = q_call + b_call,
                 out_bytes = q_pred + b_pred;
    if ((rc = ctx->out0.ensure(out_bytes))) return rc;
    if ((rc = ctx->pin_out.ensure(out_bytes))) return rc;
    if ((rc = ctx->fp_ws.ensure((size_t)fingerprint_workspace_bytes(n_reads)))) return rc;
    if ((rc = ctx->fp_big.ensure((size_t)fingerprint_big_bytes(max_len)))) return rc;
    unsigned char *din = (unsigned char *)ctx->in0.p, *dout = (unsigned char *)ctx->out0.p;
    unsigned char *hout = (unsigned char *)ctx->pin_out.p;

    StreamDrain drain(s);
    WDX_HIP_TRY(hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, s));
    {
        Timed t(ctx, WDX_K_FINGERPRINT, s);
        if ((rc = launch_fingerprint((const float *)(din + o_sig), (const int64_t *)(din + o_off), nullptr, 0,
                                     max_len, n_reads, (const int32_t *)(din + o_zero), (const int32_t *)(din + o_len),
                                     (const uint8_t *)(din + o_ok), *p, (double *)(dout + q_fpt), nullptr, nullptr,
                                     (int32_t *)(dout + q_status), s, ctx->fp_ws.p, ctx->knobs, &t.n_launches, nullptr, 0,
                                     0, nullptr, nullptr, (double *)ctx->fp_big.p)))
            return rc;
    }
    if (R.nY > 0) {
        if ((rc = dtw_dev_locked(ctx, (const double *)(dout + q_fpt), n_reads, (float *)(dout + q_dist),
                                 (int32_t *)(dout + q_call), s)))
            return rc;
        if ((rc = launch_count_calls((int32_t *)(dout + q_call), (const int32_t *)(dout + q_status), n_reads, R.nY,
                                     nullptr, s)))
            return rc;
        if (use_svm) {
            Timed t(ctx, WDX_K_SVM, s);
            if ((rc = launch_svm_predict(ctx->svm, (const float *)(dout + q_dist), n_reads, (double *)(dout + q_prob),
                                         (int32_t *)(dout + q_pred), (double *)(dout + q_conf), s, ctx->knobs)))
                return rc;
        }
    }
    // one device->host copy of what the caller asked for: [first wanted byte, last wanted byte)
    size_t lo = q_status, hi = q_call + b_call;
    if (use_svm && pred) hi = q_pred + b_pred;
    if (dist) lo = std::min(lo, q_dist);
    if (use_svm && conf) lo = std::min(lo, q_conf);
    if (use_svm && prob) lo = std::min(lo, q_prob);
    if (fpt) lo = std::min(lo, q_fpt);
    WDX_HIP_TRY(hipMemcpyAsync(hout + lo, dout + lo, hi - lo, hipMemcpyDeviceToHost, s));
    WDX_HIP_TRY(hipStreamSynchronize(s));
    drain.done();
    memcpy(status, hout + q_status, b_status);
    if (call) {
        if (R.nY > 0) memcpy(call, hout + q_call, b_call);
        else for (int64_t r = 0; r < n_reads; ++r) call[r] = -1;
    }
    if (dist && R.nY > 0) memcpy(dist, hout + q_dist, b_dist);
    if (fpt) memcpy(fpt, hout + q_fpt, b_fpt);
    if (use_svm && R.nY > 0) {
        if (prob) memcpy(prob, hout + q_prob, b_prob);
        if (pred) memcpy(pred, hout + q_pred, b_pred);
        if (conf) memcpy(conf, hout + q_conf, b_conf);
    }
    return WDX_SUCCESS;
}

}  // extern "C"
