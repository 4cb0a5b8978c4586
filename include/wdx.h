/*
 * wdx.h -- C ABI of libwdx_hip.so, the MI355X (gfx950) engine for the WarpDemuX
 *          sig_proc / parallel_distances hot path.
 *
 * Plain pointers and sizes only; no torch / numpy / HIP types.  Paths below are relative to the
 * reference tree (KleistLab/WarpDemuX v1.0.0).
 *
 * Two families of entry points:
 *   - HOST-BUFFER calls  (wdx_dtw_matrix, wdx_fingerprint_batch, ...): the caller owns every
 *     buffer (NumPy arrays); the library copies in/out and keeps nothing but the opaque context.
 *     These are what a ctypes binding inside warpdemux.parallel_distances / warpdemux.sig_proc
 *     would call.
 *   - DEVICE-RESIDENT calls (*_dev): every data pointer is a HIP device pointer on the context's
 *     device, `stream` is a hipStream_t passed as void* (NULL = default stream).  They enqueue
 *     work and return without synchronising.  Used by the fused pipeline and by bench.py.
 *
 * All functions return WDX_SUCCESS (0) or a negative WDX_ERR_* code; wdx_last_error() gives the
 * message for the calling thread.  Per-read soft failures are reported in status[] exactly like
 * the reference's ReadResult.success / fail_reason (sig_proc.py:26-62) and never fail the call.
 *
 * Threading: a context may be used from several threads (entry points serialise on it); create
 * one context per thread/stream for concurrency (live_balancing/session.py:162-169 runs thread pools):
 * every context owns a non-blocking HIP stream on which all of its host-buffer calls run, so calls
 * through different contexts overlap on the device.  The workspaces of a context are ordered by stream
 * order; when consecutive calls on one context name different streams the library waits for the earlier
 * stream first.  HIP is initialised lazily inside wdx_ctx_create, so a process may fork
 * (file_proc.py:1197-1243, ProcessPoolExecutor workers) before creating its context; a context must not
 * be used in a child forked after its creation.  The caller's current HIP device is left unchanged by
 * every entry point.  Nothing in the library reads the environment.
 */
#ifndef WDX_H
#define WDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WDX_ABI_VERSION 4

/* ---- call status ------------------------------------------------------------------------- */
#define WDX_SUCCESS 0
#define WDX_ERR_INVALID (-1)     /* bad argument (maps to ValueError in the Python shim)        */
#define WDX_ERR_NO_DEVICE (-2)   /* no usable HIP device / runtime                              */
#define WDX_ERR_HIP (-3)         /* a HIP call failed                                           */
#define WDX_ERR_UNSUPPORTED (-4) /* legal in the reference, outside this engine's limits        */
#define WDX_ERR_NO_REFS (-5)     /* a call that needs wdx_set_refs() came before it             */

/* ---- per-read status (fingerprint stage); 0 == ReadResult.success ------------------------ */
#define WDX_READ_OK 0
#define WDX_READ_FAIL_DETECT 1  /* detect_results.success False   sig_proc.py:400-407           */
#define WDX_READ_FAIL_SIGNORM 2 /* "signal normalization failed"  sig_proc.py:433-446           */
#define WDX_READ_FAIL_SEGMENT 3 /* "event segmentation failed"    sig_proc.py:537-544           */
#define WDX_READ_FAIL_SEGNORM 4 /* "segment normalization failed" sig_proc.py:546-560           */
#define WDX_READ_FAIL_UNKNOWN 5 /* exception -> "unknown"         file_proc.py:209-224          */
#define WDX_READ_FAIL_CONSENSUS 6 /* "consensus query outlier"    sig_proc.py:497-512 (refinement) */

/* normalisation selectors (sig_proc.py:114-136) */
#define WDX_NORM_NONE 0
#define WDX_NORM_MEAN 1
#define WDX_NORM_MEDIAN 2

/* The hot-path knobs of SigProcConfig (config/sig_proc.py:16-70 + ADAPTed core.*), by value. */
typedef struct wdx_seg_params {
    int32_t padding;            /* sig_extract.padding                                          */
    int32_t sig_norm;           /* sig_extract.normalization   (WDX_NORM_*)                     */
    float outlier_thresh;       /* core.sig_norm_outlier_thresh                                 */
    int32_t min_obs_per_base;   /* segmentation.min_obs_per_base                                */
    int32_t running_stat_width; /* segmentation.running_stat_width                              */
    int32_t num_events;         /* segmentation.num_events                                      */
    int32_t accept_less_cpts;   /* segmentation.accept_less_cpts                                */
    int32_t seg_norm;           /* segmentation.normalization  (WDX_NORM_*)                     */
    int32_t barcode_num_events; /* segmentation.barcode_num_events (int form)                   */
    /* How the clip bounds `med -/+ thresh*mad` (sig_proc.py:426-431) are evaluated -- the one place on the
     * path where the reference's result depends on its NumPy: 0 = in float32 from outlier_thresh (NumPy >= 2
     * promotion: float32 scalar x Python float stays float32); 1 = in float64 from outlier_thresh_f64 and
     * rounded to float32 once (NumPy 1.x value-based promotion -- the reference pins numpy 1.26.4,
     * environment.yml -- or an np.float64 threshold under NumPy 2).  The Python shim picks the rule of the
     * NumPy it runs under, so the drop-in returns what the reference would have returned in that process. */
    int32_t clip_bounds_f64;
    double outlier_thresh_f64;  /* core.sig_norm_outlier_thresh as a double (used when clip_bounds_f64 != 0) */
} wdx_seg_params;

typedef struct wdx_ctx wdx_ctx;

/* ---- context ----------------------------------------------------------------------------- */
int wdx_abi_version(void);
/* Message of the last failing call on this thread ("" if none). Never NULL. */
const char *wdx_last_error(void);
/* Number of visible HIP devices, or a negative WDX_ERR_*. Does not create a context. */
int wdx_device_count(void);
/* Create a context on HIP device `device`.  First HIP use in the process happens here. */
int wdx_ctx_create(int device, wdx_ctx **out);
void wdx_ctx_destroy(wdx_ctx *ctx);
/* Block until all work enqueued on `stream` has finished.  stream == NULL names the legacy NULL stream,
 * exactly as in the *_dev entry points, AND the context's own stream (wdx_ctx_stream) is waited for too --
 * so `wdx_demux_dev(..., NULL); wdx_ctx_synchronize(ctx, NULL);` is complete when it returns (ABI 2 briefly
 * waited for the context stream only: fixed in ABI 3). */
int wdx_ctx_synchronize(wdx_ctx *ctx, void *stream);
/* The context's own stream (hipStream_t as void*): the one its host-buffer calls run on. */
int wdx_ctx_stream(wdx_ctx *ctx, void **stream);

/* Diagnostic switches (all 0 by default = the product path); used by tests and profiling tools only. */
#define WDX_OPT_EXACT_PATH 1        /* fingerprint every read on the exact general kernel               */
#define WDX_OPT_NO_WAVEFRONT_DTW 2  /* never dispatch the anti-diagonal DTW kernel                       */
#define WDX_OPT_NO_SHORT_DTW 3      /* never dispatch the unrolled 25-point DTW kernel                   */
#define WDX_OPT_SVM_SCALAR 4        /* scalar SVM tail kernel instead of the matrix-core one             */
#define WDX_OPT_DEBUG_OCCUPANCY 5   /* print the fast fingerprint kernel's workgroups per CU to stderr   */
#define WDX_OPT_FAST_PEAK_CAP 6     /* peak-list capacity of the fast fingerprint kernel (0 = built-in)  */
#define WDX_OPT_FAST_EXACT_SCORES 7 /* fast fingerprint kernel: exact t-scores, no approximate keys       */
#define WDX_OPT_FAST_MAIN_CAP 8     /* main fast instantiation: 5120 or 6144 samples (0 = chosen by batch)  */
#define WDX_OPT_FAST_CHAIN_MIN_READS 9 /* smallest batch that takes the approximate-keys launch chain (0 = 2048) */
#define WDX_OPT_EXACT_NO_PEAK_LIST 10  /* exact kernel: suppression / top-E over positions, never over the peak list */
#define WDX_OPT_MAX_LAUNCH_SLICE 11    /* fingerprint chain: at most this many workgroups per launch slice (0 = built-in) */
#define WDX_OPT_NO_PEAK_FILTER 12      /* fast fingerprint kernels: no threshold filter of the peak list (every local maximum) */
#define WDX_OPT_NO_WAVE_CLIP_LONG 13    /* windows beyond 6144 samples: clip bounds by the workgroup kernel alone (A/B, tests) */
#define WDX_OPT_NO_CLIP_REUSE 14        /* exact kernel behind the launch chain: recompute the clip bounds (A/B, tests) */
#define WDX_OPT_NO_SPLIT_TAIL 15        /* main fast fingerprint kernel in one piece: no tile kernel + tail kernel split (A/B, tests) */
#define WDX_OPT_DTW_UNFUSED 16          /* DTW cells: 0 (default) one v_fma_f64 per cell, pairs whose float32 could differ from the
                                         * reference's are run again on its six operations (wdx_dtw.hip: dtw_unsettled) | 1 the six
                                         * operations only (A/B) | 2 fused and every pair run again (tests) | 3 fused, never run again
                                         * (diagnostic: NOT the reference's results) */
int wdx_ctx_set_option(wdx_ctx *ctx, int32_t option, int64_t value);

/* ---- seam 1: batched DTW  (replaces parallel_distances.py:48-67 `distance_matrix_to`,
 *      i.e. dtaidistance.dtw.distance_matrix(vstack[X,Y], block=((0,nX),(nX,nX+nY)),
 *      window, penalty, use_c=True)[:nX, nX:].astype(float32) ) ------------------------------ */

/* X: (nX,L) float64 row-major host; Y: (nY,L) float64 row-major host; out: (nX,nY) float32 host.
 * window <= 0 means unbanded (reference: None/0); penalty is the un-squared dtaidistance penalty
 * (0 = none).  argmin (nullable): int32[nX] = np.argmin(out, axis=1).  Both nX and nY may be 0. */
int wdx_dtw_matrix(wdx_ctx *ctx, const double *X, int64_t nX, const double *Y, int64_t nY,
                   int64_t L, int32_t window, double penalty, float *out, int32_t *argmin);

/* Upload the reference set once (model._X, models/dtw_base.py:14-17) and keep it resident.
 * Y is a HOST pointer; it is re-uploaded only if its content/params differ from the cached set. */
int wdx_set_refs(wdx_ctx *ctx, const double *Y, int64_t nY, int64_t L, int32_t window,
                 double penalty);
/* Counter that changes whenever the resident reference set changes (samples, window or penalty) -- by
 * wdx_set_refs or by wdx_dtw_matrix, which installs its Y.  A caller that keeps "its" references resident
 * across calls (the live tick loop, worker.py:26-131) compares it with the value it saw after its own
 * wdx_set_refs to learn that another user of the context has replaced them. */
int wdx_refs_generation(wdx_ctx *ctx, int64_t *generation);

/* Device-resident DTW against the resident reference set.
 * dX: (nX,L) float64 row-major DEVICE; d_out: (nX,nY) float32 DEVICE;
 * d_argmin (nullable): int32[nX] DEVICE. */
int wdx_dtw_matrix_dev(wdx_ctx *ctx, const double *dX, int64_t nX, float *d_out,
                       int32_t *d_argmin, void *stream);

/* ---- seam 2: batched fingerprinting (replaces the per-read loop file_proc.py:418-428 over
 *      sig_proc.py:394-605 `detect_results_to_fpt`, non-refinement branch) ------------------- */

/* sig: (n_reads, stride) float32 host minibatch, NaN tail (file_proc.py:244-260);
 * a_start/a_end: DetectResults.adapter_start/end; ok (nullable): DetectResults.success.
 * Outputs (host): fpt (n_reads,K) float64, dwell (n_reads,K) int64, stats (n_reads,6) float64 =
 * {adapter_dt_med, adapter_dt_mad, adapter_event_mean, adapter_event_std, adapter_event_med,
 * adapter_event_mad}, status int32[n_reads] (WDX_READ_*).  Rows of failed reads hold NaN / 0.
 * K = p->barcode_num_events.  The input rows are NOT clipped in place (the reference's
 * in-place clip, sig_proc.py:426-431, is never read again: file_proc.py:430). */
int wdx_fingerprint_batch(wdx_ctx *ctx, const float *sig, int64_t n_reads, int64_t stride,
                          const int32_t *a_start, const int32_t *a_end, const uint8_t *ok,
                          const wdx_seg_params *p, double *fpt, int64_t *dwell, double *stats,
                          int32_t *status);

/* Device-resident form.  Read r occupies d_sig[row_off[r] .. row_off[r] + row_len[r]) where
 *   d_row_off == NULL  -> row_off[r] = r*stride        (minibatch layout)
 *   d_row_len == NULL  -> row_len[r] = d_row_off ? d_row_off[r+1]-d_row_off[r] : stride
 * (so a packed batch passes int64 offsets[n_reads+1] and NULL lengths).  max_len bounds the
 * adapter window of every read (it sizes the LDS carve-up); windows longer than max_len or than
 * WDX_MAX_ADAPTER_SAMPLES are reported WDX_READ_FAIL_UNKNOWN.  (The largest window the reference admits is
 * max_obs_trace + 2*padding = 15 200 samples, DEPRECATED/config_files/rna002_70bps@v0.4.4.toml:2; up to 11 200
 * samples a read's score curve lives in LDS, beyond that in a context-owned HBM block of 32 MiB that is
 * allocated on first need -- a synchronising hipMalloc on that one call.)
 * Any output pointer except d_status may be NULL. */
int wdx_fingerprint_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off,
                        const int32_t *d_row_len, int64_t stride, int64_t max_len,
                        int64_t n_reads, const int32_t *d_a_start, const int32_t *d_a_end,
                        const uint8_t *d_ok, const wdx_seg_params *p, double *d_fpt,
                        int64_t *d_dwell, double *d_stats, int32_t *d_status, void *stream);

#define WDX_MAX_ADAPTER_SAMPLES 16384

/* ---- N3: consensus-guided barcode refinement (tRNA models) -- detect_results_to_fpt with
 *      segmentation.consensus_refinement = True (sig_proc.py:257-378, 452-521): segment the adapter, find the
 *      constant adapter part by subsequence DTW of a consensus signal against the normalised event means
 *      (dtaidistance warping_paths_fast + SubsequenceAlignment.best_match in the reference), re-segment the score
 *      curve behind it into barcode events, normalise them with the ADAPTER's mean/std (normalize_wrt). */
typedef struct wdx_refine_params {
    const double *query;          /* consensus signal (warpdemux/_consensus.py ALL[consensus_model]); HOST pointer   */
    int32_t n_query;              /* 1..96                                                                         */
    int32_t subseq_norm;          /* segmentation.consensus_subseq_match_normalization (WDX_NORM_*)                 */
    double penalty;               /* segmentation.consensus_subseq_match_penalty (un-squared)                      */
    int32_t psi[4];               /* segmentation.consensus_subseq_match_psi: relaxation at the begin / end of the
                                     query and the begin / end of the series (the two end values do not enter the
                                     matching function the reference reads)                                        */
    int32_t ub_start, lb_end, ub_end; /* consensus_subseq_match_ub_start / lb_end / ub_end (outlier filter)         */
    int32_t barcode_segm_events;  /* barcode_num_events[0]: events detected in the barcode tail                     */
    int32_t barcode_keep_events;  /* barcode_num_events[1]: events kept = K of fpt / dwell                          */
} wdx_refine_params;
/* As wdx_fingerprint_batch; K = rp->barcode_keep_events (p->barcode_num_events is ignored); stats are the ADAPTER's;
 * refine_idx (n_reads, 3) int32 = {seg_cons_query_start, seg_cons_query_end, sig_barcode_start} (-1 when the read
 * failed earlier).  Status WDX_READ_FAIL_CONSENSUS still reports stats and refine_idx, like the reference's
 * ReadResult.  Limits: num_events <= 127, refinement_optimal_cpts (ruptures KernelCPD) -> not offered. */
int wdx_fingerprint_refine_batch(wdx_ctx *ctx, const float *sig, int64_t n_reads, int64_t stride,
                                 const int32_t *a_start, const int32_t *a_end, const uint8_t *ok,
                                 const wdx_seg_params *p, const wdx_refine_params *rp, double *fpt, int64_t *dwell,
                                 double *stats, int32_t *refine_idx, int32_t *status);

/* Device-resident form of the refinement branch: inputs and outputs as in wdx_fingerprint_dev (d_fpt / d_dwell have
 * K = rp->barcode_keep_events columns), d_refine_idx (n_reads, 3) int32 on the device; rp and its query are HOST
 * memory (copied before the call returns).  Enqueued on `stream`; no synchronisation. */
int wdx_fingerprint_refine_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off,
                               const int32_t *d_row_len, int64_t stride, int64_t max_len, int64_t n_reads,
                               const int32_t *d_a_start, const int32_t *d_a_end, const uint8_t *d_ok,
                               const wdx_seg_params *p, const wdx_refine_params *rp, double *d_fpt,
                               int64_t *d_dwell, double *d_stats, int32_t *d_refine_idx, int32_t *d_status,
                               void *stream);

/* ---- fused path: raw adapter rows -> fingerprint -> DTW to the resident refs -> call ------ */

/* As wdx_fingerprint_dev, then DTW of every successful read against the resident refs.
 * d_dist: (n_reads,nY) float32; d_call: int32[n_reads] = argmin column or -1 for failed reads;
 * d_counts (nullable): int64[nY+1], INCREMENTED by the per-column call histogram, slot nY =
 * failed reads.  d_fpt/d_dwell/d_stats are optional as above.  d_work: DEVICE scratch of at
 * least wdx_demux_workspace_bytes(n_reads, K) bytes (fingerprints, the chain's hand-over lists and clip records -- 40 bytes per
 * read -- and the split main kernel's peak lists for one launch slice: 4 624 bytes per read of min(n_reads, 524 288)). */
int64_t wdx_demux_workspace_bytes(int64_t n_reads, int32_t barcode_num_events);
int wdx_demux_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off,
                  const int32_t *d_row_len, int64_t stride, int64_t max_len, int64_t n_reads,
                  const int32_t *d_a_start, const int32_t *d_a_end, const uint8_t *d_ok,
                  const wdx_seg_params *p, double *d_fpt, int64_t *d_dwell, double *d_stats,
                  int32_t *d_status, float *d_dist, int32_t *d_call, int64_t *d_counts,
                  void *d_work, void *stream);

/* Host-buffer form of the fused path (one call per minibatch or per live tick, one synchronisation):
 * fingerprint the (n_reads, stride) float32 minibatch, DTW every successful read against the resident
 * reference set (wdx_set_refs; K = p->barcode_num_events must equal its length), nearest-reference call.
 * n_refs = the number of references the caller sized `dist` for; it must equal the resident nY
 * (WDX_ERR_INVALID otherwise: another user of the context may have replaced the set).
 * Host outputs: status int32[n_reads]; call int32[n_reads] (argmin column, -1 for failed reads);
 * dist (n_reads, n_refs) float32 (nullable; NaN rows for failed reads); fpt (n_reads, K) float64 (nullable). */
int wdx_demux_batch(wdx_ctx *ctx, const float *sig, int64_t n_reads, int64_t stride,
                    const int32_t *a_start, const int32_t *a_end, const uint8_t *ok,
                    const wdx_seg_params *p, int64_t n_refs, double *fpt, float *dist, int32_t *call,
                    int32_t *status);

/* Pipelined form of wdx_demux_batch for the reference's worker loop (file_proc.py:380-454: fill minibatch k+1
 * while minibatch k is processed; 1197-1243: several such workers share one GPU).  A context has WDX_MAX_SLOTS slots (the
 * worker loop uses two; a feeder that serves many producers keeps more minibatches in flight), each with its own stream
 * and device workspaces:
 *   wdx_demux_submit(ctx, slot, ...)  enqueues copy-in, fingerprint, DTW, call and copy-out of one minibatch on the
 *                                     slot's stream and returns; WDX_ERR_INVALID if the slot still holds a batch.
 *                                     The input arrays must stay untouched until the matching wait returns (rows in
 *                                     page-locked memory -- wdx_host_alloc -- are copied by DMA at the bus rate and the
 *                                     call returns at once; pageable rows are staged by the runtime first).
 *   wdx_demux_wait(ctx, slot, ...)    blocks until that minibatch is done and hands the results over (same outputs
 *                                     and meanings as wdx_demux_batch; fpt / dist only if requested at submit).
 * Submitting slot 1 while slot 0 is in flight overlaps its host->device copy with slot 0's kernels.  Results are
 * bit-identical to wdx_demux_batch.  wdx_set_refs waits for every slot before it changes the resident set. */
#define WDX_MAX_SLOTS 8
int wdx_demux_submit(wdx_ctx *ctx, int32_t slot, const float *sig, int64_t n_reads, int64_t stride,
                     const int32_t *a_start, const int32_t *a_end, const uint8_t *ok, const wdx_seg_params *p,
                     int64_t n_refs, int32_t want_fpt, int32_t want_dist);
int wdx_demux_wait(wdx_ctx *ctx, int32_t slot, double *fpt, float *dist, int32_t *call, int32_t *status);
/* The same pipeline with everything the reference's worker needs from a minibatch (file_proc.py:380-454: the
 * ReadResults -- fingerprint, dwell times, six statistics, sig_proc.py:562-605 -- AND model.predict of the stacked
 * fingerprints, models/dtw_svm.py:54-98) produced by ONE pass over the rows:
 *   wdx_demux_submit_ex(ctx, slot, in, p, n_refs, want)   `want` = WDX_WANT_* bits; status and call always come back.
 *                                                         in->row_off != NULL: the rows are PACKED -- row r holds only the
 *                                                         samples the kernels read, sig[row_off[r] .. + row_len[r]), and
 *                                                         a_start / a_end are relative to it (what a worker that copies
 *                                                         [a_start - padding, a_end + padding) of each read hands over:
 *                                                         half the bytes of the NaN-padded rows); row_off[r] % 4 == 0.
 *                                                         WDX_WANT_SVM needs wdx_svm_set_model (a model trained on the
 *                                                         resident references); reads whose fingerprint failed get
 *                                                         pred -1 and NaN prob / conf, like wdx_demux_svm_dev.
 *   wdx_demux_wait_ex(ctx, slot, out)                     out->X may be NULL for anything; an output that was not asked
 *                                                         for at submit is WDX_ERR_INVALID and leaves the slot busy.
 * Bit-identical to wdx_fingerprint_batch / wdx_demux_batch / wdx_dtw_svm_predict on the same rows. */
#define WDX_WANT_FPT 0x01u   /* fpt   (n, K) float64                                      */
#define WDX_WANT_DIST 0x02u  /* dist  (n, n_refs) float32                                 */
#define WDX_WANT_DWELL 0x04u /* dwell (n, K) int64                                        */
#define WDX_WANT_STATS 0x08u /* stats (n, 6) float64 (order: wdx_fingerprint_batch)       */
#define WDX_WANT_SVM 0x10u   /* prob (n, k) float64, pred int32[n], conf float64[n]       */
typedef struct wdx_minibatch_in {
    const float *sig;          /* (n_reads, stride) rows, or the packed rows when row_off != NULL             */
    int64_t n_reads, stride;   /* stride is ignored for packed rows                                          */
    const int64_t *row_off;    /* packed: int64[n_reads + 1], multiples of 4; NULL = minibatch layout        */
    const int32_t *row_len;    /* packed: samples of row r (<= row_off[r+1] - row_off[r])                    */
    const int32_t *a_start, *a_end;
    const uint8_t *ok;         /* nullable                                                                    */
} wdx_minibatch_in;
typedef struct wdx_minibatch_out {
    int32_t *status, *call;
    float *dist;
    double *fpt;
    int64_t *dwell;
    double *stats, *prob;
    int32_t *pred;
    double *conf;
} wdx_minibatch_out;
int wdx_demux_submit_ex(wdx_ctx *ctx, int32_t slot, const wdx_minibatch_in *in, const wdx_seg_params *p, int64_t n_refs,
                        uint32_t want);
int wdx_demux_wait_ex(wdx_ctx *ctx, int32_t slot, const wdx_minibatch_out *out);
/* Page-locked host memory for minibatch buffers the caller fills (what file_proc.py:244-260 allocates with
 * np.full): the GPU reads it directly.  Needs no context; free with wdx_host_free. */
int wdx_host_alloc(size_t bytes, void **out);
/* the same with `device` made current for the allocation (a multi-GPU worker whose context lives on device N: the
 * plain call initialises HIP on the process's current device) */
int wdx_host_alloc_on(int device, size_t bytes, void **out);
int wdx_host_free(void *p);
/* Page-lock memory the caller already owns -- e.g. a shared-memory ring that producer PROCESSES fill while one feeder
 * process owns the context and submits (tools/host_workers.py --mode feeder) -- so that it is read like wdx_host_alloc
 * memory.  Undo with wdx_host_unregister before the memory is unmapped. */
int wdx_host_register(void *p, size_t bytes);
int wdx_host_unregister(void *p);

/* ---- many worker processes, ONE GPU-facing process (replaces the reference's per-worker GPU use in its `-j 8..16`
 *      forked workers, file_proc.py:1197-1243, 380-454).  Sixteen HIP processes on one device run at 40 % of the rate of
 *      four; one process that owns the context and keeps up to WDX_MAX_SLOTS minibatches in flight for everybody does not
 *      have that problem.  The ring lives in shared memory the caller maps in every process (parent: create + init BEFORE
 *      the fork; Python: warpdemux_amd.feeder.Feeder):
 *        wdx_feeder_ring_bytes / _init   size and lay out the ring: n_slots (<= WDX_FEEDER_MAX_RING_SLOTS) minibatches of at
 *                                        most max_reads x max_stride float32 samples, distances to n_refs references,
 *                                        n_events > 0: room for fingerprints / dwell times / statistics (K = n_events =
 *                                        p->barcode_num_events), n_classes > 0: room for the DTW_SVM outputs.  `p` = the
 *                                        parameters every minibatch is fingerprinted with (kept in the ring: the workers
 *                                        pack their rows with its `padding`).
 *                                        (A worker holds its ring slot while it copies its windows in and the results out;
 *                                        the feeder keeps at most WDX_MAX_SLOTS of the READY ones in flight on the device.)
 *        wdx_feeder_serve(ctx, ring)     the GPU-facing process: page-locks the ring and serves it until wdx_feeder_stop --
 *                                        every READY slot goes through wdx_demux_submit_ex, the oldest in flight through
 *                                        wdx_demux_wait_ex; the context's resident references (wdx_set_refs) classify, its
 *                                        resident model (wdx_svm_set_model) serves WDX_WANT_SVM and wdx_feeder_predict
 *        wdx_feeder_run(ring, job)       a worker: one minibatch, `job->want` = WDX_WANT_* bits -- what the reference's
 *                                        worker needs from it (file_proc.py:380-454): status + fpt + dwell + stats (the
 *                                        ReadResults, = wdx_fingerprint_batch), call + dist (= wdx_demux_batch), prob +
 *                                        pred + conf (= model.predict of the stacked fingerprints, models/dtw_svm.py:54-98;
 *                                        failed reads: pred -1, NaN) -- from ONE pass, bit-identical to those calls.  No
 *                                        context and NO HIP call: only samples [a_start - padding, a_end + padding) of each
 *                                        row are copied into a free slot (packed rows), the worker sleeps on the slot
 *                                        (futex) until the results are there.  WDX_ERR_NO_DEVICE when the feeder has
 *                                        stopped or died (a dead feeder is noticed even while it is an unreaped zombie)
 *        wdx_feeder_demux(ring, ...)     wdx_demux_batch's arguments through wdx_feeder_run (status, call, dist)
 *        wdx_feeder_predict(ring, X, ..) DTW_SVM.predict on (n, n_events) float64 fingerprints the worker holds
 *        wdx_feeder_stop(ring)           ends wdx_feeder_serve: minibatches in flight are finished and handed over, READY
 *                                        ones that were never submitted are answered WDX_ERR_NO_DEVICE, new claims are
 *                                        refused
 *      A worker that dies while it holds a slot does not leak it: the slot's owner pid is part of its state word, and the
 *      serve loop (and any claimant that finds the ring full) gives slots of dead owners back to the ring. */
#define WDX_FEEDER_MAX_RING_SLOTS 32
typedef struct wdx_feeder_geometry {
    int32_t n_slots;
    int32_t n_events;    /* K of fpt / dwell (0: no room for WDX_WANT_FPT / _DWELL / _STATS)      */
    int32_t n_classes;   /* k of prob (0: no room for WDX_WANT_SVM / wdx_feeder_predict), <= 16  */
    int32_t pad_;
    int64_t max_reads, max_stride, n_refs;
} wdx_feeder_geometry;
typedef struct wdx_feeder_job {
    const float *sig;          /* (n_reads, stride) float32 rows, NaN tail (file_proc.py:244-260)               */
    int64_t n_reads, stride;
    const int32_t *a_start, *a_end;
    const uint8_t *ok;         /* nullable                                                                       */
    uint32_t want, pad_;       /* WDX_WANT_* bits                                                                */
    /* outputs: caller-owned host arrays; status is required, call nullable, the others where their bit is set   */
    int32_t *status, *call;
    float *dist;               /* (n_reads, n_refs)                                                              */
    double *fpt;               /* (n_reads, n_events)                                                            */
    int64_t *dwell;            /* (n_reads, n_events)                                                            */
    double *stats;             /* (n_reads, 6)                                                                   */
    double *prob;              /* (n_reads, n_classes)                                                           */
    int32_t *pred;
    double *conf;
} wdx_feeder_job;
size_t wdx_feeder_ring_bytes(const wdx_feeder_geometry *g);
int wdx_feeder_ring_init(void *mem, size_t bytes, const wdx_feeder_geometry *g, const wdx_seg_params *p);
int wdx_feeder_serve(wdx_ctx *ctx, void *ring);
int wdx_feeder_run(void *ring, const wdx_feeder_job *job);
int wdx_feeder_demux(void *ring, const float *sig, int64_t n_reads, int64_t stride, const int32_t *a_start,
                     const int32_t *a_end, const uint8_t *ok, int64_t n_refs, float *dist, int32_t *call, int32_t *status);
int wdx_feeder_predict(void *ring, const double *X, int64_t n, double *prob, int32_t *pred, double *conf);
int wdx_feeder_stop(void *ring);
int wdx_feeder_served(void *ring, int64_t *minibatches);   /* minibatches handed back so far */
/* served minibatches, slots taken back from dead workers, slots FREE right now (outputs nullable) */
int wdx_feeder_stats(void *ring, int64_t *served, int64_t *reclaimed, int32_t *free_slots);
/* Test hooks (no GPU): 1 = claim a slot for the calling process and leave it FILLING (returns its index), 2 = pose as the
 * serving feeder without serving.  Never on the product path. */
int wdx_feeder_selftest(void *ring, int32_t what);
int wdx_feeder_alive(void *ring);   /* 1 while a feeder process serves the ring, 0 once it has stopped or died (< 0: error) */

/* Live path (BASELINE config 5; N4): every read of one 100 ms chunk round in one call -- the batched form of
 * live_balancing/worker.py:26-96 (segmentation_worker) + :99-131 (classification_worker).  rows[r] points at
 * read r's float32 samples (row_len[r] of them; ragged, caller-owned, only the adapter window
 * [max(0, a_start-padding), min(row_len, a_end+padding)) is read -- the live caller passes a_start = 0 and
 * a_end = polya_start, worker.py:39-44).  The windows are packed into a page-locked staging block, copied
 * once, run through fingerprint -> DTW against the resident references [-> SVM tail when use_svm != 0 and a
 * model trained on those references is resident] on the context's own stream, and the requested outputs
 * come back in one copy; one synchronisation per call.  Outputs are HOST pointers: status int32[n] is
 * required; call int32[n], dist (n, n_refs) float32, fpt (n, K) float64, prob (n, k) float64,
 * pred int32[n] (barcode label or -1 = outlier, worker.py:125), conf float64[n] are nullable. */
int wdx_live_tick(wdx_ctx *ctx, const float *const *rows, const int32_t *row_len, int64_t n_reads,
                  const int32_t *a_start, const int32_t *a_end, const uint8_t *ok, const wdx_seg_params *p,
                  int64_t n_refs, int32_t use_svm, double *fpt, float *dist, int32_t *call, int32_t *status,
                  double *prob, int32_t *pred, double *conf);

/* ---- N1: classifier tail of DTW_SVM.predict (models/dtw_svm.py:90-93 + models/utils.py:45-61):
 *      K = exp(-gamma * d^pwr_dist) -> SVC.predict_proba(K) (libsvm, precomputed kernel) -> argmax,
 *      label map, top1-top2 margin, per-class thresholds.  Arrays are HOST pointers, copied at set time. */
typedef struct wdx_svm_model {
    int32_t n_classes;         /* k = len(svc.classes_) (barcodes + noise class), 2..16              */
    int32_t n_sv;              /* total support vectors                                              */
    int32_t n_train;           /* columns of the distance matrix = len(model._X)                     */
    int32_t pwr_dist;          /* DTW_SVM.pwr_dist                                                   */
    double gamma;              /* DTW_SVM.gamma                                                      */
    const int32_t *n_support;  /* [k]    svc._n_support                                              */
    const int32_t *support;    /* [n_sv] svc.support_ (column index of every support vector)         */
    const double *dual_coef;   /* [(k-1) x n_sv] svc._dual_coef_ (libsvm sign convention)            */
    const double *rho;         /* [k(k-1)/2]  = -svc._intercept_                                     */
    const double *probA;       /* [k(k-1)/2]  svc._probA                                             */
    const double *probB;       /* [k(k-1)/2]  svc._probB                                             */
    const int32_t *label_map;  /* [k] class index -> barcode label (model.label_mapper); nullable     */
    const double *thresholds;  /* [k] model.thresholds; nullable = no thresholding                    */
} wdx_svm_model;
int wdx_svm_set_model(wdx_ctx *ctx, const wdx_svm_model *m);
/* d_dist: (n, n_train) float32 DEVICE distances (wdx_dtw_matrix_dev output); outputs DEVICE, nullable:
 * d_prob (n,k) float64 = y_prob, d_pred int32[n] = predicted barcode or -1, d_conf float64[n] = margin. */
int wdx_svm_predict_dev(wdx_ctx *ctx, const float *d_dist, int64_t n, double *d_prob, int32_t *d_pred,
                        double *d_conf, void *stream);

/* The shipped models' whole path in ONE device-resident call (file_proc.py:418-450 + models/dtw_svm.py:54-98 +
 * models/utils.py:19-61): raw adapter rows -> fingerprint (K = the reference length) -> DTW against the resident
 * training set -> exp(-gamma d^p) -> one-vs-one decision values -> Platt sigmoids + coupling -> process_probs.
 * Inputs as wdx_demux_dev; needs wdx_set_refs and wdx_svm_set_model (a model trained on the resident set).
 * The distance matrix is produced and consumed in row blocks of `block_rows` reads (0 = chosen so that a block is
 * <= 96 MiB, i.e. stays in the 256 MB memory-side cache between the DTW kernel that writes it and the SVM tail that
 * reads it) in a context-owned buffer; it is only written out in full when d_dist (n, nY) is given.
 * Outputs DEVICE: d_status int32[n]; d_prob (n,k) float64, d_pred int32[n] (barcode label or -1), d_conf float64[n],
 * each nullable; reads whose fingerprint failed get pred -1 and NaN probabilities / margin.  d_fpt (n,K) nullable.
 * d_work: wdx_demux_workspace_bytes(n_reads, K) bytes.  Enqueued on `stream`; no synchronisation. */
int wdx_demux_svm_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off, const int32_t *d_row_len, int64_t stride,
                      int64_t max_len, int64_t n_reads, const int32_t *d_a_start, const int32_t *d_a_end,
                      const uint8_t *d_ok, const wdx_seg_params *p, double *d_fpt, int32_t *d_status, float *d_dist,
                      double *d_prob, int32_t *d_pred, double *d_conf, void *d_work, int64_t block_rows, void *stream);
/* DTW_SVM.predict on host buffers: X (n, L) float64 fingerprints -> DTW against the resident reference
 * set (wdx_set_refs with model._X, window, penalty) -> SVM tail.  Outputs host, nullable. */
int wdx_dtw_svm_predict(wdx_ctx *ctx, const double *X, int64_t n, double *prob, int32_t *pred, double *conf);

/* ---- multi-GPU: the only exchange on the path ---------------------------------------
 * Reads shard over one process per GPU with no data-path collective; after the last batch the per-barcode
 * call histogram -- the engine's form of the reference's shared run counters `ridx_dict`
 * (file_proc.py:1059-1066) -- is summed over ranks by ONE RCCL all-reduce (xGMI within a node).
 * librccl is dlopen'ed on first use (the copy already loaded in the process, e.g. PyTorch's, is
 * preferred), so a single-GPU user never needs it. */
#define WDX_COMM_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */
/* WDX_SUCCESS when librccl could be bound in this process, WDX_ERR_NO_DEVICE (and the reason in
 * wdx_last_error) when it could not.  Local and cheap: every rank asks BEFORE anyone enters the collective
 * wdx_comm_init, so that all ranks take the same road.  Needs no context. */
int wdx_comm_available(void);
/* Rank 0 creates the rendezvous id and hands the 128 bytes to the other ranks by any means (the
 * launcher's store, a file, MPI ...).  Needs no context. */
int wdx_comm_unique_id(void *id_out /* WDX_COMM_ID_BYTES */);
/* Collective over all `world` ranks: bind the context to rank `rank` of the communicator `id`. */
int wdx_comm_init(wdx_ctx *ctx, const void *id, int32_t rank, int32_t world);
int wdx_comm_destroy(wdx_ctx *ctx);
/* What the context is bound to: rank/world as given to wdx_comm_init (0 / 1 without a communicator) and
 * rccl_count = the rank count RCCL itself reports for the communicator (ncclCommCount; 0 without a
 * communicator, -1 if this librccl lacks the query).  Outputs nullable. */
int wdx_comm_info(wdx_ctx *ctx, int32_t *rank, int32_t *world, int32_t *rccl_count);
/* In-place SUM all-reduce of d_counts int64[n] (DEVICE pointer, e.g. wdx_demux_dev's d_counts) over the
 * communicator, enqueued on `stream`; no synchronisation.  Without a communicator (single process) it is
 * a no-op that returns WDX_SUCCESS. */
int wdx_reduce_counts(wdx_ctx *ctx, int64_t *d_counts, int32_t n, void *stream);
/* Host-buffer form: counts int64[n] on the host, reduced in place; synchronises. */
int wdx_reduce_counts_host(wdx_ctx *ctx, int64_t *counts, int32_t n);

/* ---- measurement helpers ------------------------------------------------------------------ */

/* Kernel ids for wdx_kernel_time */
#define WDX_K_FINGERPRINT 0
#define WDX_K_DTW 1
#define WDX_K_TRANSPOSE 2
#define WDX_K_COUNT 3
#define WDX_K_SVM 4
#define WDX_K_REDUCE 5
#define WDX_K_FINGERPRINT_MAIN 6 /* the main fast fingerprint kernel alone (WDX_K_FINGERPRINT = the whole chain) */
#define WDX_K_FINGERPRINT_CLIP 7 /* clip_bounds_kernel alone (median / MAD / clip bounds ahead of the main kernel) */
#define WDX_K_FINGERPRINT_TAIL 8 /* fingerprint_split_tail_kernel alone (the split main kernel's second half; its time is part of
                                    WDX_K_FINGERPRINT_MAIN, which brackets the tile-kernel / tail-kernel launch pairs) */
/* When enabled, every kernel launch through this context is bracketed by hipEvents on its
 * stream; wdx_kernel_time() synchronises them and returns accumulated ms and launch count. */
int wdx_kernel_timing(wdx_ctx *ctx, int enable);
int wdx_kernel_time(wdx_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
int wdx_kernel_time_reset(wdx_ctx *ctx);
/* Diagnostic build of the fingerprint kernel with s_memtime stamps between its phases:
 * d_prof receives 32 int64 per read for the first prof_reads reads (slots 0..9 = shader-clock
 * stamps at the phase boundaries P0..P7, 10 = suppression iterations, 11 = adapter samples,
 * 12 = score positions / peaks).  fast_path selects the 256-thread fast kernel (+ slow-path list;
 * slot 15 of read 0 then holds the number of reads it declined) or the one-kernel exact path.
 * fast_path == 2: the split pair of the RNA004 main kernel (tile kernel: slots 0, 3, 4 = start, samples clipped in LDS, tile pass
 * done, 5 = export done; tail kernel: 6 entries loaded, 7 boundaries written, 8 event means + mean / sd, 9 end; 12 = exported peaks).
 * stop_phase k > 0 makes the fast kernel return after phase k (ablation timing; results are garbage).
 * Outputs other than d_status are discarded.  Never on the product path. */
int wdx_fingerprint_profile_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off,
                                int64_t stride, int64_t max_len, int64_t n_reads,
                                const int32_t *d_a_start, const int32_t *d_a_end,
                                const wdx_seg_params *p, int32_t *d_status, long long *d_prof,
                                int64_t prof_reads, int32_t fast_path, int32_t stop_phase, void *stream);

/* Self-test: d_fast[i] = the fast fingerprint kernel's unscaled quotient/sqrt sequence for dm[i] / sqrt(vs[i]),
 * d_ref[i] = the compiler's general float64 expansions of the same expression.  The two must agree bit for bit
 * on the kernel's value range (variance sums in [2^-402, 2^261], mean differences in {0} U [2^-201, 2^129]). */
int wdx_selftest_score_dev(wdx_ctx *ctx, const double *d_dm, const double *d_vs, int64_t n, double *d_fast,
                           double *d_ref, void *stream);

/* Self-test of clip_bounds_kernel (A1 ahead of the fast fingerprint kernels: one wave per read): packed
 * reads (d_row_off int64[n_reads+1]) or rows of `stride` samples; cap = 4096, 5120 or 6144 selects the instantiation
 * (windows of 256..cap samples are taken).  d_rec: n_reads records of 16 bytes {float lo, hi, cmax; int32 flag}
 * (flag 0 not taken, 1 bounds valid + sums provably exact, 2 NaN / infinity / no non-negative sample, 3 exactness gate
 * fails or the negative-sample shortcut does not apply).  Must equal sig_proc.py:421-431's med -/+ thresh * mad bit for bit. */
int wdx_selftest_clip_dev(wdx_ctx *ctx, const float *d_sig, const int64_t *d_row_off, int64_t stride, int64_t n_reads,
                          const int32_t *d_a_start, const int32_t *d_a_end, const wdx_seg_params *p, int32_t cap,
                          void *d_rec, void *stream);

/* Diagnostic: stream n floats with coalesced dword loads (known byte count) to calibrate the
 * FETCH_SIZE PMC counter for the fingerprint kernel's access pattern. */
int wdx_calib_read_dev(wdx_ctx *ctx, const float *d_p, int64_t n, float *d_out, void *stream);

/* ---- synthetic input generator (bench / tests; spec "wdx-synth v1", warpdemux_amd/synth.py) */

/* Lengths (incl. both 100-sample pads) of reads first_read .. first_read+n-1 -> d_len int64[n] */
int wdx_synth_lengths_dev(wdx_ctx *ctx, uint64_t seed, int64_t first_read, int64_t n_reads,
                          int32_t n_barcodes, const int32_t *d_dwell_table /*1024*/,
                          int64_t *d_len, void *stream);
/* Fill d_sig (packed, offsets d_off int64[n+1]) and d_barcode int32[n] (nullable). */
int wdx_synth_fill_dev(wdx_ctx *ctx, uint64_t seed, int64_t first_read, int64_t n_reads,
                       int32_t n_barcodes, int32_t n_bc_events, float noise_scale, int32_t spikes,
                       const int32_t *d_dwell_table, const float *d_lead /*160*/,
                       const float *d_bc /*n_barcodes x 64*/, const int64_t *d_off, float *d_sig,
                       int32_t *d_barcode, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WDX_H */
