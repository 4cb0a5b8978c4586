"""Run by tests/test_gpu_live.py in a FRESH interpreter: the reference worker's whole minibatch
(`worker_detect_and_predict_on_preloaded_signals`, file_proc.py:380-454: ReadResults AND model.predict of the stacked
fingerprints) through warpdemux_amd.feeder.Feeder with a REAL reference model -- fixture g6b = the numbers of
WDX10_rna004_v1_0.joblib and what the reference's own DTW_SVM.predict returned for 256 query fingerprints.

Forked workers (which never touch the GPU):
  * feeder.predict(Xq) against g6b (the reference's probabilities within 1e-5, labels wherever its margin is not at a tie);
  * feeder.detect_and_predict(minibatch) -- fingerprints / dwell / statistics / status against the oracle bit for bit, the
    predictions against the oracle's libsvm restatement on the oracle's distances (1e-5) and, in the parent afterwards,
    against this package's own DTW_SVM.predict on the same fingerprints (same kernels: bit for bit);
  * the ReadResult records built from the batch (sig_proc.read_results_from_batch) and the predictions DataFrame.
Prints one JSON line."""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import multiprocessing as mp  # noqa: E402

import numpy as np  # noqa: E402

from oracle import wdx_oracle as orc  # noqa: E402
from warpdemux_amd import sig_proc, synth  # noqa: E402
from warpdemux_amd.feeder import Feeder  # noqa: E402
from warpdemux_amd.models import DTW_SVM  # noqa: E402

FEEDER = None
G = None
K = 25


def load_model():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g6b_dtw_svm_wdx10.npz"))
    label_mapper = {int(k): int(v) for k, v in zip(g["label_keys"], g["label_vals"])}
    m = DTW_SVM(g["X_train"], g["n_support"], g["support"], g["dual_coef"], -g["intercept"], g["probA"], g["probB"],
                label_mapper, g["thresholds"], window=int(g["window"]), penalty=float(g["penalty"]),
                gamma=float(g["gamma"]), pwr_dist=int(g["pwr_dist"]), block_size=int(g["block_size"]))
    return g, m


def minibatch(widx):
    spec = synth.SynthSpec(n_barcodes=10)
    mb, a_s, a_e, _ = synth.generate_minibatch(spec, 5000 * widx, 96, 9000, start_jitter=700)
    ok_in = np.ones(96, dtype=np.uint8)
    ok_in[3::17] = 0
    return mb, a_s, a_e, ok_in


def work(widx):
    g = G
    rec = {}
    # model.predict on fingerprints the worker holds, against the reference's own output
    pred, prob = FEEDER.predict(g["Xq"])
    srt = np.sort(g["y_prob"], axis=1)
    conf = srt[:, -1] - srt[:, -2]
    safe = (conf > 1e-4) & (np.abs(conf - g["thresholds"][np.argmax(g["y_prob"], axis=1)]) > 1e-4)
    rec["predict_max_abs_prob_err_vs_reference"] = float(np.abs(prob - g["y_prob"]).max())
    rec["predict_labels"] = bool(np.array_equal(pred[safe], g["y_pred"][safe]))
    df = FEEDER.predict(g["Xq"], return_df=True)
    rec["predict_df"] = bool(list(df.columns) == list(g["df_cols"])
                             and np.allclose(df["confidence_score"].to_numpy(), g["df_conf"], atol=1.5e-3))
    # the worker's minibatch: ReadResults' arrays and predictions from one pass
    mb, a_s, a_e, ok_in = minibatch(widx)
    fb, (y_pred, y_prob) = FEEDER.detect_and_predict(mb, a_s, a_e, success=ok_in)
    fpt, dwell, stats, status = orc.fingerprint_batch(mb, a_s, a_e, orc.SegParams(barcode_num_events=K), ok=ok_in)
    ok = status == 0
    rec["fingerprints"] = bool(np.array_equal(fb.status, status) and np.array_equal(fb.fpt[ok].view(np.uint64), fpt[ok].view(np.uint64))
                               and np.array_equal(fb.dwell[ok], dwell[ok])
                               and np.array_equal(fb.stats[ok].view(np.uint64), stats[ok].view(np.uint64)))
    rec["rows"] = [int(ok.sum()), int(y_pred.shape[0]), int(y_prob.shape[0])]
    D = orc.dtw_matrix(fpt[ok], np.ascontiguousarray(g["X_train"], dtype=np.float64), int(g["window"]), float(g["penalty"]))
    Kq = np.exp(-float(g["gamma"]) * np.power(D, int(g["pwr_dist"])))      # float32, like the reference (models/dtw_svm.py:21-22)
    o_prob = orc.svm_predict_proba(Kq, g["n_support"].astype(np.int32), g["support"].astype(np.int32), g["dual_coef"],
                                   -g["intercept"], g["probA"], g["probB"])
    rec["e2e_max_abs_prob_err_vs_oracle"] = float(np.abs(y_prob - o_prob).max())
    # the reference worker's two products: ReadResult records and the predictions DataFrame
    det = [sig_proc.DetectResults(bool(ok_in[i]), "" if ok_in[i] else "adapter not found", int(a_s[i]), int(a_e[i])) for i in range(96)]
    rr = sig_proc.read_results_from_batch(fb, det, [f"read{i}" for i in range(96)])
    fb2, dfp = FEEDER.detect_and_predict(mb, a_s, a_e, success=ok_in, return_df=True)
    rec["records"] = bool(len(rr) == 96 and sum(r.success for r in rr) == int(ok.sum()) and len(dfp) == int(ok.sum())
                          and all(np.array_equal(r.barcode_fpt, fb.fpt[i]) for i, r in enumerate(rr) if r.success)
                          and rr[3].fail_reason == "adapter not found" and list(dfp.columns) == list(g["df_cols"]))
    return os.getpid(), rec, y_pred.tolist(), y_prob.tolist()


if __name__ == "__main__":
    G_, model = load_model()
    G = {k: G_[k] for k in G_.files}
    FEEDER = Feeder(model=model, params=sig_proc.SegParams(barcode_num_events=K), max_reads=256, stride=9000, n_slots=4)
    ctx = mp.get_context("fork")
    with ProcessPoolExecutor(max_workers=3, mp_context=ctx) as ex:
        res = list(ex.map(work, range(3)))
    FEEDER.close()
    # the parent's own context, after every fork: this package's DTW_SVM.predict on the oracle-checked fingerprints must
    # give the feeder's numbers bit for bit (same kernels on the same distance rows)
    same = []
    for widx, (_, _, y_pred, y_prob) in enumerate(res):
        mb, a_s, a_e, ok_in = minibatch(widx)
        fb = sig_proc.fingerprint_batch(mb, a_s, a_e, sig_proc.SegParams(barcode_num_events=K), success=ok_in)
        p2, q2 = model.predict(fb.fpt[fb.status == 0])
        same.append(bool(np.array_equal(p2, np.array(y_pred)) and np.array_equal(q2.view(np.uint64), np.array(y_prob).view(np.uint64))))
    print(json.dumps({"pids": sorted({r[0] for r in res}), "parent": os.getpid(), "workers": [r[1] for r in res],
                      "feeder_equals_in_process_predict": same}))
