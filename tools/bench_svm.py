"""N1 throughput: fingerprints resident in HBM -> DTW vs the model's training set -> SVM tail.

    python tools/bench_svm.py [n_reads] [n_train] [n_classes]

The model is a scikit-learn SVC fitted here on synthetic fingerprints of the shipped models' shape
(25-point fingerprints, window 15, penalty 0.1); sizes default to the largest shipped model's order
(nY ~ 3.6k training rows, 12 classes would be WDX12; default 851 x 5 is WDX4_rna004_v1_0's).
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import wdx_oracle as orc  # model fitting needs a kernel matrix (test infrastructure)
from warpdemux_amd import _lib
from warpdemux_amd.engine import DemuxEngine
from warpdemux_amd.models import DTW_SVM


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 851
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    from sklearn.svm import SVC

    rng = np.random.default_rng(1)
    L = 25
    centers = rng.normal(size=(k, L))
    y = rng.integers(0, k, n_train)
    Xtr = centers[y] + 0.9 * rng.normal(size=(n_train, L))
    K = np.exp(-orc.dtw_matrix(Xtr, Xtr, 15, 0.1))
    svc = SVC(kernel="precomputed", probability=True, random_state=0).fit(K, y)
    p = orc.svm_params(svc)
    model = DTW_SVM(Xtr, *p[:6], {i: i for i in range(k)}, None, 15, 0.1, block_size=1000)
    eng = DemuxEngine(Xtr, 15, 0.1)
    eng.set_svm(model)
    yq = torch.randint(0, k, (n,), device="cuda")
    Xq = torch.from_numpy(centers).cuda()[yq] + 0.9 * torch.randn((n, L), dtype=torch.float64, device="cuda")
    chunk = max(1, min(n, (1 << 30) // (4 * n_train)))
    _lib.check(eng.L.wdx_kernel_timing(eng.ctx.handle, 1))

    def run():
        out = []
        for r0 in range(0, n, chunk):
            d, _ = eng.dtw(Xq[r0:r0 + chunk], want_argmin=False)
            out.append(eng.svm_predict(d)[1])
        return out

    run()
    torch.cuda.synchronize()
    eng.kernel_time_reset()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        pred = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    dtw_ms, _ = eng.kernel_time(1)
    tr_ms, _ = eng.kernel_time(2)
    svm_ms, _ = eng.kernel_time(4)
    acc = float((torch.cat(pred).cpu().numpy() == yq.cpu().numpy()).mean())
    print(json.dumps({
        "reads": n, "n_train": n_train, "n_sv": int(svc.support_.size), "classes": k,
        "reads_per_s": n / dt, "ms_total": dt * 1e3, "ms_dtw": dtw_ms / reps, "ms_transpose": tr_ms / reps,
        "ms_svm": svm_ms / reps, "dtw_pairs_per_s": n * n_train / (dtw_ms / reps * 1e-3),
        "svm_dist_GBps": n * n_train * 4 / (svm_ms / reps * 1e-3) / 1e9, "accuracy_vs_true_class": acc}))


if __name__ == "__main__":
    main()
