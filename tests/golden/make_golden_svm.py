"""Golden vectors for N1 (DTW_SVM.predict) from the REFERENCE's own model class and a shipped model.

Runs only in the build container (needs /root/reference).  It unpickles
warpdemux/models/model_files/WDX4_rna004_v1_0.joblib (a reference data file), calls the reference's
``DTW_SVM.predict`` (models/dtw_svm.py:54-98; scikit-learn's libsvm does the classification) on
query fingerprints made here, and stores the model's numeric parameters, the queries and the
reference's outputs.  The un-vendored ``dtaidistance`` is absent from the image, so its
``dtw.distance_matrix`` is provided by this repo's oracle DTW (stage B stays "parity unpinned",
oracle/wdx_oracle.c header); everything after the distance matrix is the reference + scikit-learn.

    python tests/golden/make_golden_svm.py        # writes tests/golden/g6_dtw_svm_wdx4.npz
    python tests/golden/make_golden_svm.py more   # g6b (WDX10_rna004_v1_0: 2 601 x 25, 11 classes -- the headline's barcode
                                                  # set) and g6c (DEPRECATED WDX12_rna002_v0_4_4: 3 617 x 25, 13 classes, gamma
                                                  # 1.2 -- the model of the reference's live run); g6 itself is left alone
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import wdx_oracle as orc  # noqa: E402


def _distance_matrix(s, block=None, window=None, penalty=None, **kw):
    """Only the call shape parallel_distances.py:34-43,59-67 uses: rows [0,nx) against rows [nx,n)."""
    (r0, r1), (c0, c1) = block
    s = np.asarray(s, dtype=np.float64)
    out = np.full((s.shape[0], s.shape[0]), np.inf)
    out[r0:r1, c0:c1] = orc.dtw_matrix(s[r0:r1], s[c0:c1], window, penalty)  # float32, as the caller casts anyway
    return out


def main():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("dtaidistance")
    sys.modules["dtaidistance"].dtw = mod("dtaidistance.dtw", distance_matrix=_distance_matrix)
    try:
        import toml  # noqa: F401
    except ImportError:
        mod("toml", load=lambda p: {})
    sys.path.insert(0, REF)
    import joblib

    warnings.simplefilter("ignore")
    jobs = [("warpdemux/models/model_files/WDX4_rna004_v1_0.joblib", "g6_dtw_svm_wdx4.npz", 20261003)]
    if sys.argv[1:] == ["more"]:
        jobs = [("warpdemux/models/model_files/WDX10_rna004_v1_0.joblib", "g6b_dtw_svm_wdx10.npz", 20261004),
                ("DEPRECATED/model_files/WDX12_rna002_v0_4_4.joblib", "g6c_dtw_svm_wdx12_rna002.npz", 20261005)]
    for rel, dst, seed in jobs:
        make(joblib, os.path.join(REF, rel), os.path.join(HERE, dst), seed)


def make(joblib, path, out, seed):
    m = joblib.load(path)
    svc = m.model
    rng = np.random.default_rng(seed)
    # queries: training fingerprints with noise (confident calls), mixtures of two classes (low margins,
    # exercises the thresholds) and pure noise rows (the noise class)
    n_tr = m._X.shape[0]
    a = m._X[rng.integers(0, n_tr, 160)] + 0.25 * rng.normal(size=(160, m._X.shape[1]))
    w = rng.uniform(0.3, 0.7, (64, 1))
    b = w * m._X[rng.integers(0, n_tr, 64)] + (1 - w) * m._X[rng.integers(0, n_tr, 64)]
    c = rng.normal(size=(32, m._X.shape[1]))
    Xq = np.ascontiguousarray(np.vstack([a, b, c]))
    y_pred, y_prob = m.predict(Xq, nproc=1)
    df = m.predict(Xq, nproc=1, return_df=True)
    np.savez_compressed(
        out,
        X_train=m._X, window=m.window, penalty=m.penalty, block_size=m.block_size, gamma=m.gamma,
        pwr_dist=m.pwr_dist, label_keys=np.array(sorted(m.label_mapper)),
        label_vals=np.array([m.label_mapper[k] for k in sorted(m.label_mapper)]), thresholds=m.thresholds,
        n_support=svc._n_support, support=svc.support_, dual_coef=svc._dual_coef_, intercept=svc._intercept_,
        probA=svc._probA, probB=svc._probB,
        Xq=Xq, y_pred=y_pred, y_prob=y_prob, df_pred=df["predicted_barcode"].to_numpy(),
        df_conf=df["confidence_score"].to_numpy(), df_cols=np.array(list(df.columns)),
        df_probs=df[[c for c in df.columns if c.startswith("p")]].to_numpy(),
    )
    print(out, os.path.getsize(out), "bytes; labels", np.unique(y_pred, return_counts=True))


if __name__ == "__main__":
    main()
