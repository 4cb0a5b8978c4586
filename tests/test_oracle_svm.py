"""N1: the classifier tail of DTW_SVM.predict.  The oracle restates libsvm's
svm_predict_probability; here it is pinned against scikit-learn's own SVC (the library the reference
calls at models/dtw_svm.py:92), which is importable wherever these tests run."""
import numpy as np
import pytest

from oracle import wdx_oracle as orc

sklearn = pytest.importorskip("sklearn")


def make_model(n_classes, n_train, seed, L=25):
    from sklearn.svm import SVC

    rng = np.random.default_rng(seed)
    centers = rng.normal(size=(n_classes, L))
    y = rng.integers(0, n_classes, n_train)
    Xtr = centers[y] + 0.9 * rng.normal(size=(n_train, L))
    D = orc.dtw_matrix(Xtr, Xtr, 15, 0.1)
    K = np.exp(-1.0 * np.power(D, 1))            # pdist_kernel, models/dtw_svm.py:21-22 (float32)
    svc = SVC(kernel="precomputed", probability=True, C=1.0, random_state=seed).fit(K, y)
    return svc, Xtr, centers, rng


@pytest.mark.parametrize("n_classes,n_train", [(2, 80), (3, 150), (5, 300), (11, 500)])
def test_predict_proba_matches_sklearn(n_classes, n_train):
    svc, Xtr, centers, rng = make_model(n_classes, n_train, seed=n_classes)
    yq = rng.integers(0, n_classes, 64)
    Xq = centers[yq] + 0.9 * rng.normal(size=(64, Xtr.shape[1]))
    Kq = np.exp(-1.0 * np.power(orc.dtw_matrix(Xq, Xtr, 15, 0.1), 1))   # float32, like the reference
    ref = svc.predict_proba(Kq)
    n_support, support, dual_coef, rho, probA, probB, k = orc.svm_params(svc)
    got, dec = orc.svm_predict_proba(Kq, n_support, support, dual_coef, rho, probA, probB, want_dec=True)
    assert got.shape == ref.shape == (64, n_classes)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    # one-vs-one decision values against sklearn's own
    svc.decision_function_shape = "ovo"
    sign = -1.0 if n_classes == 2 else 1.0   # sklearn flips the sign of the binary decision function
    np.testing.assert_allclose(dec, sign * svc.decision_function(Kq).reshape(64, -1), rtol=1e-12, atol=1e-12)
    assert np.array_equal(np.argmax(got, axis=1), np.argmax(ref, axis=1))


G6 = {"WDX4_rna004": "g6_dtw_svm_wdx4.npz", "WDX10_rna004": "g6b_dtw_svm_wdx10.npz", "WDX12_rna002": "g6c_dtw_svm_wdx12_rna002.npz"}


def load_g6(which="WDX4_rna004"):
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", G6[which]))
    label_mapper = {int(k): int(v) for k, v in zip(g["label_keys"], g["label_vals"])}
    return g, label_mapper


@pytest.mark.parametrize("which", list(G6))
def test_oracle_matches_reference_model_golden(which):
    """g6 / g6b / g6c: the reference's DTW_SVM.predict on its shipped WDX4_rna004_v1_0 and WDX10_rna004_v1_0 models
    and on DEPRECATED WDX12_rna002_v0_4_4 (made by tests/golden/make_golden_svm.py).  The oracle's tail from the
    stored parameters must reproduce it."""
    g, label_mapper = load_g6(which)
    D = orc.dtw_matrix(g["Xq"], g["X_train"], int(g["window"]), float(g["penalty"]))
    K = np.exp(-float(g["gamma"]) * np.power(D, int(g["pwr_dist"])))
    assert K.dtype == np.float32
    prob = orc.svm_predict_proba(K, g["n_support"].astype(np.int32), g["support"].astype(np.int32), g["dual_coef"],
                                 -g["intercept"], g["probA"], g["probB"])
    np.testing.assert_allclose(prob, g["y_prob"], rtol=0, atol=1e-12)
    idx = np.argmax(prob, axis=1)
    pred = np.array([label_mapper[i] for i in idx])
    srt = np.sort(prob, axis=1)
    conf = srt[:, -1] - srt[:, -2]
    pred[conf < g["thresholds"][idx]] = -1
    assert np.array_equal(pred, g["y_pred"])
    assert (g["y_pred"] == -1).sum() > 10 and len(np.unique(g["y_pred"])) == len(label_mapper)
