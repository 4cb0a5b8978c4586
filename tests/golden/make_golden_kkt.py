"""Fixtures g9 / g9b: the numeric parameters of the reference's DTW_SVM models, for the KKT pin of the DTW
restatement (tests/helpers/kkt.py explains the conditions).  g9 = the five shipped rna004 models
(warpdemux/models/model_files, gamma 1, C 1); g9b = the six rna002 v0.4.4 models the reference keeps under
DEPRECATED/model_files (gamma 1.2, C 10, up to 3 617 x 25 / 13 classes -- WDX12 is the model of the reference's
live run, notebooks/Live_Run_8_FINAL_RUN_SARS2).

Runs only in the build container (needs /root/reference).  It unpickles every
*.joblib of the two directories (reference DATA files, CC BY-NC: kept as a test fixture only),
stores `_X`, the libsvm dual coefficients / intercepts / class bounds and the model's DTW parameters,
and records the KKT residuals the oracle DTW and each negative control gave at generation time
(`<model>__residuals`, rows in the order of `variant_names`; columns free_max_abs, bound_max,
zero_min).  Nothing of dtaidistance is needed or stubbed with arithmetic here: the models themselves
encode the genuine library's distances.

    python tests/golden/make_golden_kkt.py        # writes tests/golden/g9_kkt_models.npz and g9b_kkt_models_rna002.npz
"""
import glob
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from helpers import kkt  # noqa: E402
from oracle import wdx_oracle as orc  # noqa: E402


def main():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # import-time names only (models/dtw_svm.py:18 -> parallel_distances.py:14); never called here
    mod("dtaidistance")
    sys.modules["dtaidistance"].dtw = mod("dtaidistance.dtw", distance_matrix=None)
    try:
        import toml  # noqa: F401
    except ImportError:
        mod("toml", load=lambda p: {})
    sys.path.insert(0, REF)
    import joblib

    warnings.simplefilter("ignore")
    only = sys.argv[1:]
    for sub, dst_name in (("warpdemux/models/model_files", "g9_kkt_models.npz"),
                          ("DEPRECATED/model_files", "g9b_kkt_models_rna002.npz")):
        if only and dst_name.split("_")[0] not in only:
            continue
        make(joblib, sorted(glob.glob(os.path.join(REF, sub, "*.joblib"))), os.path.join(HERE, dst_name))


def threaded(fn, X, w, p, rows=128):
    """the oracle matrix in row blocks on all cores (ctypes drops the GIL): WDX12 is 13 M pairs"""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as ex:
        return np.concatenate(list(ex.map(lambda a: fn(X[a:a + rows], X, w, p), range(0, X.shape[0], rows))))


def make(joblib, paths, dst):
    out = {}
    names = []
    vnames = None
    for path in paths:
        name = os.path.basename(path).split("_")[0]
        m = joblib.load(path)
        svc = m.model
        assert svc.kernel == "precomputed" and np.array_equal(svc.support_, np.arange(m._X.shape[0]))
        assert svc.tol == kkt.EPS_LIBSVM
        X = np.ascontiguousarray(m._X, dtype=np.float64)
        mdl = {"X": X, "n_support": np.asarray(svc._n_support, dtype=np.int32),
               "dual_coef": np.asarray(svc._dual_coef_, dtype=np.float64),
               "intercept": np.asarray(svc._intercept_, dtype=np.float64),
               "c_bound": np.asarray(svc.C * svc.class_weight_, dtype=np.float64),
               "gamma": np.float64(m.gamma), "pwr_dist": np.int32(m.pwr_dist),
               "window": np.int32(m.window), "penalty": np.float64(m.penalty)}
        dtw = lambda X_, w, p: threaded(orc.dtw_matrix, X_, w, p)  # noqa: E731
        vs = kkt.variants(dtw, X, int(m.window), float(m.penalty))
        vnames = list(vs)
        rows = []
        for vn, fn in vs.items():
            r = kkt.kkt_residuals(fn(), mdl["n_support"], mdl["dual_coef"], mdl["intercept"], mdl["c_bound"],
                                  float(m.gamma), int(m.pwr_dist))
            rows.append([r["free_max_abs"], r["bound_max"], r["zero_min"]])
            print("%-6s %-22s free %.2e bound %+.2e zero %+.2e  (n %d/%d/%d)" % (
                name, vn, r["free_max_abs"], r["bound_max"], r["zero_min"], r["n_free"], r["n_bound"], r["n_zero"]))
        mdl["residuals"] = np.array(rows)
        for k, v in mdl.items():
            out[f"{name}__{k}"] = v
        names.append(name)
    out["models"] = np.array(names)
    out["variant_names"] = np.array(vnames)
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
