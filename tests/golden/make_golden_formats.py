"""Golden files for N4 from the REFERENCE's own writers (needs /root/reference; un-vendored deps stubbed).

    python tests/golden/make_golden_formats.py   # writes tests/golden/g7_formats.npz

The fixture stores the inputs, the bytes of barcode_fpts_0.npz written by file_proc.save_fpts_signals (with and
without dwell times) and the decompressed CSV text written by save_predictions after
add_read_id_col_to_predictions + models.utils.predictions_to_df.
"""
import gzip
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def main():
    from make_golden import import_reference

    import_reference()  # installs the stubs for adapted / dtaidistance / ruptures
    for name in ("toml",):
        try:
            __import__(name)
        except ImportError:
            m = types.ModuleType(name)
            m.load = lambda p: {}
            sys.modules[name] = m
    import importlib

    src = open("/root/reference/warpdemux/file_proc.py").read()
    ns = {}
    # only the three format functions are needed; file_proc imports far more than is installed here, so the
    # functions are taken from the module source by name and executed with their own imports
    import ast
    import pandas as pd
    from typing import List, Union

    tree = ast.parse(src)
    wanted = {"save_fpts_signals", "save_predictions", "add_read_id_col_to_predictions"}
    code = ast.Module([n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted], [])
    ns.update(np=np, pd=pd, List=List, Union=Union)
    exec(compile(code, "file_proc.py", "exec"), ns)
    utils = importlib.import_module("warpdemux.models.utils")

    rng = np.random.default_rng(5)
    n, K, k = 7, 25, 5
    ids = np.array([f"read-{i:04d}-{rng.integers(1 << 30):08x}" for i in range(n)])
    fpt = rng.normal(size=(n, K))
    dwell = rng.integers(5, 90, size=(n, K)).astype(np.int64)
    res = [SimpleNamespace(read_id=ids[i], barcode_fpt=fpt[i], dwell_times=dwell[i]) for i in range(n)]
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, flag in (("with_dwell", True), ("without_dwell", False)):
            p = os.path.join(d, f"barcode_fpts_{tag}.npz")
            ns["save_fpts_signals"](res, p, save_dwell_time=flag)
            out["npz_" + tag] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        prob = rng.dirichlet(np.ones(k), size=n)
        label_mapper = {0: 3, 1: 4, 2: 5, 3: 7, 4: -1}
        y_pred, conf = utils.process_probs(prob, label_mapper, np.array([0.17, 0.28, 0.23, 0.47, 1.01]))
        df = utils.predictions_to_df(y_pred, prob, conf, label_mapper)
        df = ns["add_read_id_col_to_predictions"](df, ids)
        p = os.path.join(d, "predictions.csv.gz")
        ns["save_predictions"](df, p)
        out["csv_text"] = np.frombuffer(gzip.open(p, "rb").read(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g7_formats.npz"), read_ids=ids, fpt=fpt, dwell=dwell, prob=prob,
                        y_pred=y_pred, conf=conf, label_keys=np.array(list(label_mapper)),
                        label_vals=np.array(list(label_mapper.values())), **out)
    print("written", {k: v.size for k, v in out.items()})


if __name__ == "__main__":
    main()
