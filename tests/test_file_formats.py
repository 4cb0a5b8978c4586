"""N4: the .npz fingerprint files and the predictions CSV written by warpdemux_amd.file_formats
against files written by the reference's own file_proc functions (fixture g7, tests/golden/make_golden_formats.py)."""
import gzip
import io
import os
from types import SimpleNamespace

import numpy as np
import pytest

from warpdemux_amd import file_formats as ff
from warpdemux_amd.models import predictions_to_df

G = os.path.join(os.path.dirname(__file__), "golden", "g7_formats.npz")


def _members(buf):
    with np.load(io.BytesIO(buf.tobytes())) as z:
        return [(k, z[k].dtype, z[k].shape, z[k].copy()) for k in z.files]


@pytest.mark.parametrize("with_dwell", [True, False])
def test_fingerprint_npz_matches_reference_writer(tmp_path, with_dwell):
    g = np.load(G)
    ids, fpt, dwell = g["read_ids"], g["fpt"], g["dwell"]
    ref = _members(g["npz_with_dwell" if with_dwell else "npz_without_dwell"])
    res = [SimpleNamespace(read_id=ids[i], barcode_fpt=fpt[i], dwell_times=dwell[i]) for i in range(len(ids))]
    for writer, args in ((ff.save_fpts_signals, (res,)), (ff.save_fpts_arrays, (ids, fpt, dwell))):
        p = str(tmp_path / f"barcode_fpts_{writer.__name__}.npz")
        writer(*args, p, save_dwell_time=with_dwell)
        got = _members(np.frombuffer(open(p, "rb").read(), dtype=np.uint8))
        assert [(k, d, s) for k, d, s, _ in got] == [(k, d, s) for k, d, s, _ in ref]   # keys in order, dtypes, shapes
        for (_, _, _, a), (_, _, _, b) in zip(got, ref):
            assert np.array_equal(a, b)
        rid, sig, dw = ff.load_fpts_signals(p)
        assert np.array_equal(rid, ids) and np.array_equal(sig, fpt)
        assert (dw is None) == (not with_dwell) and (dw is None or np.array_equal(dw, dwell))


def test_predictions_csv_matches_reference_writer(tmp_path):
    g = np.load(G)
    label_mapper = {int(k): int(v) for k, v in zip(g["label_keys"], g["label_vals"])}
    df = predictions_to_df(g["y_pred"], g["prob"], g["conf"], label_mapper)
    df = ff.add_read_id_col_to_predictions(df, g["read_ids"])
    assert list(df.columns[:3]) == ["#read_id", "predicted_barcode", "confidence_score"] and df.columns[-1] == "p-1"
    with pytest.raises(ValueError):
        ff.add_read_id_col_to_predictions(df, g["read_ids"])
    p = str(tmp_path / "predictions.csv.gz")
    ff.save_predictions(df, p)
    assert gzip.open(p, "rb").read() == g["csv_text"].tobytes()
    # the same table straight from the arrays (no intermediate frame, no per-read objects)
    p2 = str(tmp_path / "predictions2.csv.gz")
    ff.save_predictions(ff.predictions_frame(g["read_ids"], g["y_pred"], g["prob"], g["conf"], label_mapper), p2)
    assert gzip.open(p2, "rb").read() == g["csv_text"].tobytes()
