"""Data formats on either side of the hot path, written straight from batch arrays.

Mirrors of /root/reference/warpdemux/file_proc.py:
  save_fpts_signals              :726-754   barcode_fpts_<batch>.npz  {num_reads, read_ids, signals[, dwell_times]}
  save_predictions               :757-766   gzip CSV of the predictions DataFrame
  add_read_id_col_to_predictions :769-780   '#read_id' as first column
The reference builds these from a list of per-read ``ReadResult`` objects; the batch entry points here take the
arrays of a ``FingerprintBatch`` (sig_proc.fingerprint_batch) so that no per-read Python objects are needed, and
produce byte-for-byte the same files for the same content (tests/test_file_formats.py, fixture g7).
"""
from __future__ import annotations

from typing import List, Sequence, Union

import numpy as np


_ID_COLUMN = "#read_id"


def save_fpts_arrays(read_ids: Sequence[str], barcode_fpts: np.ndarray, dwell_times: np.ndarray, filename: str,
                     save_dwell_time: bool = True):
    """barcode_fpts_<batch>.npz from batch arrays: (n,) ids, (n, K) float64 fingerprints and (n, K) int64 dwell times
    of the successful reads.  Member names and ORDER are the file format (file_proc.py:741-752): num_reads, read_ids,
    signals[, dwell_times] -- `continue` mode and `predict` read them back (file_proc.py:128-185, 282-330)."""
    members = {"num_reads": len(read_ids), "read_ids": np.asarray(read_ids), "signals": np.asarray(barcode_fpts)}
    if save_dwell_time:
        members["dwell_times"] = np.asarray(dwell_times)
    np.savez(filename, **members)
    return members["read_ids"], members["signals"], np.asarray(dwell_times)


def save_fpts_signals(list_of_processing_results, filename: str, save_dwell_time: bool = True):
    """Same signature as file_proc.py:726-754 (a list of objects with read_id / barcode_fpt / dwell_times), for callers
    that still hold per-read results; the arrays go through `save_fpts_arrays`."""
    rows = [(r.read_id, r.barcode_fpt, r.dwell_times) for r in list_of_processing_results]
    ids, fpts, dwells = (np.array(col) for col in zip(*rows)) if rows else (np.array([]),) * 3
    return save_fpts_arrays(ids, fpts, dwells, filename, save_dwell_time)


def load_fpts_signals(filename: str):
    """(read_ids, signals, dwell_times or None) of a file written by either writer above."""
    with np.load(filename) as z:
        n = int(z["num_reads"])
        ids, sig = z["read_ids"], z["signals"]
        dw = z["dwell_times"] if "dwell_times" in z.files else None
    if len(ids) != n or len(sig) != n:
        raise ValueError(f"{filename}: num_reads={n} does not match the stored arrays")
    return ids, sig, dw


def add_read_id_col_to_predictions(predictions, read_ids: Union[List[str], np.ndarray]):
    """'#read_id' as the first column of the predictions frame (file_proc.py:769-780; same error on a second call)."""
    if _ID_COLUMN in predictions.columns:
        raise ValueError(f"'{_ID_COLUMN}' already in dataframe")
    out = predictions.copy(deep=False)
    out.insert(0, _ID_COLUMN, read_ids)
    predictions[_ID_COLUMN] = read_ids   # the reference leaves the column on its argument too
    return out


def predictions_frame(read_ids, y_pred, y_prob, conf, label_mapper):
    """The batch's predictions table straight from the engine's arrays (models/utils.py:36-43 + file_proc.py:769-780):
    #read_id, predicted_barcode, confidence_score (3 decimals), one pXX column per class (4 decimals)."""
    import pandas as pd

    cols = {_ID_COLUMN: np.asarray(read_ids), "predicted_barcode": np.asarray(y_pred),
            "confidence_score": np.round(np.asarray(conf), 3)}
    prob = np.asarray(y_prob)
    for i in range(prob.shape[1]):
        cols[f"p{label_mapper[i]:02d}"] = np.round(prob[:, i], 4)
    return pd.DataFrame(cols)


def save_predictions(predictions, filename: str) -> None:
    """barcode_predictions_<batch>.csv.gz (file_proc.py:757-766): no index column, gzip."""
    predictions.to_csv(filename, index=False, compression="gzip")
