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


def save_fpts_signals(list_of_processing_results, filename: str, save_dwell_time: bool = True):
    """file_proc.py:726-754 -- same signature (a list of objects with read_id / barcode_fpt / dwell_times)."""
    read_ids = np.array([res.read_id for res in list_of_processing_results])
    barcode_fpts = np.array([res.barcode_fpt for res in list_of_processing_results])
    dwell_times = np.array([res.dwell_times for res in list_of_processing_results])
    return save_fpts_arrays(read_ids, barcode_fpts, dwell_times, filename, save_dwell_time)


def save_fpts_arrays(read_ids: Sequence[str], barcode_fpts: np.ndarray, dwell_times: np.ndarray, filename: str,
                     save_dwell_time: bool = True):
    """The same file from batch arrays: (n,) ids, (n, K) float64 fingerprints, (n, K) int64 dwell times of the
    successful reads."""
    read_ids = np.asarray(read_ids)
    barcode_fpts = np.asarray(barcode_fpts)
    dwell_times = np.asarray(dwell_times)
    num_reads = len(read_ids)
    if save_dwell_time:
        np.savez(filename, num_reads=num_reads, read_ids=read_ids, signals=barcode_fpts, dwell_times=dwell_times)
    else:
        np.savez(filename, num_reads=num_reads, read_ids=read_ids, signals=barcode_fpts)
    return read_ids, barcode_fpts, dwell_times


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
    """file_proc.py:769-780"""
    cols = predictions.columns.tolist()
    if "#read_id" in cols:
        raise ValueError("'#read_id' already in dataframe")
    predictions["#read_id"] = read_ids
    return predictions[["#read_id", *cols]]


def save_predictions(predictions, filename: str) -> None:
    """file_proc.py:757-766"""
    predictions.to_csv(filename, index=False, compression="gzip")
