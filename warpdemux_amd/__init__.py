"""warpdemux_amd -- MI355X (gfx950) engine for the WarpDemuX sig_proc / parallel_distances hot path.

Host side is Python over a C-ABI shared library (include/wdx.h, warpdemux_amd/csrc).  Modules:

* ``parallel_distances`` -- drop-in for ``warpdemux.parallel_distances`` (same four functions)
* ``sig_proc``           -- batched ``detect_results_to_fpt`` + ``ReadResult`` mirror
* ``engine``             -- device-resident fused pipeline (raw adapter rows -> distances -> calls)
* ``dist``               -- one-process-per-GPU sharding + call-count all-reduce
* ``synth``              -- deterministic synthetic RNA004-like adapter signals
"""
from __future__ import annotations

__version__ = "0.1.0"


def install():
    """Make WarpDemuX use this engine for its DTW stage: replaces
    ``warpdemux.parallel_distances.distance_matrix_to`` (and the names the model modules bound at
    import: models/dtw_svm.py:18, models/dtw_mlp.py:17).  Call once per process, before predict."""
    import importlib
    import sys

    from . import parallel_distances as mine

    pd = importlib.import_module("warpdemux.parallel_distances")
    for name in ("distance_matrix_to", "parallel_distance_matrix_to", "parallel_distance_matrix", "compute_block_distance"):
        setattr(pd, name, getattr(mine, name))
    for modname in ("warpdemux.models.dtw_svm", "warpdemux.models.dtw_mlp"):
        m = sys.modules.get(modname)
        if m is not None:
            m.distance_matrix_to = mine.distance_matrix_to
