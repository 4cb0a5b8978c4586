"""Multi-GPU layer: one process per GPU, reads sharded contiguously, no data-path collective.

The path is embarrassingly parallel over reads; the only exchange is the final
per-barcode call histogram, an int64[nY+1] all-reduce (RCCL over xGMI when the backend is "nccl",
gloo in the CPU tests).
"""
from __future__ import annotations

import os
from typing import Tuple


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous read range [lo, hi) of `rank`: GPU g gets reads [g*ceil(n/G), (g+1)*ceil(n/G))."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-n_total // world)
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def init_process_group(backend: str | None = None):
    """Initialise torch.distributed from the environment if WORLD_SIZE > 1.  backend defaults to
    "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo"."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank_world()
    if world == 1 or dist.is_initialized():
        return rank, local_rank, world
    if backend == "nccl" and torch.cuda.device_count() <= local_rank:
        raise RuntimeError(f"rank {rank}: LOCAL_RANK {local_rank} has no GPU (visible: {torch.cuda.device_count()})")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def reduce_counts(counts):
    """In-place SUM all-reduce of the per-barcode call histogram (int64 tensor, any device the
    process group supports).  No-op for a single process."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


class CountReducer:
    """The path's one exchange: in-place SUM of the int64 call histogram over all ranks.

    ``mode == "rccl"``: the C ABI's own communicator (``wdx_comm_init`` / ``wdx_reduce_counts``,
    include/wdx.h) -- rank 0 draws the id with ``wdx_comm_unique_id`` and the 128 bytes travel over the
    already initialised torch.distributed group (any side channel would do).  ``mode == "torch"``:
    ``torch.distributed.all_reduce`` on the process group (gloo in the CPU tests; nccl = RCCL too).
    ``mode == "single"``: one process, nothing to do.  ``prefer`` = "rccl" | "torch" | None (rccl when
    the group's backend is nccl and the engine is on a GPU).

    Every rank takes the same road, decided BEFORE anyone enters a collective of the C ABI:
    1. each rank asks ``wdx_comm_available()`` (local) and the answers meet in a MIN all-reduce on the
       process group -- if librccl cannot be bound on ANY rank (WDX_ERR_NO_DEVICE) ALL ranks use
       torch.distributed, and ``note`` says why;
    2. rank 0 draws the id; the broadcast ALWAYS runs -- an error on rank 0 travels in it and is raised
       on every rank (nobody is left waiting in the broadcast);
    3. ``wdx_comm_init`` (collective); any failure there is a real RCCL fault and is raised, never
       turned into a silent torch.distributed run;
    4. ``rccl_ranks`` = what RCCL itself reports for the communicator (``ncclCommCount``) must equal
       the world size."""

    def __init__(self, ctx=None, prefer: str | None = None):
        import torch.distributed as dist

        self.ctx = ctx
        self.mode = "single"
        self.note = ""
        self.rccl_ranks = None
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        self.mode = "torch"
        if prefer not in (None, "rccl", "torch"):
            raise ValueError("prefer must be 'rccl', 'torch' or None")
        want_rccl = prefer == "rccl" or (prefer is None and dist.get_backend() == "nccl")
        if not (want_rccl and ctx is not None):
            return
        import torch

        why = self._rccl_unavailable()
        flag = torch.tensor([0 if why else 1], dtype=torch.int32,
                            device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            self.note = ("RCCL not bindable on %s (%s); torch.distributed used on all ranks"
                         % ("this rank" if why else "another rank", why or "see that rank's note"))
            return
        self._init_rccl(dist)
        self.mode = "rccl"

    def _rccl_unavailable(self) -> str:
        """'' when this process can bind librccl, else the reason (only WDX_ERR_NO_DEVICE counts as
        'unavailable'; anything else is raised)."""
        from . import _lib

        self._L = _lib.load()
        try:
            _lib.check(self._L.wdx_comm_available())
        except _lib.WdxNoDevice as e:
            return str(e)
        return ""

    def _draw_id(self) -> bytes:
        import ctypes as C

        from . import _lib

        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(self._L.wdx_comm_unique_id(buf))
        return bytes(buf.raw)

    def _init_rccl(self, dist):
        import ctypes as C

        from . import _lib

        rank, world = dist.get_rank(), dist.get_world_size()
        box = [None]
        if rank == 0:
            try:
                box[0] = ("id", self._draw_id())
            except Exception as e:  # noqa: BLE001 -- travels to every rank in the broadcast below, raised there
                box[0] = ("error", f"{type(e).__name__}: {e}")
        dist.broadcast_object_list(box, src=0)
        kind, payload = box[0]
        if kind != "id":
            raise _lib.WdxError(f"rank 0 could not create the RCCL id: {payload}")
        _lib.check(self._L.wdx_comm_init(self.ctx.handle, payload, rank, world))
        r, w, cnt = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        _lib.check(self._L.wdx_comm_info(self.ctx.handle, C.byref(r), C.byref(w), C.byref(cnt)))
        if (r.value, w.value) != (rank, world) or cnt.value != world:
            raise _lib.WdxError(f"RCCL communicator reports {cnt.value} ranks (bound as rank {r.value} of {w.value}); "
                                f"the process group has {world}")
        self.rccl_ranks = cnt.value

    def __call__(self, counts, stream=None):
        """counts: int64 torch tensor (device tensor for "rccl"); reduced in place."""
        if self.mode == "rccl":
            import ctypes as C

            from . import _lib

            if counts.device.type != "cuda":
                raise ValueError("wdx_reduce_counts needs the device histogram")
            _lib.check(self._L.wdx_reduce_counts(self.ctx.handle, C.c_void_p(counts.data_ptr()),
                                                 int(counts.numel()), stream))
            return counts
        if self.mode == "torch":
            return reduce_counts(counts)
        return counts

    def close(self):
        if self.mode == "rccl":
            self._L.wdx_comm_destroy(self.ctx.handle)
            self.mode = "torch"


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def min_over_ranks(value: float, device=None) -> float:
    """MIN all-reduce of a host scalar (agreement between ranks)."""
    return -max_over_ranks(-value, device)


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (timing)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, device=None):
    """All ranks' host scalars, in rank order (per-rank timing: which rank was the straggler)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.zeros(world, dtype=torch.float64, device=device if device is not None else "cpu")
    t[rank] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu().tolist()]
