"""Run by tests/test_gpu_live.py in fresh interpreters: TWO ranks (gloo process group for the id exchange) that both
bind the C ABI's RCCL communicator on the ONE GPU of the test box.  RCCL either refuses duplicate devices at
ncclCommInitRank -- after its bootstrap has connected the two ranks through the id, which is what this exercises:
wdx_comm_available on every rank, wdx_comm_unique_id on rank 0, the broadcast, the collective init, the error on
EVERY rank without a hang -- or accepts them, in which case the all-reduce must give the sum.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as tdist

    from warpdemux_amd import _lib, dist

    rank, local_rank, world = dist.init_process_group("gloo")
    ctx = _lib.Context(0)
    counts = torch.tensor([rank + 1, 10 * (rank + 1), 100], dtype=torch.int64, device="cuda:0")
    out = {"rank": rank, "world": world}
    try:
        red = dist.CountReducer(ctx, prefer="rccl")
        out["mode"] = red.mode
        out["rccl_ranks"] = red.rccl_ranks
        red(counts, stream=None)
        torch.cuda.synchronize()
        out["counts"] = counts.cpu().tolist()
        red.close()
    except _lib.WdxError as e:
        out["error"] = str(e)
    tdist.barrier()
    print(json.dumps(out), flush=True)
    ctx.close()
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()
