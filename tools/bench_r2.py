import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warpdemux_amd import sig_proc
from warpdemux_amd.engine import DemuxEngine
from oracle import wdx_oracle as orc
rng = np.random.default_rng(0)
for nY in (851, 2601):
    Y = rng.normal(size=(nY, 25)); eng = DemuxEngine(Y, 15, 0.1, sig_proc.SegParams(barcode_num_events=25))
    for nX in (1000, 100000):
        Xh = rng.normal(size=(nX, 25)); X = torch.from_numpy(Xh).cuda()
        d, am = eng.dtw(X); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): d, am = eng.dtw(X)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        ref = orc.dtw_matrix(Xh[:64], Y, 15, 0.1)
        ok = np.array_equal(d[:64].cpu().numpy(), ref) and np.array_equal(am[:64].cpu().numpy(), ref.argmin(1))
        print(f"nY={nY} nX={nX}: {dt*1e3:.2f} ms  {nX/dt/1e6:.3f} M reads/s  {nX*nY*515/dt/1e12:.2f} T cells/s  parity={ok}")
    eng.close()
