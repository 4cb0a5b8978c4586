import ctypes as C, sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from warpdemux_amd import _lib, sig_proc, synth
from warpdemux_amd.engine import DemuxEngine, _dp
spec = synth.SynthSpec(n_barcodes=10)
mb, a_s, a_e, _ = synth.generate_minibatch(spec, 31_000, 2000, 9000)
scale = np.float32(0.1755)
for name, data in (("adc 0.1755 pA", np.round(mb / scale).astype(np.float32) * scale), ("coarse 2 pA", np.round(mb/np.float32(2)).astype(np.float32)*np.float32(2)), ("raw", mb)):
    eng = DemuxEngine(np.zeros((10, 110)), 15, 0.1, sig_proc.SegParams(barcode_num_events=110))
    n = data.shape[0]
    sig = torch.from_numpy(np.nan_to_num(data, nan=0.0)).cuda()
    # NaN tail replaced by zeros would change semantics: keep NaN
    sig = torch.from_numpy(data).cuda()
    s_ = torch.from_numpy(a_s).cuda(); e_ = torch.from_numpy(a_e).cuda()
    status = torch.empty(n, dtype=torch.int32, device="cuda"); prof = torch.zeros((n, 32), dtype=torch.int64, device="cuda")
    pc = eng.params.to_c()
    _lib.check(eng.L.wdx_fingerprint_profile_dev(eng.ctx.handle, _dp(sig), None, 9000, 9000, n, _dp(s_), _dp(e_), C.byref(pc), _dp(status), _dp(prof), n, 1, 0, None))
    torch.cuda.synchronize()
    p = prof.cpu().numpy()
    dec = p[p[:, 9] == 0]
    print(name, "declined", int(p[0, 15]), "of", n, "reasons", np.bincount(dec[:, 13].astype(int), minlength=7).tolist(), "status", np.bincount(status.cpu().numpy(), minlength=6).tolist())
    fin = p[p[:, 9] != 0]
    print("   score mode (1 = approximate keys, 2 = exact scores):", np.bincount(fin[:, 14].astype(int), minlength=3).tolist())
    eng.close()
