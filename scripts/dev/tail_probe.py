"""tail kernel (limbs -> maps -> metric, one workgroup) against the number of accumulators it converts:
128 bins x n containers; limbs filled with plausible sums; cleared limbs refilled by a copy each round (timed separately)"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pisa_amd import kernels as K

nb = 128
for nc in (12, 8, 4, 2, 1):
    ws = K.HistWorkspace(nc, nb)
    rs = np.random.RandomState(0)
    fill = torch.from_numpy(rs.randint(0, 2 ** 40, size=tuple(ws.limbs.shape)).astype(np.int64)).cuda()
    fill[..., 0] = 0; fill[..., 5] = 0
    data = torch.from_numpy(rs.poisson(100.0, nb).astype(np.float64)).cuda()
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    mst = torch.zeros(1, dtype=torch.int32, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for rep in range(60):
        ws.limbs.copy_(fill)
        torch.cuda.synchronize()
        ev[0].record()
        K.finalize_metric(ws, "llh", data, out, mst, clear_limbs=True)
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    ts = sorted(ts[10:])
    print("containers %2d (%4d accumulators): median %.2f us  min %.2f" % (nc, nc * nb * 2, ts[len(ts) // 2], ts[0]))
