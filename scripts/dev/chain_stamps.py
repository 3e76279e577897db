"""time stamps inside prob3_chain_kernel (library built with EXTRA=-DPISA_CHAIN_STAMPS; sign 0, energy tile 0):
per wavefront: entry, set-up done, own steps done, barrier passed, join done, final product done, stored -- in us
relative to the first wavefront's entry (wall_clock64 = 100 MHz)"""
import ctypes as C
import sys

import numpy as np
import torch

from pisa_amd import _lib, synthetic

wl = synthetic.Workload(n_events=1_000_000, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True)
p = wl.osc_params()
st.make_pseudo_data(p)
for _ in range(5):
    st.eval_host(p)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 4096))()
lib = _lib.lib()
lib.pisa_hip_debug_chain_stamps.argtypes = [C.c_void_p]
assert lib.pisa_hip_debug_chain_stamps(buf) == 0
raw = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)
used = raw[:, 0] > 0
raw = raw[used]
info = raw[:, 7]
cnt = (info & 0xffff).astype(int); g = ((info >> 16) & 0xff).astype(int); Gr = ((info >> 24) & 0xff).astype(int)
steps = ((info >> 32) & 0xffff).astype(int); live = ((info >> 48) & 1).astype(int)
t = raw[:, :7].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0
t[raw[:, :7] == 0] = np.nan
print("wavefronts", len(t), "live", int(live.sum()), "kernel span %.2f us" % np.nanmax(t))
names = ["entry", "set-up", "steps done", "barrier", "join", "final", "stored"]
for sel, lab in ((live == 1, "all live"), ((live == 1) & (Gr == 4) & (g == 0), "leaders of 4-wave rows"), ((live == 1) & (Gr == 4) & (g > 0), "helpers of 4-wave rows"),
                 ((live == 1) & (Gr == 1), "1-wave rows")):
    if sel.sum() == 0:
        continue
    print("== %s: %d wavefronts, layers %d..%d, own steps %d..%d" % (lab, sel.sum(), cnt[sel].min(), cnt[sel].max(), steps[sel].min(), steps[sel].max()))
    for k, nm in enumerate(names):
        col = t[sel][:, k]
        if np.all(np.isnan(col)):
            continue
        print("   %-12s min %6.2f  median %6.2f  max %6.2f" % (nm, np.nanmin(col), np.nanmedian(col), np.nanmax(col)))
# the slowest wavefronts
order = np.argsort(-np.nan_to_num(np.nanmax(t, axis=1)))[:6]
for i in order:
    print("slow: layers %2d g %d/%d steps %d : %s" % (cnt[i], g[i], Gr[i], steps[i], " ".join("%6.2f" % v for v in t[i])))
