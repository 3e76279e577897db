"""time stamps inside hist_accumulate_kernel (library built with EXTRA=-DPISA_HIST_STAMPS):
per workgroup: entry, container found + first loads issued, accumulators cleared, loop done, barrier,
flushed -- in us relative to the first workgroup's entry (wall_clock64 = 100 MHz)"""
import ctypes as C
import sys

import numpy as np
import torch

from pisa_amd import _lib, synthetic

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
binning = sys.argv[2] if len(sys.argv) > 2 else "dragon"
wl = synthetic.Workload(n_events=n, grid=(200, 100), out_binning=binning, seed=0)
st = synthetic.DeviceState(wl, compact=True)
import os
if os.environ.get("REVERSE"):   # position or identity?  (timing only: the maps come out permuted)
    st.cont.reverse()
    st._cont_arr = (_lib.Container * len(st.cont))(*st.cont)
p = wl.osc_params()
st.make_pseudo_data(p)
for _ in range(5):
    st.eval_host(p)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 1024))()
lib = _lib.lib()
lib.pisa_hip_debug_hist_stamps.argtypes = [C.c_void_p]
assert lib.pisa_hip_debug_hist_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
used = t[:, 0] > 0
t = t[used][:, :6]
t0 = t[:, 0].min()
t = (t - t0) / 100.0   # us
print("workgroups", len(t))
names = ["entry", "found+first loads", "cleared", "loop done", "barrier", "flushed"]
for k, nm in enumerate(names):
    print("%-18s min %6.2f  median %6.2f  max %6.2f" % (nm, t[:, k].min(), np.median(t[:, k]), t[:, k].max()))
d = np.diff(t, axis=1)
for k in range(5):
    print("%-18s -> %-18s median %6.2f max %6.2f" % (names[k], names[k + 1], np.median(d[:, k]), d[:, k].max()))
order = np.argsort(-t[:, 3])
print("slowest workgroups (index: loop done, flushed):", " ".join("%d:%.1f/%.1f" % (i, t[i, 3], t[i, 5]) for i in order[:24]))
print("loop done by workgroup index:", " ".join("%.0f" % x for x in t[:, 3]))
# per container (workgroups are numbered container by container)
tot = sum(int(c.n_events) for c in st.cont)
chunk = -(-tot // 512)
chunk = max(4096, min(chunk, 1 << 18))
chunk = -(-chunk // 2048) * 2048
b0 = 0
for ci, c in enumerate(st.cont):
    nb = -(-int(c.n_events) // chunk)
    tt = t[b0:b0 + nb]
    print("container %2d events %8d wgs %3d: first loads %5.2f  loop done med %6.2f max %6.2f  barrier med %6.2f max %6.2f  flushed max %6.2f"
          % (ci, c.n_events, nb, np.median(tt[:, 1]), np.median(tt[:, 3]), tt[:, 3].max(), np.median(tt[:, 4]), tt[:, 4].max(), tt[:, 5].max()))
    b0 += nb
