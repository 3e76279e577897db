import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pisa_amd import synthetic, _lib
wl = synthetic.Workload(n_events=int(1e7), grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
lib = _lib.lib()
lib.pisa_hip_tail_stamps.restype = C.c_int
buf = (C.c_ulonglong * 8)()
acc = np.zeros(6)
n = 50
for i in range(n + 5):
    st.eval_host(wl.osc_params(theta23_deg=40 + 0.1 * i), "llh")
    torch.cuda.synchronize()
    lib.pisa_hip_tail_stamps(buf)
    t = np.array(list(buf)[:6], dtype=np.float64)
    if i >= 5:
        acc += (t - t[0]) * 10.0   # 100 MHz -> ns
print("phase ends, ns from kernel start: start, loads landed, converted, barrier, metric, result stored")
print(np.round(acc / n))
