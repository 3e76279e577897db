"""host time of the three C-ABI calls of one headline evaluation (they are asynchronous: this is what the host spends
issuing them) and of the Python around them"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from pisa_amd import synthetic, _lib, kernels as K

wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True)
p = wl.osc_params()
st.make_pseudo_data(p)
for _ in range(20):
    st.eval_host(p)
a = st._lean
lib, s = a["lib"], K._stream()
T = np.zeros(5)
N = 300
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.pisa_hip_prob3_grid_planned(C.byref(p), a["plan"], a["energy"], a["n_e"], a["e_major"], None, None, a["pepmu"], s)
    t1 = time.perf_counter()
    rc |= lib.pisa_hip_reweight_hist_acc(a["cont"], a["n_cont"], a["grid"], a["nu"], a["nubar"], a["pepmu"], a["outb"], a["limbs"], a["status"], s)
    t2 = time.perf_counter()
    st._metric_host_np[:] = np.nan
    rc |= lib.pisa_hip_finalize_metric_split(a["limbs"], 1, a["n_cont"], st.n_bins, a["hist"], a["sumw2"], 0, a["data"], None, 0, None, a["out"], a["status"], a["mstatus"], 1, s)
    t3 = time.perf_counter()
    v = st._poll_split()
    t4 = time.perf_counter()
    assert rc == 0
    T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]
T = T / N * 1e6
print("prob3_grid_planned %.2f us, reweight_hist_acc %.2f us, finalize_metric_split %.2f us, poll until the value %.2f us, total %.2f us" % tuple(T))
t0 = time.perf_counter()
for _ in range(N):
    st.eval_host(p)
print("eval_host %.2f us per evaluation" % ((time.perf_counter() - t0) / N * 1e6))
