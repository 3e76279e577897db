"""Development probe: cost of the tail kernel against the floor of a trivial launch."""
import torch

from pisa_amd import synthetic

wl = synthetic.Workload(n_events=120000, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
x = torch.zeros(1, device="cuda")


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def tail():
    st._maps_valid = False
    st._tail("llh", st.metric_out)


def tail_host():
    st._maps_valid = False
    st._tail("llh", st.metric_host)


print("trivial torch kernel back to back: %.2f us" % timeit(lambda: x.add_(1.0)))
print("tail kernel (device result):       %.2f us" % timeit(tail))
print("tail kernel (pinned host result):  %.2f us" % timeit(tail_host))
st.fused_tail = False
print("separate finalize + metric:        %.2f us" % timeit(tail))
