"""One small invocation of the whole hot path on cuda:0, checked against the
CPU oracle (the oracle is only the checker here)."""
import numpy as np


def run():
    import torch

    assert torch.cuda.is_available(), "smoke() needs a HIP device"
    torch.cuda.set_device(0)
    from pisa_amd import kernels as K
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=24000, grid=(24, 16), out_binning="dragon", seed=11)
    st = synthetic.DeviceState(wl)
    p = wl.osc_params(theta23_deg=45.0, dm31=2.4e-3)
    data = st.make_pseudo_data(p, seed=1)
    llh = float(st.eval(p, "llh").item())
    st.check_status()
    hist, sumw2 = st.maps()

    from oracle import oracle as orc
    from oracle.pipeline_oracle import oracle_eval

    ref = oracle_eval(wl)
    np.testing.assert_allclose(st.prob_nu.cpu().numpy(), ref["prob_nu"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(hist, ref["hist"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(sumw2, ref["sumw2"], rtol=1e-12, atol=1e-300)
    _, want = orc.metric("llh", data, ref["hist"].sum(axis=0))
    np.testing.assert_allclose(llh, want, rtol=1e-10)
    print("smoke OK: llh=%.12g (oracle %.12g), %d events, %d bins" %
          (llh, want, wl.n_events, wl.n_bins))
