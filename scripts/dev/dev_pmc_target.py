"""Target for PMC passes: a few launches of the fused kernel on the default workload
(PISA_LDS_ORDER=0/1 selects the second-level event order)."""
import os

import torch

from pisa_amd import synthetic

wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True, lds_order=os.environ.get("PISA_LDS_ORDER", "1") == "1")
st.compute_probs(wl.osc_params())
for _ in range(4):
    st.accumulate()
torch.cuda.synchronize()
