"""Development probe (not part of the product): time the fused kernel under a
few variants on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pisa_amd import synthetic

n_events = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
wl = synthetic.Workload(n_events=int(n_events), grid=(200, 100), out_binning="dragon", seed=0)
p = wl.osc_params()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for packed, srt in ((True, True),):
    st = synthetic.DeviceState(wl, packed=packed, sort_events=srt)
    st.eval(p)
    for blocks, threads, copies in ((1024, 256, 1), (1024, 256, 2), (1024, 256, 4), (768, 512, 4), (512, 1024, 4),
                                    (768, 1024, 4), (768, 1024, 2), (1536, 256, 4), (2048, 256, 4)):
        os.environ["PISA_HIP_HIST_BLOCKS"] = str(blocks)
        os.environ["PISA_HIP_HIST_THREADS"] = str(threads)
        os.environ["PISA_HIP_HIST_COPIES"] = str(copies)
        res = []
        for dbg in ("0", "4", "2"):
            os.environ["PISA_HIP_HIST_DBG"] = dbg
            res.append(timeit(lambda: st.accumulate()))
        print("sorted=%d packed=%d copies=%d blocks=%4d threads=%4d : full %.1f us | no-flush %.1f | no-atomics %.1f"
              % (srt, packed, copies, blocks, threads, *res))
    del st
