"""Round-6 experiment R6-3: the headline step with the accumulate kernel on a side stream, resident and polling the chain
kernel's hand-over counters (development library, PISA_HIP_EVAL_OVERLAP = 0 off / 1 oscillation launches first /
2 accumulate launch first).  One mode per process (the switch is read when the evaluator is made); prints one JSON
line: us per evaluation (median of blocks), the LLH bits of the last point, a digest of the maps, the status word.

    PISA_HIP_LIB=pisa_amd/libpisa_hip_dev.so PISA_HIP_EVAL_OVERLAP=1 python3 scripts/dev/overlap_ab.py [blocks] [evals per block]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from pisa_amd import synthetic  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 500
wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
pts = bench.param_list(wl, 20 + per)
for p in pts[:20]:
    st.eval_host(p, "llh")
ts = []
for _ in range(blocks):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in pts[20:]:
        v = st.eval_host(p, "llh")
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / per * 1e6)
st.check_status()
h, s2 = st.maps()
print(json.dumps({"mode": os.environ.get("PISA_HIP_EVAL_OVERLAP", "0"), "us_per_eval": float(np.median(ts)), "blocks": [round(t, 2) for t in ts],
                  "llh": v, "llh_bits": "%016x" % int(np.float64(v).view(np.int64)),
                  "maps_sha": hashlib.sha256(np.ascontiguousarray(h).tobytes() + np.ascontiguousarray(s2).tobytes()).hexdigest()[:16],
                  "evaluations": blocks * per + 20}))
