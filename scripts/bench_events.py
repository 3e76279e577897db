#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[1] / C2): N synthetic events,
prob3 EVENT BY EVENT (layers rebuilt per event in-kernel, all 12 containers in
one launch) + fused reweight + 10x10 (reco_energy x reco_coszen) histogram with
sumw2 + LLH.  Prints one JSON line.  Not the headline metric (bench.py is)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=float, default=1e6)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--nsi", action="store_true", help="standard-NSI matter potential (config C5)")
ap.add_argument("--decay", action="store_true", help="neutrino decay on (decay_alpha3 = 1e-4 eV^2): the decay "
                                                     "instantiation of the event kernel")
ap.add_argument("--on-device", action="store_true", help="generate the sample in HBM (C5 at its full 1e8 events)")
args = ap.parse_args()

from pisa_amd import kernels as K
from pisa_amd import synthetic

wl = synthetic.Workload(n_events=int(args.events), grid=(10, 10), out_binning="example2d", seed=0,
                        on_device=args.on_device)
st = synthetic.DeviceState(wl, osc_mode="events", compact=True)
mat_pot = None
if args.nsi:
    from pisa_amd.stages.osc.nsi_params import StdNSIParams

    n = StdNSIParams()
    n.eps_emu, n.eps_etau, n.eps_mutau = ((0.07, np.deg2rad(340)), (0.06, np.deg2rad(35)),
                                          (0.003, np.deg2rad(175)))  # numba_osc_tests.py:129-136
    mat_pot = np.diag([1.0, 0, 0]).astype(complex) + n.eps_matrix
dec = 1e-4 if args.decay else None
st.make_pseudo_data(wl.osc_params(mat_pot=mat_pot, decay_alpha3=dec), seed=0)
rs = np.random.RandomState(7)
plist = [wl.osc_params(theta23_deg=31 + 28 * rs.rand(), dm31=1e-3 + 6e-3 * rs.rand(), mat_pot=mat_pot, decay_alpha3=dec)
         for _ in range(args.warmup + args.steps)]
for p in plist[: args.warmup]:
    st.eval_host(p, "llh")
torch.cuda.synchronize()
t0 = time.perf_counter()
for p in plist[args.warmup:]:
    llh = st.eval_host(p, "llh")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
st.check_status()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for p in plist[args.warmup:]:
    st.compute_probs(p)
e1.record()
torch.cuda.synchronize()
t_osc = e0.elapsed_time(e1) / args.steps * 1e-3
# executed fp64 flops per event from the committed SQ_INSTS_VALU_*_F64 passes (scripts/profile_round.sh)
import glob
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "events_flops.json")),
             key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(os.path.dirname(f)))])
out = {
    "workload": "%d events, prob3 event-by-event (PREM-12%s) + fused reweight + 10x10 hist + LLH"
                % (wl.n_events, (", std NSI" if args.nsi else "") + (", decay" if args.decay else "")),
    "evals_per_s": 1.0 / dt, "event_evals_per_s": wl.n_events / dt, "ms_per_eval": dt * 1e3,
    "prob3_events_kernel_ms": t_osc * 1e3, "fp64_vector_peak_tflops": 78.6, "last_llh": llh}
if cal and not args.decay:
    d = json.load(open(cal[-1]))["nsi" if args.nsi else "std"]
    out["executed_fp64_flop_per_event"] = d["flop_per_event"]
    out["executed_fp64_tflops"] = d["flop_per_event"] * wl.n_events / t_osc / 1e12
    out["flop_source"] = os.path.relpath(cal[-1], ROOT)
print(json.dumps(out))
