"""Randomised differential run of the evaluation engine against the CPU oracle (development tool, GPU box):
random sample sizes (down to a handful of events per container), calc grids, output binnings of 1-3 dimensions
from 1 to ~6 000 bins (LDS accumulators, LDS windows, partitioned windows), engine layouts (reference order /
compact / 16-bit indices / coordinate form, any event order, dropped unbinned events) and
oscillation parameters.  Every trial: maps and sumw2 of all containers against `oracle.pipeline_oracle.oracle_eval`
(rtol 1e-10), the LLH against the oracle's metric, kernel status clean, and the same bits from a second engine
with another event order.   usage: fuzz_engine.py [trials] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc, pipeline_oracle  # noqa: E402
from pisa_amd import synthetic  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t_start = time.time()
for trial in range(trials):
    dims = rs.randint(1, 4)
    target = int(10 ** rs.uniform(0, 3.8))                  # total bins aimed at
    per = max(1, int(round(target ** (1.0 / dims))))
    nb = [max(1, int(per * rs.uniform(0.5, 1.6))) for _ in range(dims)]
    if dims == 3:
        nb[2] = int(rs.randint(1, 4))
    lo_e, hi_e = np.log(rs.uniform(2.0, 8.0)), np.log(rs.uniform(40.0, 300.0))
    spec = dict(mins=[lo_e, -1.0, -1000.0][:dims], maxs=[hi_e, rs.choice([0.0, 1.0]), 1000.0][:dims], nbins=nb,
                log=[True, False, False][:dims])
    synthetic.BINNINGS["fuzz"] = spec
    n_events = int(12 * max(1, int(10 ** rs.uniform(0, 4.3))))
    grid = (int(rs.randint(3, 70)), int(rs.randint(3, 50)))
    wl = synthetic.Workload(n_events=n_events, grid=grid, out_binning="fuzz", seed=int(rs.randint(1 << 30)))
    form = ["reference", "compact", "compact16", "coordinate"][rs.randint(4)]
    kw = dict(sort_events=[True, False, "node", "bin", "part"][rs.randint(5)] if form != "coordinate" else True)
    if form == "coordinate":
        kw.update(indexed=False)
    elif form == "compact":
        kw.update(compact=True, index16=False)
    elif form == "compact16":
        kw.update(compact=True)
        if rs.rand() < 0.3:
            kw.update(drop_unbinned=True)
    kind = ["llh", "poisson_llh", "chi2", "mod_chi2"][rs.randint(4)]
    events_mode = form == "compact16" and "drop_unbinned" not in kw and rs.rand() < 0.25 and wl.n_events <= 24000
    decay = rs.uniform(1e-5, 1e-3) if (events_mode and rs.rand() < 0.4) else None
    if events_mode:
        kw.update(osc_mode="events")

    def point():
        return wl.osc_params(theta23_deg=rs.uniform(30, 60), dm31=rs.uniform(1e-3, 7e-3) * rs.choice([1, -1]),
                             deltacp_deg=rs.uniform(0, 360), theta13_deg=rs.uniform(5, 12), decay_alpha3=decay)

    extra = [(point(), dict(wl.last_matrices)) for _ in range(int(rs.randint(0, 4)))]     # points for one sweep
    p = point()
    mats = dict(wl.last_matrices)
    tag = "trial %d: %d events, grid %s, bins %s, %s %s" % (trial, wl.n_events, grid, nb, form, kw)
    tol_maps = 1e-9 if events_mode else 1e-10       # event-mode prob3 is contracted (fma): the prob3 tolerance
    try:
        st = synthetic.DeviceState(wl, **kw)
        st.make_pseudo_data(wl.osc_params(), seed=1)
        llh = float(st.eval(p, kind).item())
        st.check_status()
        h, s2 = (x.cpu().numpy().copy() for x in st.finalize())
        oracle_eval = pipeline_oracle.oracle_eval_events if events_mode else pipeline_oracle.oracle_eval
        ref = oracle_eval(wl, mats)
        ref_h = np.asarray(ref["hist"]).reshape(len(wl.events), -1)
        ref_s2 = np.asarray(ref["sumw2"]).reshape(len(wl.events), -1)
        ok = np.allclose(h, ref_h, rtol=tol_maps, atol=1e-3 * tol_maps * max(np.abs(ref_h).max(), 1e-300)) and \
            np.allclose(s2, ref_s2, rtol=tol_maps, atol=1e-3 * tol_maps * max(np.abs(ref_s2).max(), 1e-300))
        data = st.data.cpu().numpy()
        _, want_llh = orc.metric(kind, data.ravel(), ref_h.sum(axis=0), ref_s2.sum(axis=0) if kind == "mod_chi2" else None)
        want_llh = float(want_llh)
        terms = np.abs(data.ravel() * np.log(np.clip(ref_h.sum(axis=0), 1e-10, None))).sum() + np.abs(ref_h).sum()
        llh_ok = (np.isnan(want_llh) and np.isnan(llh)) or \
            abs(llh - want_llh) <= max(10 * tol_maps * abs(want_llh), 1e-2 * tol_maps * terms) or \
            (kind in ("chi2", "mod_chi2") and abs(llh - want_llh) <= 1e-6 * abs(want_llh))
        # another event order: the same bits
        other = synthetic.DeviceState(wl, **dict(kw, sort_events=not bool(kw["sort_events"]) if form != "coordinate" else True))
        other.set_data(data.reshape(st.data.shape))
        llh2 = float(other.eval(p, kind).item())
        h2, s22 = (x.cpu().numpy() for x in other.finalize())
        same = np.array_equal(h, h2) and np.array_equal(s2, s22) and (llh == llh2 or (np.isnan(llh) and np.isnan(llh2)))
        many_ok = True
        if extra and not events_mode and form == "compact16" and "drop_unbinned" not in kw:
            pts = [q for q, _ in extra] + [p]
            got = [float(v) for v in st.eval_many(pts, kind)]
            one_by_one = [float(st.eval(q, kind).item()) for q in pts]
            many_ok = all(a == b or (np.isnan(a) and np.isnan(b)) for a, b in zip(got, one_by_one)) and \
                (one_by_one[-1] == llh or np.isnan(llh))
        if extra and not events_mode:
            # the same points with the oscillation kernels of point k + 1 overlapping the fused kernel of point k (two streams)
            pts = [q for q, _ in extra] + [p]
            one_by_one = [float(st.eval(q, kind).item()) for q in pts]
            got = [float(v) for v in st.eval_batch(pts, kind).cpu().numpy()]
            many_ok = many_ok and all(a == b or (np.isnan(a) and np.isnan(b)) for a, b in zip(got, one_by_one))
        if not (ok and llh_ok and same and many_ok):
            bad += 1
            print("MISMATCH", tag, kind, "events-mode" if events_mode else "", "decay" if decay else "", "maps", ok, "metric", llh,
                  want_llh, llh_ok, "order-independent", same, "sweep", many_ok, flush=True)
        st.close() if hasattr(st, "close") else None
        other.close() if hasattr(other, "close") else None
    except Exception as e:  # pylint: disable=broad-except
        bad += 1
        print("ERROR", tag, type(e).__name__, str(e)[:300], flush=True)
    if trial % 20 == 19:
        print("... %d trials, %d bad, %.0f s" % (trial + 1, bad, time.time() - t_start), flush=True)
print("fuzz_engine: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
