"""Randomised differential run of the KDE MAP chain (`utils/kde_hist.py`: oversampling, coszen reflection at either or both
ends, bin volumes, pid stacking, the batch path of the library) against `oracle/kde_oracle.py` (development tool, GPU box).
Every trial: 200 ... 8 000 events, an energy x coszen(-x pid) binning with 2-14 bins per dimension, coszen range [-1, 1] /
[-1, 0.x] / [-0.x, 1] / inside, dimension order either way, oversampling 1-5, reflection fraction 0-0.5, scott / silverman,
fixed / adaptive (alpha 0.05-0.5), weights none / positive / with zeros; the stacked form through `kde_histogramdd`, the
2-D form through `get_hist`, and two samples at once through `kde_histogramdd_batch` (bit for bit the one-by-one maps).
usage: fuzz_kde_maps.py [trials] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import kde_oracle  # noqa: E402
from pisa_amd.core.binning import MultiDimBinning, OneDimBinning  # noqa: E402
from pisa_amd.utils import kde_hist  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for trial in range(trials):
    n = int(10 ** rs.uniform(2.3, 3.9))
    ne, ncz = int(rs.randint(2, 15)), int(rs.randint(2, 15))
    e_edges = np.linspace(np.log(rs.uniform(3, 8)), np.log(rs.uniform(60, 200)), ne + 1)
    kind = rs.randint(4)
    cz_lo, cz_hi = [(-1.0, 1.0), (-1.0, rs.uniform(0.0, 0.8)), (rs.uniform(-0.8, 0.0), 1.0), (rs.uniform(-0.9, -0.2), rs.uniform(0.2, 0.9))][kind]
    cz_edges = np.linspace(cz_lo, cz_hi, ncz + 1)
    pid_edges = np.array([-1000.0, 0.0, 1000.0]) if rs.rand() < 0.7 else np.array([-3.0, 0.0, 0.5, 1000.0])
    reco_e = 10 ** rs.uniform(0.5, 2.4, n)
    reco_cz = np.clip(rs.uniform(-1, 1, n) + rs.randn(n) * 0.1, -1, 1)
    pid = rs.choice([-1.0, 0.25, 1.0], size=n)
    wmode = rs.randint(3)
    w = None if wmode == 0 else rs.rand(n) * 3 + 0.01
    if wmode == 2:
        w[rs.rand(n) < 0.2] = 0.0
    kw = dict(bw_method=["scott", "silverman"][rs.randint(2)], adaptive=bool(rs.rand() < 0.7), alpha=float(rs.uniform(0.05, 0.5)),
              coszen_reflection=float(rs.choice([0.0, 0.25, rs.uniform(0.05, 0.5)])), coszen_name="reco_coszen",
              oversample=int(rs.randint(1, 6)))
    cz_first = rs.rand() < 0.5
    if (cz_lo == -1.0 or cz_hi == 1.0) and int(ncz * kw["oversample"] * kw["coszen_reflection"]) == 0:
        # a reflecting edge with a reflection of zero points: the reference's own slicing breaks there (kde_hist.py:176-186,
        # `hist_[-0:]`), and so does this build's -- not a configuration anybody runs
        kw["coszen_reflection"] = max(0.25, 1.01 / (ncz * kw["oversample"]))
    tag = "trial %d: n %d, bins %dx%d, cz [%.2f, %.2f], pid edges %d, weights %d, %s" % (trial, n, ne, ncz, cz_lo, cz_hi, len(pid_edges), wmode, kw)
    try:
        problems = []
        d_e, d_cz = OneDimBinning("reco_energy", bin_edges=e_edges), OneDimBinning("reco_coszen", bin_edges=cz_edges)
        d_pid = OneDimBinning("pid", bin_edges=pid_edges)
        order = [d_cz, d_e, d_pid] if cz_first else [d_e, d_cz, d_pid]
        cols = dict(reco_energy=np.log(reco_e), reco_coszen=reco_cz, pid=pid)
        sample = np.stack([cols[d.name] for d in order]).T
        dims = [(d.name, d.edge_magnitudes, False) for d in order]
        got = kde_hist.kde_histogramdd(sample=sample, binning=MultiDimBinning(order), weights=w, stack_pid=True, **kw)
        want = kde_oracle.kde_histogramdd(sample, dims, w, **kw)
        if got.shape != want.shape or not np.allclose(got, want, rtol=1e-9, atol=1e-12 * max(np.abs(want).max(), 1e-300)):
            problems.append("stacked maps (worst %.2e of the peak)" % (np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)))
        s2 = sample[:, :2]
        g2 = kde_hist.get_hist(s2, binning=MultiDimBinning(order[:2]), weights=w, **kw)
        o2 = kde_oracle.get_hist(s2, dims[:2], w, kw["bw_method"], kw["adaptive"], kw["alpha"], kw["coszen_reflection"], "reco_coszen",
                                 kw["oversample"])
        g2 = g2[0] if isinstance(g2, tuple) else g2
        if not np.allclose(g2, o2, rtol=1e-9, atol=1e-12 * max(np.abs(o2).max(), 1e-300)):
            problems.append("2-D get_hist (worst %.2e of the peak)" % (np.abs(g2 - o2).max() / max(np.abs(o2).max(), 1e-300)))
        half = n // 2
        smp = [dict(sample=sample[:half], weights=None if w is None else w[:half]), dict(sample=sample[half:], weights=None if w is None else w[half:])]
        batch = kde_hist.kde_histogramdd_batch(smp, MultiDimBinning(order), stack_pid=True, **kw)
        single = [kde_hist.kde_histogramdd(sample=s["sample"], binning=MultiDimBinning(order), weights=s["weights"], stack_pid=True, **kw) for s in smp]
        if not all(np.array_equal(a, b) for a, b in zip(batch, single)):
            problems.append("batch differs from one by one")
        if problems:
            bad += 1
            print("MISMATCH", tag, "|", "; ".join(problems), flush=True)
    except Exception as e:  # pylint: disable=broad-except
        bad += 1
        print("ERROR", tag, type(e).__name__, str(e)[:300], flush=True)
        if bad <= 2:
            import traceback

            traceback.print_exc()
    if trial % 20 == 19:
        print("... %d trials, %d bad, %.0f s" % (trial + 1, bad, time.time() - t0), flush=True)
print("fuzz_kde_maps: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
