"""Randomised differential run of the small kernels against the CPU oracle (development tool, GPU box):
  * `pisa_hip_histogram_regular` / `pisa_hip_lookup_regular`: 1-3 dimensions, 1 ... 3 000 bins, samples with values ON
    edges, outside, NaN and +-inf, weights of both signs / zero / 1e+-100, averaged or summed, vector histograms;
  * `pisa_hip_metric`: the four metrics on 1 ... 5 000 bins with zeros, tiny and huge expectations, several maps summed
    first, variances for mod_chi2; negative data refused;
  * `pisa_hip_flux_2d` (Honda / Bartol tables) against the reference-style per-event spline oracle, energies 0.1 GeV - 10 TeV;
  * `pisa_hip_barr_simple` with random systematic parameters.
usage: fuzz_misc.py [trials] [seed]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import flux_oracle, oracle as orc  # noqa: E402
from pisa_amd import _lib, kernels as K  # noqa: E402
from pisa_amd.utils import flux_weights as fw  # noqa: E402
from pisa_amd.utils.resources import find_resource  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
TABLES = {}
for name in ("flux/honda-2015-spl-solmin-aa.d", "flux/bartol-2004-sno-solmax-aa.d"):
    try:
        TABLES[name] = (fw.load_2d_table(name), flux_oracle.load_2d_honda_table(find_resource(name)))
    except Exception as e:  # pylint: disable=broad-except
        print("table %s not used: %s" % (name, str(e)[:80]))


def report(tag, what):
    global bad
    bad += 1
    print("MISMATCH", tag, "|", what, flush=True)


for trial in range(trials):
    tag = "trial %d" % trial
    try:
        # ---- histogram / lookup
        dims = int(rs.randint(1, 4))
        nb = [int(rs.randint(1, 16 if dims == 3 else (60 if dims == 2 else 3000))) for _ in range(dims)]
        mins = rs.uniform(-5, 5, dims)
        maxs = mins + 10 ** rs.uniform(-2, 2, dims)
        n = int(10 ** rs.uniform(0, 4.7))
        sample = []
        for d in range(dims):
            x = rs.uniform(mins[d] - 0.2 * (maxs[d] - mins[d]), maxs[d] + 0.2 * (maxs[d] - mins[d]), n)
            edges = np.linspace(mins[d], maxs[d], nb[d] + 1)
            k = rs.rand(n)
            x = np.where(k < 0.05, edges[rs.randint(0, nb[d] + 1, n)], x)          # exactly on an edge
            x = np.where((k > 0.05) & (k < 0.06), np.nan, x)
            x = np.where((k > 0.06) & (k < 0.065), np.inf, x)
            x = np.where((k > 0.065) & (k < 0.07), -np.inf, x)
            sample.append(np.ascontiguousarray(x))
        wkind = rs.randint(4)
        w = [rs.rand(n), rs.randn(n), rs.rand(n) * (rs.rand(n) > 0.5), 10 ** rs.uniform(-100, 20, n)][wkind]     # (|w| >= 2^76 is refused loudly: the accumulator range)
        b = _lib.make_binning(list(mins), list(maxs), nb)
        cols = [K.to_device(c) for c in sample]
        want = orc.histogram_regular(sample, w, mins, maxs, nb)
        got = K.histogram_regular(cols, K.to_device(w), b).cpu().numpy()
        # sums are exact in 192-bit fixed point with an ABSOLUTE resolution of 2^-116 per deposit:
        # weights below 1e-35 vanish, as documented
        scale = max(np.abs(w).sum(), 1e-300)
        floor = n * 2.0 ** -116
        if not np.allclose(got, want, rtol=1e-12, atol=1e-15 * scale + floor):
            report(tag, "histogram dims %d bins %s n %d weights %d: max %.3e" % (dims, nb, n, wkind, np.abs(got - want).max()))
        counts = orc.histogram_regular(sample, None, mins, maxs, nb)
        if not np.array_equal(K.histogram_regular(cols, None, b).cpu().numpy(), counts):
            report(tag, "counts dims %d bins %s" % (dims, nb))
        with np.errstate(divide="ignore", invalid="ignore"):
            avg = np.nan_to_num(want / counts)
        got_avg = K.histogram_regular(cols, K.to_device(w), b, averaged=True).cpu().numpy()
        if not np.allclose(got_avg, avg, rtol=1e-12, atol=1e-15 * scale + floor):
            report(tag, "averaged histogram dims %d bins %s" % (dims, nb))
        width = int(rs.randint(1, 4))
        flat = rs.randn(int(np.prod(nb))) if width == 1 else rs.randn(int(np.prod(nb)), width)
        lk_want = orc.lookup_regular(sample, flat, mins, maxs, nb)
        lk_got = K.lookup_regular(cols, K.to_device(flat), b).cpu().numpy()
        if not np.array_equal(lk_got, lk_want):
            report(tag, "lookup dims %d bins %s width %d: %d differ" % (dims, nb, width, np.count_nonzero(lk_got != lk_want)))
        # ---- metrics
        nbins = int(10 ** rs.uniform(0, 3.7))
        n_maps = int(rs.randint(1, 5))
        exp = 10 ** rs.uniform(-3, 4, (n_maps, nbins)) * (rs.rand(n_maps, nbins) > 0.03)
        if rs.rand() < 0.2:
            exp[:, rs.randint(nbins)] = 0.0                               # a bin nobody expects anything in
        act = rs.poisson(np.clip(exp.sum(axis=0), 0, 1e6)).astype(float)
        if rs.rand() < 0.3:
            act = act + rs.rand(nbins)                                    # weighted pseudo-data
        s2 = 10 ** rs.uniform(-4, 3, (n_maps, nbins))
        for kind in ("llh", "poisson_llh", "chi2", "mod_chi2"):
            pb_want, tot_want = orc.metric(kind, act, exp.sum(axis=0), s2.sum(axis=0) if kind == "mod_chi2" else None)
            tot, pb = K.metric(kind, K.to_device(act), K.to_device(exp if n_maps > 1 else exp[0]),
                               K.to_device(s2 if n_maps > 1 else s2[0]) if kind == "mod_chi2" else None, per_bin=True)
            tot, pb = float(tot.item()), pb.cpu().numpy()
            # a bin's value is a small difference of large terms (a ln e - e - a ln a + a): the rounding of the TERMS bounds
            # the agreement of two correct fp64 evaluations (logs differ by an ulp), the floor of the LLH gates of the tests
            e_sum = np.clip(exp.sum(axis=0), 1e-10, None)
            with np.errstate(divide="ignore", invalid="ignore"):
                mag = act * np.abs(np.log(e_sum)) + e_sum + np.nan_to_num(act * np.abs(np.log(act))) + act
            ok = bool(np.all((np.abs(pb - pb_want) <= 1e-11 * np.abs(pb_want) + 8e-16 * mag) | (np.isnan(pb) & np.isnan(pb_want)))) and \
                ((np.isnan(tot) and np.isnan(tot_want)) or abs(tot - tot_want) <= 1e-11 * abs(tot_want) + 8e-16 * np.nansum(mag))
            if not ok:
                report(tag, "metric %s bins %d maps %d: total %r vs %r" % (kind, nbins, n_maps, tot, float(tot_want)))
        try:
            K.metric("llh", K.to_device(np.r_[act[:-1], -1.0] if nbins > 1 else np.array([-1.0])), K.to_device(exp.sum(axis=0)))
            report(tag, "negative data accepted")
        except ValueError:
            pass
        # ---- flux table, Barr
        m = int(rs.randint(1, 300))
        e = 10 ** rs.uniform(-1, 4, m)
        cz = np.clip(rs.uniform(-1.05, 1.05, m), -1, 1)
        for name, (table, ref) in TABLES.items():
            nu, nubar = fw.calculate_2d_flux_weights(e, cz, table)
            nu, nubar = nu.cpu().numpy(), nubar.cpu().numpy()
            sel = slice(0, min(m, 40))                                    # the oracle is a per-event Python loop
            for col, prim in ((nu[:, 0], "nue"), (nu[:, 1], "numu"), (nubar[:, 0], "nuebar"), (nubar[:, 1], "numubar")):
                want_f = flux_oracle.calculate_2d_flux_weights(e[sel], cz[sel], ref[prim])
                if not np.allclose(col[sel], want_f, rtol=1e-10, atol=0.0):
                    report(tag, "flux %s %s: max rel %.2e" % (name, prim, np.max(np.abs(col[sel] / want_f - 1))))
        nu_nom, nubar_nom = 10 ** rs.uniform(-3, 2, (m, 2)), 10 ** rs.uniform(-3, 2, (m, 2))
        ps = (float(rs.uniform(0.7, 1.3)), float(rs.uniform(0.7, 1.3)), float(rs.uniform(-0.3, 0.3)), float(rs.uniform(-2, 2)),
              float(rs.uniform(-2, 2)))
        for nubar_sign in (1, -1):
            want_b = orc.barr_simple(e, cz, nu_nom, nubar_nom, nubar_sign, *ps)
            got_b = K.barr_simple(K.to_device(e), K.to_device(cz), K.to_device(nu_nom), K.to_device(nubar_nom), nubar_sign, *ps).cpu().numpy()
            if not np.allclose(got_b, want_b, rtol=1e-12, atol=1e-300):
                report(tag, "barr nubar %d: max rel %.2e" % (nubar_sign, np.max(np.abs(got_b / want_b - 1))))
    except Exception as e:  # pylint: disable=broad-except
        bad += 1
        print("ERROR", tag, type(e).__name__, str(e)[:300], flush=True)
    if trial % 50 == 49:
        print("... %d trials, %d bad, %.0f s" % (trial + 1, bad, time.time() - t0), flush=True)
torch.cuda.synchronize()
print("fuzz_misc: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
