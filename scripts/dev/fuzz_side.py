"""The kernels of the services around the path (`csrc/stages.hip`, the wide metrics of `pisa_hip_metric`) against
`oracle/stages_oracle.py` on random shapes and values (development tool, GPU box): sizes 0 .. 3e5 (not multiples of the
workgroup), 1-3 binning dimensions with 1-40 bins, coordinates on edges / outside / NaN, parameters of either sign,
empty and degenerate inputs.  usage: fuzz_side.py [trials] [seed]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import stages_oracle as so  # noqa: E402
from pisa_amd import kernels as K  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def dev(a):
    return K.to_device(np.ascontiguousarray(a, dtype=np.float64))


def close(got, want, rtol, atol=0.0):
    return got.shape == want.shape and np.allclose(got, want, rtol=rtol, atol=atol, equal_nan=True)


for trial in range(trials):
    n = int(rs.choice([0, 1, 255, 256, 257, int(10 ** rs.uniform(0, 5.5))]))
    problems = []
    try:
        e, cz = 10 ** (rs.rand(n) * 5 - 1), rs.rand(n) * 2 - 1
        w0, flux = rs.rand(n) * 10 ** rs.uniform(-3, 3), rs.rand(n, 2)
        # lookup_indices
        nd = int(rs.randint(1, 4))
        edges = [np.sort(np.unique(rs.uniform(-2, 2, int(rs.randint(2, 42))))) for _ in range(nd)]
        edges = [ed if len(ed) >= 2 else np.array([-1.0, 1.0]) for ed in edges]
        cols = [rs.uniform(-2.5, 2.5, n) for _ in range(nd)]
        for c, ed in zip(cols, edges):
            k = min(len(ed), n)
            c[:k] = ed[:k]
            if n > k:
                c[k] = np.nan
        got = K.lookup_indices([dev(c) for c in cols], [dev(x) for x in edges]).cpu().numpy()
        if not np.array_equal(got, so.lookup_indices(cols, edges).astype(np.int64).reshape(got.shape)):
            problems.append("lookup_indices")
        # two_nu_osc
        t23, dm31 = rs.uniform(0, 1.6), rs.uniform(-4e-3, 4e-3)
        for flav in (0, 1, 2):
            w = dev(w0)
            K.two_nu_osc(dev(flux), t23, dm31, dev(e), dev(cz), flav, w)
            if not close(w.cpu().numpy(), so.two_nu_weights(flux, t23, dm31, e, cz, flav, w0), 1e-9, 1e-12 * (w0.max() if n else 1)):
                problems.append("two_nu_osc flav %d" % flav)
        # power_law / shift_toward
        idx, norm, piv = rs.uniform(-4, 2), 10 ** rs.uniform(-20, 3), 10 ** rs.uniform(0, 5)
        if not close(K.power_law(dev(e), piv, idx, norm, nominal=dev(w0)).cpu().numpy(), so.power_law(e, piv, idx, norm, w0), 1e-13):
            problems.append("power_law")
        frac = rs.uniform(-0.5, 1.5)
        if not np.array_equal(K.shift_toward(dev(cz), dev(w0), frac, clip=(-1, 1)).cpu().numpy(), so.shift_toward(cz, w0, frac, (-1, 1))):
            problems.append("shift_toward")
        # poly_scale / column_combination
        k = int(rs.randint(0, 9))
        lin, quad, ps = [rs.randn(n) for _ in range(k)], [rs.randn(n) for _ in range(k)], rs.randn(k) * 2
        scale = rs.uniform(0, 3)
        w = dev(w0)
        K.poly_scale([dev(a) for a in lin], [dev(a) for a in quad], ps, w, scale=scale)
        factor = np.ones(n)
        for p, a, q in zip(ps, lin, quad):
            factor = factor * (1.0 + (a + q * p) * p)
        if not np.array_equal(w.cpu().numpy(), w0 * np.maximum(0, factor * scale)):
            problems.append("poly_scale k=%d" % k)
        kc = int(rs.randint(0, 65))
        gcols, coef = [rs.randn(n) * 0.1 for _ in range(kc)], rs.randn(kc)
        acc = np.zeros(n)
        for c_, g in zip(coef, gcols):
            acc += c_ * g
        if not close(K.column_combination([dev(g) for g in gcols], coef, n, "exp").cpu().numpy(), np.exp(acc), 4e-15):
            problems.append("column_combination exp k=%d" % kc)
        if not np.array_equal(K.column_combination([dev(g) for g in gcols], coef, n, "one_plus").cpu().numpy(), 1 + acc):
            problems.append("column_combination one_plus")
        # interp_linear inside the knots
        xk = np.sort(np.unique(rs.uniform(0, 1, int(rs.randint(2, 30)))))
        if len(xk) >= 2:
            yk = rs.randn(len(xk))
            x = rs.uniform(xk[0], xk[-1], n)
            x[:min(n, len(xk))] = xk[:min(n, len(xk))]
            if not np.array_equal(K.interp_linear(dev(xk), dev(yk), dev(x)).cpu().numpy(), np.interp(x, xk, yk)):
                problems.append("interp_linear")
        # decoherence
        th = rs.uniform(0, np.pi / 2, 3)
        u2 = so.tau_row_sq(*th)
        coef3 = [u2[1] * u2[0], u2[2] * u2[0], u2[2] * u2[1]]
        gam, dl = 10 ** rs.uniform(-25, -20, 3), rs.uniform(-3e-3, 3e-3, 3)
        length = rs.uniform(1, 12800, n)
        got = K.decoherence_probs(coef3, gam, dl, False, dev(e), dev(length)).cpu().numpy()
        if not close(got, so.decoherence_table(so.decoherence_disappearance(coef3, gam, dl, e, length)), 1e-9, 1e-12):
            problems.append("decoherence")
        # wide metrics on a map of n bins (n <= 3e5 goes through both reductions)
        nb = min(n, 20000)
        if nb:
            lam = rs.rand(nb) * 10 ** rs.uniform(-2, 3)
            sig = np.sqrt(lam) * rs.rand(nb) * (rs.rand(nb) > 0.1)
            kk = rs.poisson(np.minimum(lam, 1e6)).astype(np.float64)
            for kind in ("correct_chi2", "signed_sqrt_mod_chi2", "mcllh_mean", "mcllh_eff") + (("conv_llh",) if nb <= 3000 else ()):
                total, pb = K.metric(kind, dev(kk), dev(lam), dev(sig ** 2), per_bin=True)
                want = so.metric_wide(kind, kk, lam, sig)
                with np.errstate(all="ignore"):
                    alpha = np.where(sig > 0, np.maximum(lam, 1e-10) ** 2 / np.maximum(sig, 1e-300) ** 2, 0.0)
                size = 1.0 + ((kk + alpha) * (1.0 + np.abs(np.log(np.maximum(kk + alpha, 1e-300)))) if kind.startswith("mcllh") else 0.0)
                got = pb.cpu().numpy()
                if not np.all(np.abs(got - want) <= 1e-11 * size + 1e-10 * np.abs(want)):
                    problems.append("%s: %.3g" % (kind, np.abs(got - want).max()))
                if abs(float(total.item()) - np.nansum(got)) > 1e-12 * np.abs(got).sum() + 1e-300:
                    problems.append("%s total" % kind)
    except Exception as err:  # pylint: disable=broad-except
        problems.append("%s %s" % (type(err).__name__, str(err)[:200]))
    if problems:
        bad += 1
        print("MISMATCH trial %d (n %d): %s" % (trial, n, "; ".join(problems)), flush=True)
print("fuzz_side: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
