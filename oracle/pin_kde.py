"""One-command pin of the KDE core (TEST INFRASTRUCTURE ONLY; round-5 verdict "Next #4").

The reference's estimator is the un-vendored `kde` package (pisa/utils/kde_hist.py:8: `from kde.cudakde import
gaussian_kde, bootstrap_kde`; setup.py:88: git+https://github.com/icecubeopensource/kde.git, no version pin).  It is
absent from this image and from /root/reference, so the KDE core of this build is PARITY UNPINNED.  This script turns
"unpinnable here" into "pinnable in one command elsewhere":

    pip install git+https://github.com/icecubeopensource/kde.git      # wherever there is a network
    python -m oracle.pin_kde                                           # from the root of this repository

evaluates `kde.cudakde.gaussian_kde` (CPU path, `use_cuda=False`) with the reference's call contract
(kde_hist.py:110-120, 154-160: `gaussian_kde(x, weights=, bw_method=, adaptive=, alpha=, use_cuda=)(points)`) on

  (i)  the reference test's exact set-up (pisa_tests/test_kde_stage.py:45-153: toy_event_generator(n_events=1000,
       seed=0) -> aeff.weight(livetime 12345 s) -> utils.kde(silverman, adaptive, alpha = 0.1, oversample 1,
       coszen_reflection 0.25) on 15 log-energy x 16 coszen bins), with and without `linearize_log_dims`;
  (ii) three small adaptive 2-D cases (equal weights / unequal weights / strong adaptation)

and writes `tests/golden/kde_ref.npz`: per case the inputs (sample, weights, points, settings), the package's
densities and -- where the object exposes them -- its local bandwidth factors.  With the fixture present,
`tests/test_oracle.py::test_kde_pinned_by_the_reference_package` (oracle) and
`tests/test_gpu_kde.py::test_kde_pinned_by_the_reference_package` (device) compare at 1e-10 relative; without it they
skip and the strict xfail of `tests/test_gpu_kde_stage.py` keeps saying why.  Nothing but numpy is needed besides the
package; the cases are generated here from fixed seeds (no file of this repository is read).

`python -m oracle.pin_kde --self-check` runs the same cases through this build's ORACLE instead of the package and
writes nothing into tests/golden: it only proves that the script itself runs (used by the CPU test suite).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "kde_ref.npz")


def _reference_test_case(linearize):
    """x [2, n] in (coszen, energy) order -- kde_hist.py:95-104 swaps coszen to the front --, weights, points"""
    rs = np.random.RandomState(0)                        # toy_event_generator.py:75-76
    n = 1000
    energy = np.power(10, rs.rand(n) * 3)
    coszen = rs.rand(n) * 2 - 1
    weights = np.full(n, 12345.0)                        # aeff.weight: livetime 12345 s x weight_scale 1
    e_edges = np.logspace(1, 2, 16)                      # test_kde_stage.py:45-60: 15 log bins 10..100 GeV
    cz_edges = np.linspace(-1.0, 0.0, 17)                # 16 lin bins -1..0
    if linearize:                                        # stages/utils/kde.py:106-130: ln-space, lin binning
        x_e, c_e = np.log(energy), 0.5 * (np.log(e_edges)[:-1] + np.log(e_edges)[1:])
    else:                                                # log dim: weighted_centers = geometric centres
        x_e, c_e = energy, np.sqrt(e_edges[:-1] * e_edges[1:])
    c_cz = 0.5 * (cz_edges[:-1] + cz_edges[1:])
    refl = int(len(c_cz) * 0.25)                         # kde_hist.py:127-141: lower edge is -1 -> reflect below
    c_cz = np.concatenate([2 * c_cz[0] - c_cz[1:refl + 1][::-1], c_cz])
    grid = np.meshgrid(c_cz, c_e, indexing="ij")
    points = np.array([g.ravel() for g in grid])
    return dict(x=np.array([coszen, x_e]), w=weights, points=points, bw_method="silverman", adaptive=True, alpha=0.1)


def _small_case(seed, n, unequal, alpha, bw_method):
    rs = np.random.RandomState(seed)
    a = rs.randn(n)
    x = np.array([a + 0.3 * rs.randn(n), 0.5 * a + rs.rand(n) * 2.0])       # correlated: a full covariance matters
    w = (0.2 + rs.rand(n) * 3.0) if unequal else np.ones(n)
    g = np.meshgrid(np.linspace(x[0].min(), x[0].max(), 9), np.linspace(x[1].min(), x[1].max(), 7), indexing="ij")
    return dict(x=x, w=w, points=np.array([v.ravel() for v in g]), bw_method=bw_method, adaptive=True, alpha=alpha)


def cases():
    return {
        "ref_test_linearized": _reference_test_case(True),
        "ref_test_not_linearized": _reference_test_case(False),
        "small_equal_weights": _small_case(11, 200, False, 0.3, "scott"),
        "small_unequal_weights": _small_case(12, 300, True, 0.1, "silverman"),
        "small_strong_adaptation": _small_case(13, 150, True, 0.5, "silverman"),
    }


def _package_eval(c):
    from kde.cudakde import gaussian_kde      # the import of pisa/utils/kde_hist.py:8

    est = gaussian_kde(c["x"], weights=np.nan_to_num(c["w"]), bw_method=c["bw_method"], adaptive=c["adaptive"],
                       alpha=c["alpha"], use_cuda=False)
    dens = np.asarray(est(c["points"]), dtype=np.float64)
    lam = None
    for name in ("lambdas", "inv_loc_bw", "local_bandwidths"):    # whatever the installed version calls them
        if getattr(est, name, None) is not None:
            lam = (name, np.asarray(getattr(est, name), dtype=np.float64))
            break
    return dens, lam


def _oracle_eval(c):
    from oracle import kde_oracle

    return kde_oracle.gaussian_kde_eval(c["x"], c["w"], c["points"], c["bw_method"], c["adaptive"], c["alpha"]), None


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--self-check", action="store_true", help="run the cases through this build's oracle; write nothing under tests/")
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    if not args.self_check:
        try:
            import kde.cudakde  # noqa: F401
        except ImportError:
            print("pin_kde: the `kde` package is not installed here (pip install git+https://github.com/icecubeopensource/kde.git); "
                  "nothing written -- the KDE core stays unpinned", file=sys.stderr)
            return 2
    out = {}
    for name, c in cases().items():
        dens, lam = (_oracle_eval if args.self_check else _package_eval)(c)
        assert dens.shape == (c["points"].shape[1],) and np.all(np.isfinite(dens))
        out[name + "__x"], out[name + "__w"], out[name + "__points"] = c["x"], c["w"], c["points"]
        out[name + "__settings"] = np.array([1.0 if c["bw_method"] == "silverman" else 0.0, float(c["adaptive"]), c["alpha"]])
        out[name + "__density"] = dens
        if lam is not None:
            out[name + "__lambdas"] = lam[1]
        print("%-26s n = %4d, %3d points, density sum %.12g%s" % (name, c["x"].shape[1], dens.size, dens.sum(),
                                                                  "" if lam is None else ", %s kept" % lam[0]))
    path = args.out or (None if args.self_check else FIXTURE)
    if path:
        np.savez_compressed(path, **out)
        print("written:", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
