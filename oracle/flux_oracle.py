"""CPU restatement of the reference's 2-D Honda flux interpolation.

TEST INFRASTRUCTURE (oracle): only tests/, smoke() and bench.py's cpu_baseline
may import this.  Pinned by tests/golden/flux_ref.npz, which was produced by the
reference's own code (oracle/gen_golden.py:gen_flux).

Follows pisa/utils/flux_weights.py:
  load_2d_honda_table  (:50-131)  azimuth-averaged table, 20 coszen bands x 101
      energies; per band the running integral over log10(E) of flux*E^enpow
      (step 0.05) at 102 knots linspace(-1.025, 4.025), interpolated by a cubic
      spline (scipy splrep, s=0)
  calculate_2d_flux_weights (:267-349)  per event: the 20 band splines'
      derivatives at log10(E) -> running sum * 0.1 at the 21 coszen knots
      linspace(-1, 1) -> cubic spline -> its derivative at coszen, / E^enpow
("integral preserving": the interpolant's bin integrals reproduce the table).
The spline routines are scipy's FITPACK wrappers, as in the reference.
"""
import numpy as np
from scipy import interpolate

PRIMARIES = ("numu", "numubar", "nue", "nuebar")  # column order of the table (:45)
N_CZ = 20


def load_2d_honda_table(path, enpow=1):
    """-> {primary: [20 tck tuples, ascending coszen band centre -0.95 .. 0.95]}"""
    table = np.genfromtxt(path, usecols=range(1 + len(PRIMARIES)))
    keep = ~np.all(np.isnan(table) | (table == 0), axis=1)  # header lines (:66-67)
    table = table[keep].T
    energy = np.split(table[0], N_CZ)[0]
    log_knots = np.linspace(-1.025, 4.025, 102)
    out = {}
    for k, prim in enumerate(PRIMARIES):
        bands = np.split(table[1 + k], N_CZ)  # file order: cos(zenith) 0.95 first (:107)
        splines = []
        for band in bands:
            integral = np.concatenate(([0.0], np.cumsum(band * np.power(energy, enpow) * 0.05)))
            # np.cumsum adds left to right like the reference's running total (:99-103)
            splines.append(interpolate.splrep(log_knots, integral, s=0))
        out[prim] = splines[::-1]
    return out


def calculate_2d_flux_weights(true_energies, true_coszens, band_splines, enpow=1):
    e = np.asarray(true_energies, dtype=np.float64)
    cz = np.asarray(true_coszens, dtype=np.float64)
    if not ((cz >= -1.0).all() and (cz <= 1.0).all()):
        raise ValueError("Not all coszens found between -1 and 1")
    cz_knots = np.linspace(-1, 1, N_CZ + 1)
    out = np.empty_like(e)
    for i in range(e.size):
        vals = np.zeros(N_CZ + 1)
        for j in range(N_CZ):
            vals[j + 1] = interpolate.splev(np.log10(e[i]), band_splines[j], der=1)
        spline = interpolate.splrep(cz_knots, np.cumsum(vals) * 0.1, s=0)
        out[i] = interpolate.splev(cz[i], spline, der=1) / np.power(e[i], enpow)
    return out


def grid_flux(node_energies, node_coszens, band_splines, enpow=1):
    """calculate_2d_flux_weights on the nodes of an (E x coszen) grid -> [n_E, n_cz].
    All nodes of one energy share the coszen spline, so it is fitted once per energy;
    every node value is the same expression as in the per-event function above."""
    cz_knots = np.linspace(-1, 1, N_CZ + 1)
    cz = np.asarray(node_coszens, dtype=np.float64)
    out = np.empty((len(node_energies), len(cz)))
    for i, e in enumerate(np.asarray(node_energies, dtype=np.float64)):
        vals = np.zeros(N_CZ + 1)
        for j in range(N_CZ):
            vals[j + 1] = interpolate.splev(np.log10(e), band_splines[j], der=1)
        spline = interpolate.splrep(cz_knots, np.cumsum(vals) * 0.1, s=0)
        out[i] = interpolate.splev(cz, spline, der=1) / np.power(e, enpow)
    return out
