"""Atmospheric flux tables (counterpart of pisa/utils/flux_weights.py, 2-D
azimuth-averaged Honda tables: `load_2d_table` :205-264, `calculate_2d_flux_weights`
:267-349).

The table is read and turned into the reference's integral-preserving band splines
on the host with the same scipy/FITPACK calls (setup time, 80 small splines); the
per-event work -- which the reference does in a Python loop with a fresh `splrep`
per event -- runs on the GPU (`pisa_hip_flux_2d`, csrc/flux.hip).
"""
import numpy as np
from scipy import interpolate

from pisa_amd import _lib
from pisa_amd import kernels as K
from pisa_amd.utils.resources import find_resource

__all__ = ["PRIMARIES", "FluxTable2D", "load_2d_table", "calculate_2d_flux_weights"]

PRIMARIES = ["numu", "numubar", "nue", "nuebar"]  # column order of the tables (:45)
_DEVICE_ORDER = ("nue", "numu", "nuebar", "numubar")  # (nu_flux[:, 0:2], nubar_flux[:, 0:2])
N_CZ = 20


class FluxTable2D(dict):
    """`spline_dict` of the reference ({primary: {'%.2f' % coszen: tck}, 'name': ...})
    plus the device-resident form of the same splines."""

    def __init__(self, spline_dict, enpow):
        super().__init__(spline_dict)
        self.enpow = int(enpow)
        cz_keys = ["%.2f" % x for x in np.linspace(-0.95, 0.95, N_CZ)]  # (:330)
        t_e = None
        coef = np.empty((4, N_CZ, 0))
        rows = []
        for prim in _DEVICE_ORDER:
            for key in cz_keys:
                t, c, k = self[prim][key]
                assert k == 3
                if t_e is None:
                    t_e = np.asarray(t, dtype=np.float64)
                assert np.array_equal(t_e, t)
                rows.append(np.asarray(c, dtype=np.float64)[: len(t) - 4])
        coef = np.array(rows).reshape(4, N_CZ, len(t_e) - 4)
        # interpolation through fixed knots is linear in the data: the per-event
        # splrep (:344) equals Cmat . y with the cardinal splines' coefficients
        cz_knots = np.linspace(-1, 1, N_CZ + 1)
        t_cz = None
        cols = []
        for j in range(N_CZ + 1):
            unit = np.zeros(N_CZ + 1)
            unit[j] = 1.0
            t, c, _ = interpolate.splrep(cz_knots, unit, s=0)
            t_cz = np.asarray(t, dtype=np.float64)
            cols.append(np.asarray(c, dtype=np.float64)[: len(t) - 4])
        cmat = np.ascontiguousarray(np.array(cols).T)  # [coefficient][knot]
        self._dev = [K.to_device(a) for a in (t_e, coef, t_cz, cmat)]
        s = _lib.FluxTable()
        s.n_bands, s.n_knots_e, s.n_knots_cz, s.enpow = N_CZ, len(t_e), len(t_cz), self.enpow
        s.d_knots_e, s.d_coef_e = self._dev[0].data_ptr(), self._dev[1].data_ptr()
        s.d_knots_cz, s.d_cardinal = self._dev[2].data_ptr(), self._dev[3].data_ptr()
        s.cz_step = 0.1  # (:343)
        self.struct = s


def _load_2d_bartol_table(flux_file, enpow=1):
    """flux_weights.py:133-203: Bartol tables rewritten in the Honda layout -- 20 coszen bands, energy
    steps of 0.05 in log10 below 10 GeV and 0.1 above, hence the two-part knot vector and bin widths"""
    cols = ["energy"] + PRIMARIES
    with open(find_resource(flux_file)) as fh:
        table = np.genfromtxt(fh, usecols=list(range(len(cols))))
    header = np.all(np.isnan(table) | np.equal(table, 0), axis=1)
    table = table[~header].T
    flux = {c: np.array(np.split(col, N_CZ)) for c, col in zip(cols, table)}
    energy = flux["energy"][0]
    log_knots = np.concatenate([np.linspace(-1, 1, 41), np.linspace(1.1, 4, 30)])
    spline_dict = {}
    for prim in PRIMARIES:
        splines = {}
        for band_no, band in enumerate(flux[prim], start=1):
            running, integral = 0.0, [0.0]
            for val, e in zip(band, energy):
                running += val * np.power(e, enpow) * (0.05 if e < 10.0 else 0.1)
                integral.append(running)
            splines["%.2f" % (1.05 - band_no * 0.1)] = interpolate.splrep(log_knots, integral, s=0)
        spline_dict[prim] = splines
    return spline_dict


def _load_2d_honda_table(flux_file, enpow=1):
    """flux_weights.py:50-131 (Honda layout: 20 coszen bands of 101 energies)"""
    cols = ["energy"] + PRIMARIES
    with open(find_resource(flux_file)) as fh:
        table = np.genfromtxt(fh, usecols=list(range(len(cols))))
    header = np.all(np.isnan(table) | np.equal(table, 0), axis=1)
    table = table[~header].T
    flux = {c: np.array(np.split(col, N_CZ)) for c, col in zip(cols, table)}
    energy = flux["energy"][0]
    log_knots = np.linspace(-1.025, 4.025, 102)
    spline_dict = {}
    for prim in PRIMARIES:
        splines = {}
        for band_no, band in enumerate(flux[prim], start=1):
            running, integral = 0.0, [0.0]
            for val, e in zip(band, energy):
                running += val * np.power(e, enpow) * 0.05
                integral.append(running)
            splines["%.2f" % (1.05 - band_no * 0.1)] = interpolate.splrep(log_knots, integral, s=0)
        spline_dict[prim] = splines
    return spline_dict


def load_2d_table(flux_file, enpow=1, return_table=False):
    if not isinstance(enpow, int):
        raise TypeError("Energy power must be an integer")
    if not isinstance(flux_file, str):
        raise TypeError("Flux file name must be a string")
    if return_table:
        raise NotImplementedError("return_table is a plotting aid of the reference, not part of this build")
    if "aa" not in flux_file:
        raise ValueError("Azimuth-averaged tables are expected")
    # (:229-260: the group is read off the file name)
    if "honda" in flux_file:
        spline_dict = _load_2d_honda_table(flux_file, enpow=enpow)
        spline_dict["name"] = "honda"
    elif "bartol" in flux_file:
        spline_dict = _load_2d_bartol_table(flux_file, enpow=enpow)
        spline_dict["name"] = "bartol"
    elif "hillas" in flux_file:
        raise NotImplementedError("Hillas-Gaisser tables with tau-neutrino columns (hg_taumode) are not "
                                  "part of this build")
    else:
        raise ValueError("Flux file must be from the Honda, Hillas, or Bartol groups")
    return FluxTable2D(spline_dict, enpow)


def calculate_2d_flux_weights(true_energies, true_coszens, table, out_nu=None, out_nubar=None):
    """All four primaries at `true_energies`, `true_coszens` (device tensors or host
    arrays): returns device tensors nu_flux[n, 2] = (nue, numu) and
    nubar_flux[n, 2] = (nuebar, numubar).  (The reference's function takes ONE
    primary's splines and is called four times, honda_ip.py:92-103.)"""
    import torch

    e = true_energies if torch.is_tensor(true_energies) else K.to_device(np.asarray(true_energies))
    cz = true_coszens if torch.is_tensor(true_coszens) else K.to_device(np.asarray(true_coszens))
    if e.numel() != cz.numel():
        raise ValueError("length of energy and coszen arrays must match")
    return K.flux_2d(table, e.contiguous(), cz.contiguous(), out_nu, out_nubar)
