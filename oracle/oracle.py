"""ctypes/numpy front-end of the CPU oracle (``oracle/pisa_oracle.c``).

TEST INFRASTRUCTURE ONLY -- imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg, never by the product package ``pisa_amd``.
Each wrapper names the reference function it restates (file:line under
/root/reference); the arithmetic lives in the C file.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

c_dp = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)


def build(force=False):
    """Compile liboracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "pisa_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _c16(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def set_num_threads(n):
    lib().oracle_set_num_threads(C.c_int(int(n)))


def num_threads():
    return lib().oracle_num_threads()


# --------------------------------------------------------------------------
# prob3 (numba_osc_kernels.py / numba_osc_hostfuncs.py)
# --------------------------------------------------------------------------
def get_H_vac(mix_nubar, mix_nubar_conj_transp, dm):
    out = np.zeros((3, 3), np.complex128)
    lib().oracle_get_H_vac(_p(_c16(mix_nubar)), _p(_c16(mix_nubar_conj_transp)), _p(_f8(dm)), _p(out))
    return out


def get_H_decay(mix_nubar, mix_nubar_conj_transp, mat_decay):
    out = np.zeros((3, 3), np.complex128)
    lib().oracle_get_H_decay(
        _p(_c16(mix_nubar)), _p(_c16(mix_nubar_conj_transp)), _p(_c16(mat_decay)), _p(out)
    )
    return out


def get_H_mat(rho, mat_pot, nubar):
    out = np.zeros((3, 3), np.complex128)
    lib().oracle_get_H_mat(C.c_double(rho), _p(_c16(mat_pot)), C.c_int64(int(nubar)), _p(out))
    return out


def get_dms(energy, H_full, dm):
    dmm = np.zeros((3, 3), np.complex128)
    dmat = np.zeros((3, 3), np.complex128)
    lib().oracle_get_dms(C.c_double(energy), _p(_c16(H_full)), _p(_f8(dm)), _p(dmm), _p(dmat))
    return dmm, dmat


def get_dms_numerical(energy, H_full):
    dmm = np.zeros((3, 3), np.complex128)
    dmat = np.zeros((3, 3), np.complex128)
    lib().oracle_get_dms_numerical(C.c_double(energy), _p(_c16(H_full)), _p(dmm), _p(dmat))
    return dmm, dmat


def get_product(energy, dm_mat, dm_mat_mat, H_mass):
    out = np.zeros((3, 3, 3), np.complex128)
    lib().oracle_get_product(
        C.c_double(energy), _p(_c16(dm_mat)), _p(_c16(dm_mat_mat)), _p(_c16(H_mass)), _p(out)
    )
    return out


def get_transition_matrix_massbasis(baseline, energy, dm_mat, dm_mat_mat, H_mass):
    out = np.zeros((3, 3), np.complex128)
    lib().oracle_get_transition_matrix_massbasis(
        C.c_double(baseline), C.c_double(energy), _p(_c16(dm_mat)), _p(_c16(dm_mat_mat)),
        _p(_c16(H_mass)), _p(out),
    )
    return out


def get_transition_matrix(nubar, energy, rho, baseline, mix_nubar, mix_nubar_conj_transp, mat_pot,
                          H_vac, decay_flag, H_decay, lri_pot, dm):
    out = np.zeros((3, 3), np.complex128)
    lib().oracle_get_transition_matrix(
        C.c_int64(int(nubar)), C.c_double(energy), C.c_double(rho), C.c_double(baseline),
        _p(_c16(mix_nubar)), _p(_c16(mix_nubar_conj_transp)), _p(_c16(mat_pot)), _p(_c16(H_vac)),
        C.c_int64(int(decay_flag)), _p(_c16(H_decay)), _p(_f8(lri_pot)), _p(_f8(dm)), _p(out),
    )
    return out


def propagate_array(dm, mix, mat_pot, decay_flag, mat_decay, lri_pot, nubar, energy, densities,
                    distances):
    """numba_osc_hostfuncs.py:56-70.  ``energy`` [N]; ``densities``/``distances``
    either [L] (shared by all elements) or [N, L]. Returns probability [N,3,3]."""
    energy = _f8(np.atleast_1d(energy))
    n = energy.size
    densities = _f8(densities)
    distances = _f8(distances)
    per_elem = 1 if densities.ndim == 2 else 0
    n_layers = densities.shape[-1]
    out = np.zeros((n, 3, 3), np.float64)
    rc = lib().oracle_propagate_array(
        _p(_f8(dm)), _p(_c16(mix)), _p(_c16(mat_pot)), C.c_int64(int(decay_flag)),
        _p(_c16(mat_decay)), _p(_f8(lri_pot)), C.c_int64(int(nubar)), _p(energy), _p(densities),
        _p(distances), C.c_int64(n), C.c_int(n_layers), C.c_int(per_elem), _p(out),
    )
    if rc:
        raise ValueError("oracle_propagate_array failed rc=%d" % rc)
    return out


def fill_probs(probability, init_flav, flav):
    probability = _f8(probability)
    n = probability.shape[0]
    out = np.zeros(n)
    lib().oracle_fill_probs(_p(probability), C.c_int64(init_flav), C.c_int64(flav), C.c_int64(n), _p(out))
    return out


# --------------------------------------------------------------------------
# Earth model: host-side part of Layers (layers.py:216-289, 308-335, 411-439)
# restated in numpy, the per-coszen loop (layers.py:38-169) in C.
# --------------------------------------------------------------------------
class Layers:
    """Restates pisa/stages/osc/layers.py:172-481 (class Layers)."""

    R_INNER = 1221.5
    R_OUTER = 3480.0
    R_MANTLE = 6371.0

    def __init__(self, prem, detector_depth=1.0, prop_height=2.0):
        prem = np.asarray(prem, dtype=np.float64)  # rows (radius, density), centre -> surface
        self.rhos_unweighted = prem[:, 1][::-1].copy()
        self.radii = prem[:, 0][::-1].copy()
        r_earth = prem[-1, 0]
        self.radii = np.concatenate(([r_earth + prop_height], self.radii))
        self.rhos_unweighted = np.concatenate(([1.0], self.rhos_unweighted))
        self.rhos = self.rhos_unweighted.copy()
        self.max_layers = 2 * len(self.radii)
        assert detector_depth > 0 and detector_depth <= r_earth and prop_height >= 0
        self.r_detector = r_earth - detector_depth
        self.prop_height = prop_height
        self.detector_depth = detector_depth
        # computeMinLengthToLayers (layers.py:308-335)
        lim = []
        for rad in self.radii:
            if rad >= self.r_detector:
                lim.append(1.0)
            else:
                lim.append(-np.sqrt(1 - (rad ** 2 / self.r_detector ** 2)))
        self.coszen_limit = np.array(lim, dtype=np.float64)

    def setElecFrac(self, YeI, YeO, YeM):
        # weight_density_to_YeFrac (layers.py:411-439)
        ye = np.array([YeI, YeO, YeM], dtype=np.float64)
        r = self.radii
        inner = self.rhos_unweighted * ye[0] * (r <= self.R_INNER)
        outer = self.rhos_unweighted * ye[1] * (r <= self.R_OUTER) * (r > self.R_INNER)
        mantle = self.rhos_unweighted * ye[2] * (r <= self.R_MANTLE) * (r > self.R_OUTER)
        self.rhos = inner + outer + mantle

    def calcLayers(self, cz):
        cz = _f8(np.atleast_1d(cz))
        n = cz.size
        self.n_layers = np.zeros(n)
        self.density = np.zeros((n, self.max_layers))
        self.distance = np.zeros((n, self.max_layers))
        rc = lib().oracle_calc_layers(
            _p(cz), C.c_int64(n), C.c_double(self.r_detector), _p(_f8(self.rhos)),
            _p(_f8(self.coszen_limit)), _p(_f8(self.radii)), C.c_int(len(self.radii)),
            C.c_int(self.max_layers), _p(self.n_layers), _p(self.density), _p(self.distance),
        )
        if rc:
            raise ValueError("oracle_calc_layers failed rc=%d" % rc)


# --------------------------------------------------------------------------
# translation.py: lookup / histogram
# --------------------------------------------------------------------------
def _sample_ptrs(sample):
    cols = [_f8(s) for s in sample]
    arr = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    return cols, arr


def lookup_regular(sample, flat_hist, mins, maxs, nbins):
    """translation.py:417-501 (lookup_regular_{1,2,3}d and *_array)."""
    cols, arr = _sample_ptrs(sample)
    n = cols[0].size
    flat_hist = _f8(flat_hist)
    width = 1 if flat_hist.ndim == 1 else flat_hist.shape[1]
    out = np.zeros((n,) if flat_hist.ndim == 1 else (n, width))
    nb = np.ascontiguousarray(nbins, dtype=np.int64)
    lib().oracle_lookup_regular(
        C.c_int(len(cols)), arr, C.c_int64(n), _p(flat_hist), C.c_int(width), _p(_f8(mins)),
        _p(_f8(maxs)), _p(nb), _p(out),
    )
    return out


def histogram_regular(sample, weights, mins, maxs, nbins):
    """fast_histogram.histogramdd rule as called from translation.py:171-205."""
    cols, arr = _sample_ptrs(sample)
    n = cols[0].size
    nb = np.ascontiguousarray(nbins, dtype=np.int64)
    out = np.zeros(int(np.prod(nb)))
    w = None if weights is None else _f8(weights)
    lib().oracle_histogram_regular(
        C.c_int(len(cols)), arr, C.c_int64(n), _p(w) if w is not None else None, _p(_f8(mins)),
        _p(_f8(maxs)), _p(nb), _p(out),
    )
    return out


def reweight(initial_weights, nu_flux, prob_e, prob_mu, weighted_aeff, scale):
    """prob3.py:621-622 followed by aeff.py:78-88."""
    n = len(initial_weights)
    out = np.zeros(n)
    lib().oracle_reweight(
        _p(_f8(initial_weights)), _p(_f8(nu_flux)), _p(_f8(prob_e)), _p(_f8(prob_mu)),
        _p(_f8(weighted_aeff)), C.c_double(scale), C.c_int64(n), _p(out),
    )
    return out


class PartitionedCopy:
    """numpy view of a copy of `a` whose pages were first touched by the OpenMP threads that read them
    in `container_chain` (NUMA placement for the all-core baseline); frees the copy with the object"""

    def __init__(self, a):
        a = _f8(a)
        rows = a.shape[0]
        row_bytes = a.strides[0] if a.ndim > 1 else a.itemsize
        fn = lib().oracle_partitioned_copy
        fn.restype = C.c_void_p
        self._ptr = fn(_p(a), C.c_int64(rows), C.c_int64(row_bytes))
        if not self._ptr:
            raise MemoryError("oracle_partitioned_copy")
        buf = (C.c_double * a.size).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=np.float64).reshape(a.shape)

    def __del__(self):
        if getattr(self, "_ptr", None):
            lib().oracle_free(C.c_void_p(self._ptr))
            self._ptr = None


def container_chain(gx, gy, gmins, gmaxs, gnb, pe_grid, pmu_grid, initial_weights, nu_flux, weighted_aeff,
                    scale, sample, mins, maxs, nbins):
    """lookup (translation.py:427-438) + prob3.apply (prob3.py:621-622) + aeff.apply (aeff.py:78-88) +
    histogram of w and w^2 (hist.py:163-218) of one container in ONE OpenMP loop over its events
    (per-thread private histograms, added in thread order).  Returns (hist, sumw2)."""
    cols, arr = _sample_ptrs(sample)
    n = cols[0].size
    nb = np.ascontiguousarray(nbins, dtype=np.int64)
    gnb_ = np.ascontiguousarray(gnb, dtype=np.int64)
    hist, sumw2 = np.zeros(int(np.prod(nb))), np.zeros(int(np.prod(nb)))
    rc = lib().oracle_container_chain(
        C.c_int64(n), _p(_f8(gx)), _p(_f8(gy)), _p(_f8(gmins)), _p(_f8(gmaxs)), _p(gnb_), _p(_f8(pe_grid)),
        _p(_f8(pmu_grid)), _p(_f8(initial_weights)), _p(_f8(nu_flux)), _p(_f8(weighted_aeff)),
        C.c_double(scale), C.c_int(len(cols)), arr, _p(_f8(mins)), _p(_f8(maxs)), _p(nb), _p(hist), _p(sumw2))
    if rc:
        raise RuntimeError("oracle_container_chain: %d" % rc)
    return hist, sumw2


METRIC_KIND = {"llh": 0, "poisson_llh": 1, "chi2": 2, "mod_chi2": 3}


def metric(kind, actual, expected, sigma2=None):
    """stats.py llh/poisson_llh/chi2/mod_chi2 + np.nansum (map.py:1604).
    Returns (per_bin, total)."""
    actual = _f8(actual).ravel()
    expected = _f8(expected).ravel()
    per_bin = np.zeros(actual.size)
    total = C.c_double(0.0)
    s2 = None if sigma2 is None else _f8(sigma2).ravel()
    rc = lib().oracle_metric(
        C.c_int(METRIC_KIND[kind]), _p(actual), _p(expected), _p(s2) if s2 is not None else None,
        C.c_int64(actual.size), _p(per_bin), C.byref(total),
    )
    if rc:
        raise ValueError("`actual_values`/`expected_values` must all be >= 0")
    return per_bin, total.value


def barr_simple(true_energy, true_coszen, nu_flux_nominal, nubar_flux_nominal, nubar,
                nue_numu_ratio, nu_nubar_ratio, delta_index, Barr_uphor_ratio,
                Barr_nu_nubar_ratio):
    """flux/barr_simple.py:147-233."""
    n = len(true_energy)
    out = np.zeros((n, 2))
    lib().oracle_barr_simple(
        _p(_f8(true_energy)), _p(_f8(true_coszen)), _p(_f8(nu_flux_nominal)),
        _p(_f8(nubar_flux_nominal)), C.c_int64(int(nubar)), C.c_double(nue_numu_ratio),
        C.c_double(nu_nubar_ratio), C.c_double(delta_index), C.c_double(Barr_uphor_ratio),
        C.c_double(Barr_nu_nubar_ratio), C.c_int64(n), _p(out),
    )
    return out


# --------------------------------------------------------------------------
# host-side parameter matrices (numpy restatements; tiny, per-eval)
# --------------------------------------------------------------------------
def mix_matrix(theta12, theta13, theta23, deltacp, reparam=False):
    """osc_params.py:174-211 (standard) / :213-258 (reparameterised).
    Angles in rad; the reference stores sin(theta) and uses c = sqrt(1 - s^2)."""
    s12, s13, s23 = np.sin(theta12), np.sin(theta13), np.sin(theta23)
    sd, cd = np.sin(deltacp), np.cos(deltacp)
    c12 = np.sqrt(1.0 - s12 ** 2)
    c23 = np.sqrt(1.0 - s23 ** 2)
    c13 = np.sqrt(1.0 - s13 ** 2)
    m = np.zeros((3, 3, 2))
    if not reparam:
        m[0, 0, 0] = c12 * c13
        m[0, 1, 0] = s12 * c13
        m[0, 2, 0] = s13 * cd
        m[0, 2, 1] = -s13 * sd
        m[1, 0, 0] = -s12 * c23 - c12 * s23 * s13 * cd
        m[1, 0, 1] = -c12 * s23 * s13 * sd
        m[1, 1, 0] = c12 * c23 - s12 * s23 * s13 * cd
        m[1, 1, 1] = -s12 * s23 * s13 * sd
        m[1, 2, 0] = s23 * c13
        m[2, 0, 0] = s12 * s23 - c12 * c23 * s13 * cd
        m[2, 0, 1] = -c12 * c23 * s13 * sd
        m[2, 1, 0] = -c12 * s23 - s12 * c23 * s13 * cd
        m[2, 1, 1] = -s12 * c23 * s13 * sd
        m[2, 2, 0] = c23 * c13
    else:
        m[0, 0, 0] = c12 * c13
        m[0, 1, 0] = s12 * c13 * cd
        m[0, 1, 1] = s12 * c13 * sd
        m[0, 2, 0] = s13
        m[1, 0, 0] = -s12 * c23 * cd - c12 * s23 * s13
        m[1, 0, 1] = s12 * c23 * sd
        m[1, 1, 0] = c12 * c23 - s12 * s23 * s13 * cd
        m[1, 1, 1] = -s12 * s23 * s13 * sd
        m[1, 2, 0] = s23 * c13
        m[2, 0, 0] = s12 * s23 * cd - c12 * c23 * s13
        m[2, 0, 1] = -s12 * s23 * sd
        m[2, 1, 0] = -c12 * s23 - s12 * c23 * s13 * cd
        m[2, 1, 1] = -s12 * c23 * s13 * sd
        m[2, 2, 0] = c23 * c13
    return m[:, :, 0] + m[:, :, 1] * 1.0j


def dm_matrix(dm21, dm31):
    """osc_params.py:265-292."""
    dm = np.zeros((3, 3))
    m = np.zeros(3)
    delta = 5.0e-9
    m[1] = dm21
    m[2] = dm31
    if m[1] == 0.0:
        m[0] -= delta
    if m[2] == 0.0:
        m[2] += delta
    dm[0, 1] = m[0] - m[1]
    dm[1, 0] = -dm[0, 1]
    dm[0, 2] = m[0] - m[2]
    dm[2, 0] = -dm[0, 2]
    dm[1, 2] = m[1] - m[2]
    dm[2, 1] = -dm[1, 2]
    return dm


def kde_eval(src, coef, s2, qry, inv_cov):
    """all-pairs Gaussian kernel sums (KDE core; parity unpinned, see pisa_oracle.c)"""
    src, qry = _f8(src), _f8(qry)
    dim, n = src.shape
    m = qry.shape[1]
    out = np.zeros(m)
    lib().oracle_kde_eval(C.c_int(dim), _p(src), _p(_f8(coef)), _p(_f8(s2)), C.c_int64(n), _p(qry),
                          C.c_int64(m), _p(_f8(inv_cov)), _p(out))
    return out
