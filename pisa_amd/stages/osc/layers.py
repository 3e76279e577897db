"""Earth model and neutrino paths.

Counterpart of pisa/stages/osc/layers.py (class Layers, :172-481).  The shell
table (radii, electron-density-weighted rho, tangency cosines) is tiny and is
prepared on the host exactly as the reference does; the per-coszen path
construction (`extCalcLayers`, layers.py:38-169) runs on the GPU
(`pisa_hip_calc_layers`), or is skipped altogether in event mode where the
prob3 kernel rebuilds each event's path from the shell table held in LDS.
"""
import numpy as np

from pisa_amd import FTYPE, _lib
from pisa_amd.utils.resources import find_resource

__all__ = ["Layers"]


class Layers:
    R_INNER, R_OUTER, R_MANTLE = 1221.5, 3480.0, 6371.0  # layers.py:419-421

    def __init__(self, prem_file, detector_depth=1.0, prop_height=2.0):
        if prem_file is not None:
            self.using_earth_model = True
            prem = prem_file if isinstance(prem_file, np.ndarray) else np.loadtxt(find_resource(prem_file))
            self.prem = np.asarray(prem, dtype=FTYPE)
            r_earth = self.prem[-1][0]
            # surface -> centre, with the production shell prepended (layers.py:226-241)
            self.radii = np.concatenate(([r_earth + prop_height], self.prem[:, 0][::-1])).astype(FTYPE)
            self.rhos_unweighted = np.concatenate(([1.0], self.prem[:, 1][::-1])).astype(FTYPE)
            self.rhos = self.rhos_unweighted.copy()
            self.rhos_neutron_weighted = self.rhos_unweighted.copy()
            self.default_elec_frac = 0.5
            self.max_layers = 2 * len(self.radii)
        else:
            self.using_earth_model = False
            r_earth = 6371.0
        assert detector_depth > 0, "ERROR: detector depth must be a positive value"
        assert detector_depth <= r_earth, "ERROR: detector depth is deeper than one Earth radius!"
        assert prop_height >= 0, "ERROR: neutrino production height must be positive"
        self.r_detector = r_earth - detector_depth
        self.prop_height = prop_height
        self.detector_depth = detector_depth
        self._n_layers = self._density = self._distance = None
        self._dev = None
        if self.using_earth_model:
            self.computeMinLengthToLayers()

    def _need_model(self, what):
        if not self.using_earth_model:
            raise ValueError("Cannot %s when not using an Earth model" % what)

    def computeMinLengthToLayers(self):
        """cos(zenith) at which a track is tangent to each shell (layers.py:308-335)."""
        lim = np.ones(len(self.radii), dtype=FTYPE)
        inner = self.radii < self.r_detector
        lim[inner] = -np.sqrt(1 - (self.radii[inner] ** 2 / self.r_detector ** 2))
        self.coszen_limit = lim

    def _weight(self, frac):
        r = self.radii
        w = (frac[0] * (r <= self.R_INNER)
             + frac[1] * ((r <= self.R_OUTER) & (r > self.R_INNER))
             + frac[2] * ((r <= self.R_MANTLE) & (r > self.R_OUTER)))
        return w

    def setElecFrac(self, YeI, YeO, YeM):
        """Ye-weight the shell densities by region (layers.py:262-276, 411-439)."""
        self._need_model("set electron fraction")
        self.YeFrac = np.array([YeI, YeO, YeM], dtype=FTYPE)
        self.YnFrac = np.array([1 - YeI, 1 - YeO, 1 - YeM], dtype=FTYPE)
        r = self.radii
        u = self.rhos_unweighted
        # same association as the reference: (rho*Ye)*mask summed region by region
        self.rhos = (u * self.YeFrac[0] * (r <= self.R_INNER)
                     + u * self.YeFrac[1] * (r <= self.R_OUTER) * (r > self.R_INNER)
                     + u * self.YeFrac[2] * (r <= self.R_MANTLE) * (r > self.R_OUTER))
        self.rhos_neutron_weighted = (u * self.YnFrac[0] * (r <= self.R_INNER)
                                      + u * self.YnFrac[1] * (r <= self.R_OUTER) * (r > self.R_INNER)
                                      + u * self.YnFrac[2] * (r <= self.R_MANTLE) * (r > self.R_OUTER))

    def scaling(self, scaling_array):
        """Tomography density scaling (layers.py:278-292)."""
        if not (self.using_earth_model and hasattr(self, "prem")):
            raise ValueError("Cannot scale densities when not using an Earth model")
        rhos = self.prem[:, 1][::-1].astype(FTYPE)
        if scaling_array is not None:
            rhos = rhos * scaling_array
        self.rhos = np.concatenate((np.ones(1, dtype=FTYPE), rhos))
        # as in the reference, `rhos_unweighted` is NOT touched: a following setElecFrac
        # (which prob3.compute_function always issues, prob3.py:533) re-derives `rhos` from
        # the unscaled densities (layers.py:433-439), i.e. in this version of the reference
        # the scaling does not reach the propagation (see `prob3._apply_tomography`).

    def earth_struct(self):
        """`pisa_hip_earth` block for the C ABI."""
        self._need_model("export the shell table")
        return _lib.make_earth(self.radii, self.rhos, self.coszen_limit, self.r_detector)

    def calcLayers(self, cz):
        """Per-coszen densities/distances [n, max_layers] on the device
        (layers.py:338-361 -> extCalcLayers)."""
        self._need_model("calculate layers")
        from pisa_amd import kernels as K

        cz_d = cz if hasattr(cz, "is_cuda") else K.to_device(np.asarray(cz, dtype=FTYPE))
        self._dev = K.calc_layers(self.earth_struct(), cz_d, self.max_layers)
        self._n_layers = self._density = self._distance = None

    @property
    def device_arrays(self):
        """(n_layers, densities, distances) device tensors of the last calcLayers."""
        return self._dev

    @property
    def n_layers(self):
        self._need_model("get layers")
        if self._n_layers is None:
            self._n_layers = self._dev[0].cpu().numpy()
        return self._n_layers

    @property
    def density(self):
        self._need_model("get density")
        if self._density is None:
            self._density = self._dev[1].cpu().numpy()
        return self._density

    @property
    def distance(self):
        if self._distance is None:
            self._distance = self._dev[2].cpu().numpy()
        return self._distance

    def calcPathLength(self, cz):
        """Vacuum path length through an Earth-sized sphere (layers.py:384-405)."""
        r_prop = self.r_detector + self.detector_depth + self.prop_height
        cz = np.atleast_1d(np.asarray(cz, dtype=FTYPE))
        self._distance = -self.r_detector * cz + np.sqrt(
            self.r_detector ** 2.0 * cz ** 2 - (self.r_detector ** 2.0 - r_prop ** 2.0))
