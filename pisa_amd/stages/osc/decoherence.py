"""Oscillations with decoherence in vacuum (counterpart of pisa/stages/osc/decoherence.py:32-494): nue do not
oscillate, numu disappear into nutau with
D = 2 sum_{j>k} |U[2][j]|^2 |U[2][k]|^2 (1 - exp(-Gamma_jk L) cos(dm2_jk L / 2E))   (:229-269; natural units through the
reference's 5.07e18 / 1e-18 factors), the path length L that of a sphere of 6371 km (`Layers(None, ...)`,
layers.py:384-405).  `earth_model` must be None as there.  Per container one launch of `pisa_hip_decoherence_probs`
(the [N, 3, 3] table), `pisa_hip_fill_probs` twice and, per run, `weights *= sys_flux . (prob_e, prob_mu)`."""
import math

import numpy as np

from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage
from pisa_amd.core.units import Quantity, ureg
from pisa_amd.stages.osc.layers import Layers
from pisa_amd.stages.osc.osc_params import OscParams

__all__ = ["DecoherenceParams", "calc_decoherence_probs", "decoherence"]


class DecoherenceParams(OscParams):
    """standard oscillation parameters plus the three Gamma_jk (energies); all kept as quantities (decoherence.py:32-63)"""

    def __init__(self, deltam21, deltam31, theta12, theta13, theta23, deltacp, gamma21, gamma31, gamma32):
        super().__init__()
        self.__dict__.update(dm21=deltam21, dm31=deltam31, gamma21=gamma21, gamma31=gamma31, gamma32=gamma32)
        # the base class keeps the angles as sines behind properties of the same names: the quantities live beside them
        self._q = dict(theta12=theta12, theta13=theta13, theta23=theta23, deltacp=deltacp)
        self.dm32 = self.dm31 - self.dm21

    def angle(self, name):
        """the reference assigns the quantity to OscParams' `thetaXY` property, which keeps sin(theta) and hands back
        arcsin of it (osc_params.py:86-152)"""
        return float(np.arcsin(np.sin(self._q[name].m_as("rad"))))


def _tau_row_sq(p):
    """|U[2][k]|^2 of the real matrix of decoherence.py:176-227 (delta_cp phases entered as 0.0 there)"""
    c12, c13, c23 = (math.cos(p.angle(n)) for n in ("theta12", "theta13", "theta23"))
    s12, s13, s23 = (math.sin(p.angle(n)) for n in ("theta12", "theta13", "theta23"))
    eid = 0.0
    row = [(s12 * s23) - (c12 * c23 * s13 * eid), (0.0 - c12 * s23) - (s12 * c23 * s13 * eid), c23 * c13]
    return [abs(v) ** 2 for v in row]


def _kernel_arguments(p, two_flavor):
    if two_flavor:                                                  # decoherence.py:135-139
        return ([0.5 * (np.sin(2.0 * p.angle("theta23")) ** 2), 0.0, 0.0], [p.gamma32.m_as("eV"), 0.0, 0.0],
                [p.dm32.m_as("eV**2"), 0.0, 0.0])
    u2 = _tau_row_sq(p)
    pairs = ((1, 0), (2, 0), (2, 1))
    gamma = {(1, 0): p.gamma21, (2, 0): p.gamma31, (2, 1): p.gamma32}
    delta = {(1, 0): p.dm21, (2, 0): p.dm31, (2, 1): p.dm32}
    return ([u2[j] * u2[k] for j, k in pairs], [gamma[jk].m_as("GeV") for jk in pairs],
            [delta[jk].m_as("eV**2") for jk in pairs])


def _probability_table(p, energy, baseline, two_flavor=False):
    coef, gamma, delta = _kernel_arguments(p, two_flavor)
    return K.decoherence_probs(coef, gamma, delta, two_flavor, energy, baseline)


def calc_decoherence_probs(decoh_params, flav, energy, baseline, prob_e, prob_mu, prob_tau, two_flavor=False):
    """decoherence.py:66-106 with host arrays: fills `prob_e / prob_mu / prob_tau` for the initial flavour `flav`
    ('nue...' or 'numu...'); energy in GeV, baseline in km unless they carry units"""
    if not (flav.startswith("nue") or flav.startswith("numu")):
        raise ValueError("Input flavor '%s' not supported" % flav)
    e = energy.m_as("GeV") if isinstance(energy, Quantity) else energy
    length = baseline.m_as("km") if isinstance(baseline, Quantity) else baseline
    shape = np.shape(e)
    table = _probability_table(decoh_params, K.to_device(np.ascontiguousarray(np.ravel(e), dtype=np.float64)),
                               K.to_device(np.ascontiguousarray(np.ravel(length), dtype=np.float64)), two_flavor).cpu().numpy()
    row = 0 if flav.startswith("nue") else 1
    for k, out in enumerate((prob_e, prob_mu, prob_tau)):
        np.copyto(dst=out, src=table[:, row, k].reshape(shape))


class decoherence(Stage):  # pylint: disable=invalid-name
    def __init__(self, **std_kwargs):
        expected_params = ("detector_depth", "earth_model", "prop_height", "YeI", "YeO", "YeM", "theta12", "theta13",
                           "theta23", "deltam21", "deltam31", "deltacp", "gamma21", "gamma31", "gamma32")
        super().__init__(expected_params=expected_params,
                         expected_container_keys=("true_energy", "true_coszen", "weights", "nubar", "flav", "sys_flux"),
                         **std_kwargs)
        if self.params.earth_model.value is not None:
            raise ValueError("Matter effects not yet implemented for decoherence, must set 'earth_model' to None")
        self.layers = None
        self.two_flavor = False

    def setup_function(self):
        self.layers = Layers(None, self.params.detector_depth.value.m_as("km"), self.params.prop_height.value.m_as("km"))
        self.data.representation = self.calc_mode
        for container in self.data:
            self.layers.calcPathLength(container["true_coszen"])
            container["distances"] = self.layers.distance

    def compute_function(self):
        v = {n: self.params[n].value for n in ("deltam21", "deltam31", "theta12", "theta13", "theta23", "deltacp",
                                               "gamma21", "gamma31", "gamma32")}
        self.decoh_params = DecoherenceParams(**v)
        for container in self.data:
            table = _probability_table(self.decoh_params, container.device("true_energy"), container.device("distances"),
                                       self.two_flavor)
            container["probability"] = table
            container["prob_e"] = K.fill_probs(table, 0, container["flav"])
            container["prob_mu"] = K.fill_probs(table, 1, container["flav"])
            for key in ("probability", "prob_e", "prob_mu"):
                container.mark_valid(key)

    def apply_function(self):
        for container in self.data:
            weights = container.device("weights").clone()
            K.apply_osc_weights(container.device("sys_flux"), container.device("prob_e"), container.device("prob_mu"), weights)
            container["weights"] = weights


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    values = [("detector_depth", 0.5 * ureg.km), ("prop_height", 20 * ureg.km), ("earth_model", None), ("YeI", 0.5),
              ("YeO", 0.5), ("YeM", 0.5), ("theta12", 33 * ureg.degree), ("theta13", 8 * ureg.degree),
              ("theta23", 50 * ureg.degree), ("deltam21", 8e-5 * ureg.eV ** 2), ("deltam31", 3e-3 * ureg.eV ** 2),
              ("deltacp", 180 * ureg.degree), ("gamma21", 1e-11 * ureg.GeV), ("gamma31", 5e-10 * ureg.GeV),
              ("gamma32", 2.5e-13 * ureg.GeV)]
    return decoherence(params=ParamSet([Param(name=n, value=v, **param_kwargs) for n, v in values]))
