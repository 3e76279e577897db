"""Earth-tomography density scalings (counterpart of pisa/stages/osc/scaling_params.py:26-146).

Three parameterisations of per-shell density scale factors for `Layers.scaling`:

* `Mass_scaling`               one factor for every shell (`density_scale`);
* `Core_scaling_w_constrain`   the core factor alpha is free; the inner- and middle-mantle
                               factors beta, gamma follow from keeping the Earth's mass and
                               moment of inertia fixed (two linear equations);
* `Core_scaling_wo_constrain`  core, inner-mantle and middle-mantle factors independent.

The last two assume the five-shell Earth of `FIVE_LAYER_RADII` / `FIVE_LAYER_RHOS`
(scaling_params.py:14-18); `osc.prob3` checks the Earth-model file against them (prob3.py:378-390).
Arrays are ordered like `Layers.rhos` without the atmosphere shell: surface -> centre.
"""
import numpy as np

from pisa_amd import FTYPE

__all__ = ["Mass_scaling", "Core_scaling_w_constrain", "Core_scaling_wo_constrain",
           "FIVE_LAYER_RADII", "FIVE_LAYER_RHOS", "TOMOGRAPHY_ERROR_MSG"]

FIVE_LAYER_RADII = np.array([0.0, 1221.50, 3480.00, 5701.00, 6151.0, 6371.00], dtype=FTYPE)  # km
FIVE_LAYER_RHOS = np.array([13.0, 13.0, 10.96, 5.03, 3.7, 2.5], dtype=FTYPE)                 # g/cm^3

TOMOGRAPHY_ERROR_MSG = (
    "You need to provide the appropriate 5-layer Earth model, which has the same layer radii (%s km) "
    "and densities (%s g/cm^3) as the one hard-coded for the chosen type of tomography internally."
    % (FIVE_LAYER_RADII.tolist(), FIVE_LAYER_RHOS.tolist()))


class Mass_scaling:  # pylint: disable=invalid-name
    def __init__(self):
        self._density_scale = 0.0

    @property
    def density_scale(self):
        return self._density_scale

    @density_scale.setter
    def density_scale(self, value):
        assert value >= 0.0
        self._density_scale = value


def _shell_moments():
    """mass-like (4 pi/3 rho dr^3) and inertia-like (8 pi/15 rho dr^5) integrals of the five
    shells, centre outwards; radii in km and densities in g/cm^3 as in the reference, which
    switches to Gt/km^3 = g/cm^3 numerically to keep the powers in range"""
    r, rho = FIVE_LAYER_RADII, FIVE_LAYER_RHOS
    mass, inertia = [], []
    for k in range(1, 6):
        mass.append((4 * np.pi / 3) * (rho[k] * (r[k] ** 3 - r[k - 1] ** 3)))
        inertia.append((8 * np.pi / 15) * (rho[k] * (r[k] ** 5 - r[k - 1] ** 5)))
    return mass, inertia


class Core_scaling_w_constrain:  # pylint: disable=invalid-name
    def __init__(self):
        self._core_density_scale = 0.0

    @property
    def core_density_scale(self):
        return self._core_density_scale

    @core_density_scale.setter
    def core_density_scale(self, value):
        self._core_density_scale = value

    @property
    def scaling_array(self):
        """[1, gamma, beta, alpha, alpha, alpha] (outer mantle unscaled; scaling_params.py:64-107)"""
        (a1, b1, c1, d1, e1), (a2, b2, c2, d2, e2) = _shell_moments()
        inertia = a2 + b2 + c2 + d2 + e2
        mass = a1 + b1 + c1 + d1 + e1
        alpha = self.core_density_scale
        gamma = ((inertia * c1 - mass * c2) - alpha * (c1 * a2 - c2 * a1) - alpha * (c1 * b2 - b1 * c2)
                 - (c1 * e2 - e1 * c2)) / (c1 * d2 - d1 * c2)
        beta = (inertia - alpha * a2 - alpha * b2 - gamma * d2 - e2) / c2
        # density scaling factors mustn't be negative
        assert (np.asarray([alpha, beta, gamma], dtype=FTYPE) >= 0).all()
        out = np.ones(6, dtype=FTYPE)
        out[1], out[2] = gamma, beta
        out[3:] = alpha
        return out


class Core_scaling_wo_constrain:  # pylint: disable=invalid-name
    def __init__(self):
        self.core_density_scale = 0.0
        self.innermantle_density_scale = 0.0
        self.middlemantle_density_scale = 0.0

    @property
    def scaling_factor_array(self):
        out = np.ones(6, dtype=FTYPE)
        out[1] = self.middlemantle_density_scale
        out[2] = self.innermantle_density_scale
        out[3:] = self.core_density_scale
        return out
