"""Three-flavour oscillation probabilities in layered-Earth matter, with NSI,
decay, LRI and NLO options (counterpart of pisa/stages/osc/prob3.py:37-641:
same constructor kwargs, params, container keys and results).

Host side per evaluation: parameter values -> PMNS / dm / generalised matter
potential / decay / LRI matrices (prob3.py:476-578).  Device side:

* calc_mode = 2-D (true_energy x true_coszen) binning: the planned two-stage
  grid kernels evaluate nu and nubar on every node at once
  (`pisa_hip_prob3_grid_planned`); layers are computed for the coszen nodes only;
* calc_mode = "events": `pisa_hip_prob3_events` rebuilds each event's path from
  its coszen (no densities/distances[N, L] arrays);
* any other binned calc_mode: generic `pisa_hip_propagate_array` with per-node
  layer rows.

`apply_function` (prob3.py:611-622) multiplies the weights by
flux_e*prob_e + flux_mu*prob_mu; in an event representation this is recorded
as a deferred operation and fused with aeff + hist by `utils.hist`.
"""
import numpy as np

from pisa_amd import CTYPE, FTYPE, _lib
from pisa_amd import kernels as K
from pisa_amd.core.binning import MultiDimBinning
from pisa_amd.core.stage import Stage
from pisa_amd.core.units import ureg
from pisa_amd.stages import deferred
from pisa_amd.stages.osc.decay_params import DecayParams
from pisa_amd.stages.osc.layers import Layers
from pisa_amd.stages.osc.lri_params import LRIParams
from pisa_amd.stages.osc.nsi_params import StdNSIParams, VacuumLikeNSIParams
from pisa_amd.stages.osc.osc_params import OscParams
from pisa_amd.stages.osc.scaling_params import (FIVE_LAYER_RADII, FIVE_LAYER_RHOS, TOMOGRAPHY_ERROR_MSG,
                                                Core_scaling_w_constrain, Core_scaling_wo_constrain,
                                                Mass_scaling)

__all__ = ["prob3", "LRI_TYPES", "NSI_TYPES", "TOMOGRAPHY_TYPES"]

LRI_TYPES = ["emu-symmetry", "etau-symmetry", "mutau-symmetry"]
NSI_TYPES = ["standard", "vacuum-like"]
TOMOGRAPHY_TYPES = ["mass_of_earth", "mass_of_core_w_constrain", "mass_of_core_wo_constrain"]

NU = ["nue_cc", "numu_cc", "nutau_cc", "nue_nc", "numu_nc", "nutau_nc"]
NUBAR = ["nuebar_cc", "numubar_cc", "nutaubar_cc", "nuebar_nc", "numubar_nc", "nutaubar_nc"]


class prob3(Stage):  # pylint: disable=invalid-name
    def __init__(self, include_nlo=False, nsi_type=None, reparam_mix_matrix=False,
                 neutrino_decay=False, tomography_type=None, lri_type=None, **std_kwargs):
        expected_params = ("detector_depth", "earth_model", "prop_height", "YeI", "YeO", "YeM",
                           "theta12", "theta13", "theta23", "deltam21", "deltam31", "deltacp")
        expected_container_keys = ("true_energy", "true_coszen", "nubar", "flav", "nu_flux", "weights")
        self.include_nlo = include_nlo
        if nsi_type is not None:
            nsi_type = nsi_type.strip().lower()
            if nsi_type not in NSI_TYPES:
                raise ValueError('Chosen NSI type "%s" not available! Choose one of %s.' % (nsi_type, NSI_TYPES))
        self.nsi_type = nsi_type
        self.reparam_mix_matrix = reparam_mix_matrix
        self.neutrino_decay = neutrino_decay
        self.decay_flag = 1 if neutrino_decay else -1
        if nsi_type == "vacuum-like":
            expected_params += ("eps_scale", "eps_prime", "phi12", "phi13", "phi23", "alpha1",
                                "alpha2", "deltansi")
        elif nsi_type == "standard":
            expected_params += ("eps_ee", "eps_emu_magn", "eps_emu_phase", "eps_etau_magn",
                                "eps_etau_phase", "eps_mumu", "eps_mutau_magn", "eps_mutau_phase",
                                "eps_tautau")
        if neutrino_decay:
            expected_params += ("decay_alpha3",)
        if lri_type is not None:
            lri_type = lri_type.strip().lower()
            if lri_type not in LRI_TYPES:
                raise ValueError('Chosen LRI symmetry type "%s" not available! Choose one of %s.'
                                 % (lri_type, LRI_TYPES))
            expected_params += ("v_lri",)
        self.lri_type = lri_type
        if tomography_type is not None:
            tomography_type = tomography_type.strip().lower()
            if tomography_type not in TOMOGRAPHY_TYPES:
                raise ValueError('Chosen tomography type "%s" not available! Choose one of %s.'
                                 % (tomography_type, TOMOGRAPHY_TYPES))
            expected_params += {"mass_of_earth": ("density_scale",),
                                "mass_of_core_w_constrain": ("core_density_scale",),
                                "mass_of_core_wo_constrain": ("core_density_scale", "innermantle_density_scale",
                                                              "middlemantle_density_scale")}[tomography_type]
        self.tomography_type = tomography_type
        self.tomography_params = None
        super().__init__(expected_params=expected_params,
                         expected_container_keys=expected_container_keys, **std_kwargs)
        self.layers = self.osc_params = self.nsi_params = self.decay_params = self.lri_params = None
        self.gen_mat_pot_matrix_complex = self.decay_matrix = self.lri_pot = None
        self.YeI = self.YeO = self.YeM = None
        self.grid = None          # dict describing the 2-D calc grid, if any
        self.pepmu = None         # [2][3][node][2] gather tables of the last compute
        self.prob_tables = None   # (P_nu, P_nubar)

    # ---------------------------------------------------------------- setup
    def setup_function(self):
        self.osc_params = OscParams()
        self._std6_seen = None
        if self.nsi_type == "vacuum-like":
            self.nsi_params = VacuumLikeNSIParams()
        elif self.nsi_type == "standard":
            self.nsi_params = StdNSIParams()
        if self.neutrino_decay:
            self.decay_params = DecayParams()
        if self.lri_type is not None:
            self.lri_params = LRIParams()
        p = self.params
        self.YeI = p.YeI.value.m_as("dimensionless")
        self.YeO = p.YeO.value.m_as("dimensionless")
        self.YeM = p.YeM.value.m_as("dimensionless")
        self.layers = Layers(p.earth_model.value, p.detector_depth.value.m_as("km"),
                             p.prop_height.value.m_as("km"))
        self.layers.setElecFrac(self.YeI, self.YeO, self.YeM)
        if self.tomography_type == "mass_of_earth":
            self.tomography_params = Mass_scaling()
        elif self.tomography_type is not None:
            # prob3.py:378-390: the Earth-model file must be the hard-coded five-shell Earth
            radii_ext = self.layers.radii[::-1][:-1]
            rhos_ext = self.layers.rhos_unweighted[::-1][:-1]
            if not (len(radii_ext) == len(FIVE_LAYER_RADII) and len(rhos_ext) == len(FIVE_LAYER_RHOS)):
                raise ValueError(TOMOGRAPHY_ERROR_MSG)
            if not (np.allclose(np.add(radii_ext, 1), np.add(FIVE_LAYER_RADII, 1))
                    and np.allclose(np.add(rhos_ext, 1), np.add(FIVE_LAYER_RHOS, 1))):
                raise ValueError(TOMOGRAPHY_ERROR_MSG)
            self.tomography_params = (Core_scaling_w_constrain() if self.tomography_type == "mass_of_core_w_constrain"
                                      else Core_scaling_wo_constrain())
        cm = self.calc_mode
        if isinstance(cm, MultiDimBinning) and sorted(cm.names) == ["true_coszen", "true_energy"]:
            e_dim, cz_dim = cm["true_energy"], cm["true_coszen"]
            self.grid = dict(
                e_major=(cm.names[0] == "true_energy"),
                energy=K.to_device(e_dim.weighted_centers.m_as("GeV")),
                coszen=K.to_device(cz_dim.weighted_centers.magnitude),
                n_e=e_dim.num_bins, n_cz=cz_dim.num_bins)
        self._calc_layers()
        self.data["_prob3_stage"] = self  # lets utils.hist find the gather tables

    def _calc_layers(self):
        if self.grid is not None:
            self.layers.calcLayers(self.grid["coszen"])
            _, dens, dist = self.layers.device_arrays
            self.grid["plan"] = K.GridPlan(dens, dist)
            self.grid["rows"] = (dens, dist)
        elif self.calc_mode != "events":
            for container in self.data:
                self.layers.calcLayers(container.device("true_coszen"))
                _, dens, dist = self.layers.device_arrays
                container["densities"] = dens
                container["distances"] = dist

    # ---------------------------------------------------------------- compute
    def _matrices(self):
        p = self.params
        # the six parameter objects, looked up once per structural state of the sets (this runs at every
        # point of a fit); units converted once per value (`Param.m_in`)
        from pisa_amd.core.param import ParamSet

        std6 = getattr(self, "_std6", None)
        if std6 is None or std6[0] != ParamSet.struct_clock:
            std6 = self._std6 = (ParamSet.struct_clock,
                                 tuple(p[n] for n in ("theta12", "theta13", "theta23", "deltacp", "deltam21", "deltam31")))
        t12, t13, t23, dcp, d21, d31 = std6[1]
        o = self.osc_params
        # only what moved since the last point (a fit moves one or two of the six; every setter is a numpy call)
        seen = getattr(self, "_std6_seen", None)
        if seen is None or seen[0] is not std6:
            seen = self._std6_seen = [std6, None, None, None, None, None, None]

        def angle(prm):
            if prm.value.units == ureg.dimensionless:
                raise ValueError("%s is dimensionless, but needs units rad or deg!" % prm.name)
            return prm.m_in("rad")

        if seen[1] != t12._ver:
            o.theta12, seen[1] = angle(t12), t12._ver
        if seen[2] != t13._ver:
            o.theta13, seen[2] = angle(t13), t13._ver
        if seen[3] != t23._ver:
            o.theta23, seen[3] = angle(t23), t23._ver
        if seen[4] != d21._ver:
            o.dm21, seen[4] = d21.m_in("eV**2"), d21._ver
        if seen[5] != d31._ver:
            o.dm31, seen[5] = d31.m_in("eV**2"), d31._ver
        if seen[6] != dcp._ver:
            o.deltacp, seen[6] = angle(dcp), dcp._ver
        if self.nsi_type == "vacuum-like":
            n = self.nsi_params
            n.eps_scale = p.eps_scale.value.m_as("dimensionless")
            n.eps_prime = p.eps_prime.value.m_as("dimensionless")
            for a in ("phi12", "phi13", "phi23", "alpha1", "alpha2", "deltansi"):
                setattr(n, a, p[a].value.m_as("rad"))
        elif self.nsi_type == "standard":
            n = self.nsi_params
            n.eps_ee = p.eps_ee.value.m_as("dimensionless")
            n.eps_emu = (p.eps_emu_magn.value.m_as("dimensionless"), p.eps_emu_phase.value.m_as("rad"))
            n.eps_etau = (p.eps_etau_magn.value.m_as("dimensionless"), p.eps_etau_phase.value.m_as("rad"))
            n.eps_mumu = p.eps_mumu.value.m_as("dimensionless")
            n.eps_mutau = (p.eps_mutau_magn.value.m_as("dimensionless"), p.eps_mutau_phase.value.m_as("rad"))
            n.eps_tautau = p.eps_tautau.value.m_as("dimensionless")
        # generalised matter potential (prob3.py:539-557)
        # (the constant matrices are built once: this runs at every point of a fit)
        const = getattr(self, "_const_matrices", None)
        if const is None:
            std = np.zeros((3, 3), dtype=FTYPE) + 1.0j * np.zeros((3, 3), dtype=FTYPE)
            std[0, 0] += 1.020 if self.include_nlo else 1.0
            const = self._const_matrices = (std, np.zeros((3, 3), dtype=CTYPE), np.zeros((3, 3), dtype=FTYPE))
            for m in const:
                m.setflags(write=False)
        std = const[0]
        self.gen_mat_pot_matrix_complex = std + self.nsi_params.eps_matrix if self.nsi_type else std
        if self.neutrino_decay:
            self.decay_params.decay_alpha3 = p.decay_alpha3.value.m_as("eV**2")
            self.decay_matrix = self.decay_params.decay_matrix
        else:
            self.decay_matrix = const[1]
        self.lri_pot = const[2]
        if self.lri_type is not None:
            self.lri_params.v_lri = p.v_lri.value.m_as("eV")
            self.lri_pot = getattr(self.lri_params, "potential_matrix_" + self.lri_type.split("-")[0])
        if self.tomography_type is not None:
            self._apply_tomography()
        block = getattr(self, "_params_block", None)
        if block is None:
            block = self._params_block = _lib.Prob3ParamsBlock()
        # (the entries of dm_matrix / mix_matrix[_reparam]_complex as plain floats, straight into the block)
        return block.update(o.dm_floats(), o.mix_floats(self.reparam_mix_matrix), self.gen_mat_pot_matrix_complex,
                            self.decay_flag, self.decay_matrix, self.lri_pot)

    def _apply_tomography(self):
        """prob3.py:519-536.  `Layers.scaling` rewrites `rhos` from the scaled PREM column, then
        `setElecFrac` -- issued right after it by the reference, :533 -- re-derives `rhos` from the
        UNSCALED `rhos_unweighted` (layers.py:411-439).  In this version of the reference the
        scale factors therefore never reach the propagation; this build follows it call for call
        (same validation, same assertions on the factors) and, like the reference, ends with the
        unscaled shell table.  The layer rows are rebuilt only if that table really changed."""
        p, t = self.params, self.tomography_params
        before = self.layers.rhos.copy()
        if self.tomography_type == "mass_of_earth":
            t.density_scale = p.density_scale.value.m_as("dimensionless")
            self.layers.scaling(scaling_array=t.density_scale)
        elif self.tomography_type == "mass_of_core_w_constrain":
            t.core_density_scale = p.core_density_scale.value.m_as("dimensionless")
            self.layers.scaling(scaling_array=t.scaling_array)
        else:
            t.core_density_scale = p.core_density_scale.value.m_as("dimensionless")
            t.innermantle_density_scale = p.innermantle_density_scale.value.m_as("dimensionless")
            t.middlemantle_density_scale = p.middlemantle_density_scale.value.m_as("dimensionless")
            self.layers.scaling(scaling_array=t.scaling_factor_array)
        self.scaled_rhos = self.layers.rhos.copy()   # what `scaling` alone produced (diagnostic)
        self.layers.setElecFrac(self.YeI, self.YeO, self.YeM)
        if not np.array_equal(before, self.layers.rhos):
            self._calc_layers()

    def compute_function(self):
        p = self.params
        ye = (p.YeI.value.m_as("dimensionless"), p.YeO.value.m_as("dimensionless"),
              p.YeM.value.m_as("dimensionless"))
        if ye != (self.YeI, self.YeO, self.YeM):
            self.YeI, self.YeO, self.YeM = ye
            self.layers.setElecFrac(*ye)
            self._calc_layers()
        params = self._matrices()
        if self.grid is not None:
            g = self.grid
            P_nu, P_nubar, pepmu = K.prob3_grid_planned(
                params, g["plan"], g["energy"], e_major=g["e_major"],
                out_nu=None if self.prob_tables is None else self.prob_tables[0],
                out_nubar=None if self.prob_tables is None else self.prob_tables[1],
                out_pepmu=self.pepmu)
            self.prob_tables, self.pepmu = (P_nu, P_nubar), pepmu
            # the tables are rewritten in place every evaluation: the containers keep the same tensors /
            # strided views (compacted on access only), only their bookkeeping moves (`refresh_dev`)
            views = getattr(self, "_views", None)
            if views is None or views["src"] is not pepmu or views["tabs"] is not P_nu:
                views = self._views = {"src": pepmu, "tabs": P_nu, "of": {}, "cont": None}
            if views["cont"] is None or len(views["cont"]) != len(self.data.containers):
                views["cont"] = [(0 if c["nubar"] > 0 else 1, int(c["flav"])) for c in self.data.containers]
            for container, (side, flav) in zip(self.data.containers, views["cont"]):
                v = views["of"].get((side, flav))
                if v is None:
                    v = views["of"][(side, flav)] = (self.prob_tables[side], pepmu[side, flav, :, 0],
                                                    pepmu[side, flav, :, 1])
                container.refresh_dev("probability", v[0])
                container.refresh_dev("prob_e", v[1])
                container.refresh_dev("prob_mu", v[2])
        else:
            events = self.calc_mode == "events"
            earth = self.layers.earth_struct() if events else None
            for container in self.data.containers:
                if events:
                    P = K.prob3_events(params, earth, container["nubar"],
                                       container.device("true_energy"),
                                       container.device("true_coszen"))
                else:
                    P = K.propagate_array(params, container["nubar"], container.device("true_energy"),
                                          container.device("densities"), container.device("distances"))
                container["probability"] = P
                container["prob_e"] = K.fill_probs(P, 0, container["flav"])   # prob3.py:593-608
                container["prob_mu"] = K.fill_probs(P, 1, container["flav"])

    # ---------------------------------------------------------------- apply
    def apply_function(self):
        for container in self.data:
            if not container.is_map or deferred.chain_open(container):
                deferred.osc(container, "nu_flux")
            else:
                w = container.device("weights")
                # (prob_e / prob_mu may be columns of the gather tables, published as views: read at their stride)
                K.apply_osc_weights(container.device("nu_flux"), container.device_view("prob_e"),
                                    container.device_view("prob_mu"), w)
                container["weights"] = w

def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    from pisa_amd.core.param import Param, ParamSet

    param_set = ParamSet([
        Param(name="detector_depth", value=10 * ureg.km, **param_kwargs),
        Param(name="prop_height", value=18 * ureg.km, **param_kwargs),
        Param(name="earth_model", value="osc/PREM_4layer.dat", **param_kwargs),
        Param(name="YeI", value=0.5, **param_kwargs),
        Param(name="YeO", value=0.5, **param_kwargs),
        Param(name="YeM", value=0.5, **param_kwargs),
        Param(name="theta12", value=33 * ureg.degree, **param_kwargs),
        Param(name="theta13", value=8 * ureg.degree, **param_kwargs),
        Param(name="theta23", value=50 * ureg.degree, **param_kwargs),
        Param(name="deltam21", value=8e-5 * ureg.eV ** 2, **param_kwargs),
        Param(name="deltam31", value=3e-3 * ureg.eV ** 2, **param_kwargs),
        Param(name="deltacp", value=180 * ureg.degree, **param_kwargs),
    ])
    return prob3(include_nlo=True, params=param_set)
