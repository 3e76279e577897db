"""Synthetic workloads for tests, smoke and bench.

Events are drawn like the reference's toy generator
(pisa/stages/data/toy_event_generator.py:56-76): for each of the 12 containers
in `osc_example.cfg` order, from ONE RandomState(seed):
    true_energy = 10**(rand(n)*3) GeV, then true_coszen = rand(n)*2 - 1.
Builder-defined additions (the toy generator has no reco variables and unit
fluxes): reco_energy = true_energy*exp(N(0,0.2)), reco_coszen =
clip(true_coszen + N(0,0.15), -1, 1), pid ~ Bernoulli(0.3) from
RandomState(seed+1); power-law nu_flux, E-dependent weighted_aeff and uniform
initial_weights (toy_event_generator's `random=True` branch) so that no factor
of the weight chain is trivially 0 or 1.
"""
import numpy as np

from . import _lib
from .engine import GridSpec, HotPathEngine
from .stages.osc.layers import Layers
from .stages.osc.osc_params import OscParams

NAMES = ("nue_cc", "numu_cc", "nutau_cc", "nue_nc", "numu_nc", "nutau_nc",
         "nuebar_cc", "numubar_cc", "nutaubar_cc", "nuebar_nc", "numubar_nc", "nutaubar_nc")

# dragon_datarelease (settings/binning/IceCube_3y_oscillations.cfg:11-14), regularised:
# reco_energy log 8 bins, reco_coszen lin 8 bins, pid 2 bins
DRAGON = dict(mins=[np.log(5.62341325), -1.0, -0.5], maxs=[np.log(56.23413252), 1.0, 1.5],
              nbins=[8, 8, 2], log=[True, False, False])
# reco_energy x reco_coszen 10x10 (settings/binning/example.cfg:56-59 without pid)
EXAMPLE2D = dict(mins=[np.log(5.0), -1.0], maxs=[np.log(100.0), 1.0], nbins=[10, 10],
                 log=[True, False])
# a fine analysis binning (40 x 40 x 3 = 4800 bins): too large for LDS accumulators
FINE3D = dict(mins=[np.log(5.0), -1.0, -0.5], maxs=[np.log(100.0), 1.0, 2.5], nbins=[40, 40, 3],
              log=[True, False, False])
# settings/binning/example.cfg reco_binning (10 x 10 x 2): same bin populations as the pipeline's
# pid = -1 / +1 in edges [-1000, 0, 1000] with this module's pid = 0 / 1
EXAMPLE3D = dict(mins=[np.log(5.0), -1.0, -0.5], maxs=[np.log(100.0), 1.0, 1.5], nbins=[10, 10, 2],
                 log=[True, False, False])
BINNINGS = dict(dragon=DRAGON, example2d=EXAMPLE2D, fine3d=FINE3D, example3d=EXAMPLE3D)

LIVETIME_S = 2.5 * 365 * 86400.0  # 2.5 common_year (example.cfg aeff.livetime)


def flav_nubar(name):
    """toy_event_generator.py:61-67"""
    nubar = -1 if "bar" in name else 1
    flav = 0
    if "mu" in name:
        flav = 1
    if "tau" in name:
        flav = 2
    return flav, nubar


def aeff_scale_for(name, aeff_scale=1.0, livetime_s=LIVETIME_S, nutau_cc_norm=1.0,
                   nutau_norm=1.0, nu_nc_norm=1.0):
    """aeff.py:78-86"""
    scale = aeff_scale * livetime_s
    if name in ("nutau_cc", "nutaubar_cc"):
        scale *= nutau_cc_norm
    if "nutau" in name:
        scale *= nutau_norm
    if "nc" in name:
        scale *= nu_nc_norm
    return scale


def make_events(n_per, seed=0, names=NAMES):
    rs = np.random.RandomState(seed)
    rr = np.random.RandomState(seed + 1)
    out = []
    for name in names:
        flav, nubar = flav_nubar(name)
        e = np.power(10, rs.rand(n_per) * 3)
        cz = rs.rand(n_per) * 2 - 1
        reco_e = e * np.exp(rr.randn(n_per) * 0.2)
        reco_cz = np.clip(cz + rr.randn(n_per) * 0.15, -1, 1)
        pid = (rr.rand(n_per) < 0.3).astype(np.float64)
        w0 = rr.rand(n_per)
        flux_mu = 1e4 * e ** -2.7 * (1 + 0.5 * cz ** 2)
        flux = np.stack([flux_mu * (0.5 - 0.2 * cz), flux_mu], axis=1)
        aeff = 1e-9 * e ** 1.5 * (0.5 + rr.rand(n_per))
        out.append(dict(name=name, flav=flav, nubar=nubar, true_energy=e, true_coszen=cz,
                        reco_energy=reco_e, reco_coszen=reco_cz, pid=pid, nu_flux=flux,
                        weighted_aeff=aeff, initial_weights=w0))
    return out


def make_events_device(n_per, seed=0, names=NAMES, device=None):
    """`make_events` with the columns generated IN HBM (torch generator on the device, one seed per
    container): the same distributions, not the same numbers.  For samples that should never exist on the
    host -- config C5 at its full size is 1e8 events, ~10 GB of host columns and a minute of numpy otherwise."""
    import torch

    device = device or torch.device("cuda", torch.cuda.current_device())
    out = []
    for k, name in enumerate(names):
        flav, nubar = flav_nubar(name)
        g = torch.Generator(device=device)
        g.manual_seed(1_000_003 * int(seed) + k)

        def rand():
            return torch.rand(n_per, generator=g, device=device, dtype=torch.float64)

        def randn():
            return torch.randn(n_per, generator=g, device=device, dtype=torch.float64)

        e = torch.pow(10.0, rand() * 3)
        cz = rand() * 2 - 1
        reco_e = e * torch.exp(randn() * 0.2)
        reco_cz = torch.clamp(cz + randn() * 0.15, -1, 1)
        pid = (rand() < 0.3).to(torch.float64)
        w0 = rand()
        flux_mu = 1e4 * e ** -2.7 * (1 + 0.5 * cz ** 2)
        flux = torch.stack([flux_mu * (0.5 - 0.2 * cz), flux_mu], dim=1)
        aeff = 1e-9 * e ** 1.5 * (0.5 + rand())
        out.append(dict(name=name, flav=flav, nubar=nubar, true_energy=e, true_coszen=cz,
                        reco_energy=reco_e, reco_coszen=reco_cz, pid=pid, nu_flux=flux,
                        weighted_aeff=aeff, initial_weights=w0))
    return out


class Workload:
    """Host-side description of one synthetic pipeline (inputs only)."""

    def __init__(self, n_events=1200000, grid=(200, 100), out_binning="dragon", seed=0,
                 earth_model="osc/PREM_12layer.dat", detector_depth=2.0, prop_height=20.0,
                 ye=(0.4656, 0.4656, 0.4957), on_device=False):
        self.n_per = int(n_events) // len(NAMES)
        self.n_events = self.n_per * len(NAMES)
        self.grid = GridSpec((1.0, 1000.0), grid[0], (-1.0, 1.0), grid[1], energy_first=True)
        self.ob = BINNINGS[out_binning]
        self.out_binning = _lib.make_binning(self.ob["mins"], self.ob["maxs"], self.ob["nbins"])
        self.n_bins = int(np.prod(self.ob["nbins"]))
        self.layers = Layers(earth_model, detector_depth, prop_height)
        self.layers.setElecFrac(*ye)  # (YeI, YeO, YeM)
        self.on_device = bool(on_device)
        self.events = make_events_device(self.n_per, seed) if on_device else make_events(self.n_per, seed)
        cols = ("reco_energy", "reco_coszen", "pid")[: len(self.ob["nbins"])]
        for ev in self.events:
            if on_device:
                import torch

                ev["sample"] = [torch.log(ev[c]) if lg else ev[c] for c, lg in zip(cols, self.ob["log"])]
            else:
                ev["sample"] = [np.log(ev[c]) if lg else ev[c] for c, lg in zip(cols, self.ob["log"])]
            ev["scale"] = aeff_scale_for(ev["name"])

    def osc_params(self, theta23_deg=42.0, dm31=2.457e-3, theta12_deg=33.48, theta13_deg=8.5,
                   deltacp_deg=0.0, dm21=7.5e-5, mat_pot=None, decay_alpha3=None):
        """nufit v2.0 NH nominal of osc_example.cfg (settings/osc/nufitv20.cfg)."""
        o = OscParams()
        o.theta12, o.theta13, o.theta23 = (np.deg2rad(theta12_deg), np.deg2rad(theta13_deg),
                                           np.deg2rad(theta23_deg))
        o.deltacp = np.deg2rad(deltacp_deg)
        o.dm21, o.dm31 = dm21, dm31
        if mat_pot is None:
            mat_pot = _STD_POT  # prob3.py:539-543
        mat_decay = _ZERO_C
        flag = -1
        if decay_alpha3 is not None:
            mat_decay = np.zeros((3, 3), np.complex128)
            mat_decay[2, 2] = 0 - decay_alpha3 * 1j
            flag = 1
        dm, mix = o.dm_matrix, o.mix_matrix_complex      # each assembled once per point (a fit loop calls this)
        self.last_matrices = dict(dm=dm, mix=mix, mat_pot=mat_pot, decay_flag=flag, mat_decay=mat_decay,
                                  lri_pot=_ZERO_R)
        return _lib.make_prob3_params(dm, mix, mat_pot, flag, mat_decay, _ZERO_R)


_STD_POT = np.diag([1.0, 0.0, 0.0]).astype(np.complex128)
_ZERO_C = np.zeros((3, 3), np.complex128)
_ZERO_R = np.zeros((3, 3))
for _m in (_STD_POT, _ZERO_C, _ZERO_R):
    _m.setflags(write=False)


class DeviceState(HotPathEngine):
    """Engine loaded with a synthetic workload (rank's shard of the events)."""

    def __init__(self, wl, rank=0, world_size=1, group=None, indexed=True, planned=True,
                 packed=True, sort_events=True, osc_mode="grid", drop_unbinned=False, compact=False,
                 lds_order=True, index16=True, node_flux=False, block_order=True, points=None, time_setup=False):
        super().__init__(wl.events, wl.grid, wl.out_binning, wl.layers.earth_struct(),
                         wl.layers.max_layers, rank=rank, world_size=world_size, group=group,
                         indexed=indexed, planned=planned, packed=packed, sort_events=sort_events,
                         osc_mode=osc_mode, drop_unbinned=drop_unbinned, compact=compact,
                         lds_order=lds_order, index16=index16, node_flux=node_flux, block_order=block_order,
                         points=points, time_setup=time_setup)
        self.wl = wl

    def make_pseudo_data(self, params, seed=0):
        """Poisson-fluctuated nominal template (as analysis.py:2705-2707)."""
        self.accumulate(params)
        self.allreduce()
        self.finalize()
        total = self.ws.hist.sum(dim=0).cpu().numpy()
        data = np.random.RandomState(seed).poisson(total).astype(np.float64)
        self.set_data(data)
        return data
