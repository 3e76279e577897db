"""Randomised differential run of the prob3 kernels against the CPU oracle (development tool, GPU box).  Every trial
draws an Earth model (PREM 4 / 10 / 12 / 59 layers), detector depth and production height, electron fractions, the
six oscillation parameters (both orderings, any delta_CP, sometimes a mixing angle of exactly 0 or 90 degrees, sometimes
dm21 = 0), a matter potential (standard, NLO, random hermitian NSI, vacuum), optionally neutrino decay or a real
long-range potential, and energies from 0.1 GeV to 10 TeV with zenith angles over the whole sky (exactly +-1, the
horizon and shell tangents included).  Compared with `oracle.propagate_array` on the oracle's own layers
(PROB3_RTOL 1e-10 / PROB3_ATOL 1e-14, the bar of the reference's own tests):
  * `pisa_hip_calc_layers` (densities, distances, bit for bit), `pisa_hip_propagate_array`,
  * the one-kernel grid form, the planned grid form and its gather tables,
  * the event-by-event kernel with in-kernel layers (contracted arithmetic: 1e-9 / 1e-13).
usage: fuzz_prob3.py [trials] [seed] [only]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc  # noqa: E402
from pisa_amd import _lib, kernels as K  # noqa: E402
from pisa_amd.stages.osc.layers import Layers  # noqa: E402
from pisa_amd.utils.resources import find_resource  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
bad = 0
t0 = time.time()


def close(a, b, rtol, atol):
    return bool(np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True))


for trial in range(trials):
    model = ["osc/PREM_4layer.dat", "osc/PREM_10layer.dat", "osc/PREM_12layer.dat", "osc/PREM_59layer.dat"][rs.randint(4)]
    depth, height = float(rs.uniform(0.5, 3.0)), float(rs.uniform(10.0, 30.0))
    ye = rs.uniform(0.40, 0.52, 3)
    angle = lambda lo, hi: float(np.deg2rad([0.0, 90.0][rs.randint(2)] if rs.rand() < 0.04 else rs.uniform(lo, hi)))  # noqa: E731
    t12, t13, t23 = angle(25, 40), angle(3, 15), angle(30, 60)
    dcp = float(np.deg2rad(rs.uniform(0, 360)))
    dm21 = 0.0 if rs.rand() < 0.04 else float(rs.uniform(5e-5, 1e-4))
    dm31 = float(rs.uniform(1e-3, 7e-3) * (1 if rs.rand() < 0.5 else -1))
    pot_kind = ["std", "nlo", "nsi", "vacuum"][rs.choice(4, p=[0.4, 0.15, 0.35, 0.1])]
    mat_pot = np.zeros((3, 3), complex)
    if pot_kind != "vacuum":
        mat_pot[0, 0] = 1.02 if pot_kind == "nlo" else 1.0
    if pot_kind == "nsi":
        a = rs.uniform(-0.3, 0.3, (3, 3)) + 1j * rs.uniform(-0.3, 0.3, (3, 3))
        mat_pot = mat_pot + 0.5 * (a + a.conj().T)
    decay = rs.rand() < 0.2
    mat_decay = np.zeros((3, 3), complex)
    if decay:
        mat_decay[2, 2] = -1j * 10 ** rs.uniform(-6, -3)
    lri = np.zeros((3, 3))
    if rs.rand() < 0.15:
        v = 10 ** rs.uniform(-15, -12.5)
        lri = np.diag(v * np.array([[1, -1, 0], [1, 0, -1], [0, 1, -1]][rs.randint(3)], dtype=float))
    n = int(rs.randint(1, 400))
    energy = 10 ** rs.uniform(-1, 4, n)
    cz = rs.uniform(-1, 1, n)
    special = [-1.0, 1.0, 0.0, -1e-9, 1e-9]
    for k in range(min(n, len(special))):
        if rs.rand() < 0.5:
            cz[k] = special[k]
    n_e, n_cz = int(rs.randint(1, 40)), int(rs.randint(1, 30))
    tag = "trial %d: %s depth %.2f height %.1f, %s%s%s, t12 %.3f t13 %.3f t23 %.3f dcp %.3f dm21 %.2e dm31 %.2e, n %d, grid %dx%d" % (
        trial, model.split("_")[1], depth, height, pot_kind, " decay" if decay else "", " lri" if lri.any() else "", t12, t13, t23, dcp,
        dm21, dm31, n, n_e, n_cz)
    if only is not None and trial != only:
        continue
    try:
        prem = np.loadtxt(find_resource(model))
        olay = orc.Layers(prem, depth, height)
        olay.setElecFrac(*ye)
        lay = Layers(model, depth, height)
        lay.setElecFrac(*ye)
        mix, dm = orc.mix_matrix(t12, t13, t23, dcp), orc.dm_matrix(dm21, dm31)
        flag = 1 if decay else -1
        params = _lib.make_prob3_params(dm, mix, mat_pot, flag, mat_decay, lri)
        problems = []
        # layers, bit for bit
        olay.calcLayers(cz)
        dens, dist = olay.density.reshape(n, -1), olay.distance.reshape(n, -1)
        nl, d_dens, d_dist = K.calc_layers(lay.earth_struct(), K.to_device(cz), dens.shape[1])
        if not (np.array_equal(d_dens.cpu().numpy(), dens) and np.array_equal(d_dist.cpu().numpy(), dist)):
            problems.append("layers")
        if dens.shape[1] > 120:
            # more layers than the reference's kernels hold (numba_osc_kernels.py:227): the reference cannot propagate
            # through this model either; the library must refuse, not overrun
            try:
                K.propagate_array(params, 1, K.to_device(energy), d_dens, d_dist)
                problems.append("%d layers accepted" % dens.shape[1])
            except Exception:  # pylint: disable=broad-except
                pass
            if problems:
                bad += 1
                print("MISMATCH", tag, "|", "; ".join(problems), flush=True)
            continue
        for nubar in (1, -1):
            want = orc.propagate_array(dm, mix, mat_pot, flag, mat_decay, lri, nubar, energy, dens, dist)
            got = K.propagate_array(params, nubar, K.to_device(energy), d_dens, d_dist).cpu().numpy()
            if not close(got, want, 1e-10, 1e-14):
                problems.append("propagate_array nubar=%d (%.2e)" % (nubar, np.nanmax(np.abs(got - want))))
            ev = K.prob3_events(params, lay.earth_struct(), nubar, K.to_device(energy), K.to_device(cz)).cpu().numpy()
            if not close(ev, want, 1e-9, 1e-13):
                problems.append("events nubar=%d (%.2e)" % (nubar, np.nanmax(np.abs(ev - want))))
        # grids
        e_nodes = np.sort(10 ** rs.uniform(-0.5, 3.5, n_e))
        cz_nodes = np.sort(rs.uniform(-1, 1, n_cz))
        olay.calcLayers(cz_nodes)
        gd, gl = olay.density.reshape(n_cz, -1), olay.distance.reshape(n_cz, -1)
        _, dd, dl = K.calc_layers(lay.earth_struct(), K.to_device(cz_nodes), gd.shape[1])
        ee, rows = np.repeat(e_nodes, n_cz), np.tile(np.arange(n_cz), n_e)
        want = {s: orc.propagate_array(dm, mix, mat_pot, flag, mat_decay, lri, s, ee, gd[rows], gl[rows]) for s in (1, -1)}
        p_nu, p_nubar = K.prob3_grid(params, K.to_device(e_nodes), dd, dl)
        plan = K.GridPlan(dd, dl)
        q_nu, q_nubar, pepmu = K.prob3_grid_planned(params, plan, K.to_device(e_nodes))
        for name, got, s in (("grid nu", p_nu, 1), ("grid nubar", p_nubar, -1), ("planned nu", q_nu, 1), ("planned nubar", q_nubar, -1)):
            g = got.cpu().numpy()
            if not close(g, want[s], 1e-10, 1e-13 if name.startswith("planned") else 1e-14):
                problems.append("%s (%.2e)" % (name, np.nanmax(np.abs(g - want[s]))))
        t = pepmu.cpu().numpy()                                     # [sign][flav][node][e, mu]
        for si, s in enumerate((1, -1)):
            for flav in range(3):
                if not (close(t[si, flav, :, 0], want[s][:, 0, flav], 1e-10, 1e-13) and close(t[si, flav, :, 1], want[s][:, 1, flav], 1e-10, 1e-13)):
                    problems.append("gather table sign %d flav %d" % (s, flav))
        if not decay:
            for s in (1, -1):
                if np.abs(want[s].sum(axis=2) - 1).max() > 1e-9:
                    problems.append("oracle rows not unitary?")
        if problems:
            bad += 1
            print("MISMATCH", tag, "|", "; ".join(problems), flush=True)
    except Exception as e:  # pylint: disable=broad-except
        bad += 1
        print("ERROR", tag, type(e).__name__, str(e)[:300], flush=True)
    if trial % 20 == 19:
        print("... %d trials, %d bad, %.0f s" % (trial + 1, bad, time.time() - t0), flush=True)
print("fuzz_prob3: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
