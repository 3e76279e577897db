"""Cases that need the DEVELOPMENT build of the library (`make -C pisa_amd/csrc dev` ->
pisa_amd/libpisa_hip_dev.so, compiled with -DPISA_DEV_PROBES): alternative kernel forms and launch shapes
selected through PISA_HIP_* environment variables.  The product library has none of these switches
(tests/test_abi.py), so each case runs in a process of its own whose PISA_HIP_LIB points at the development
build:  python -m tests.dev_cases <case> [args...]   (started by tests.conftest.run_dev_case).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def planned_variant(fused_amp):
    """the planned grid form with the layer matrices stored by stage AB and read back ("0") / one row
    per workgroup ("split"): equal to the direct grid kernel to rounding and to the reference goldens"""
    from pisa_amd import _lib as L
    from pisa_amd import kernels as K
    from tests.conftest import PROB3_ATOL, PROB3_RTOL, load_golden

    AC = dict(rtol=PROB3_RTOL, atol=PROB3_ATOL)
    g = load_golden("prob3_grid_prem12.npz")
    e, dens, dist = K.to_device(g["energy"]), K.to_device(g["densities"]), K.to_device(g["distances"])
    n_e, n_cz = len(g["energy"]), g["densities"].shape[0]
    os.environ["PISA_HIP_PROB3_FUSED_AMP"] = "0" if fused_amp == "0" else "1"
    os.environ["PISA_HIP_CHAIN_MODE"] = "split" if fused_amp == "split" else "packed"
    plan = K.GridPlan(dens, dist)
    for name in ("no", "io", "nsi", "decay"):
        p = L.make_prob3_params(g[name + "::dm"], g[name + "::mix"], g[name + "::mat_pot"],
                                int(g[name + "::decay_flag"]), g[name + "::mat_decay"], g[name + "::lri_pot"])
        for e_major in (True, False):
            nu, nubar, pepmu = K.prob3_grid(p, e, dens, dist, e_major=e_major, want_pepmu=True)
            nu2, nubar2, pepmu2 = K.prob3_grid_planned(p, plan, e, e_major=e_major)
            for a, b in ((nu, nu2), (nubar, nubar2), (pepmu, pepmu2)):
                assert float((a - b).abs().max()) < 3e-13
            got_nu, got_nubar = nu2.cpu().numpy(), nubar2.cpu().numpy()
            if e_major:
                got_nu, got_nubar = got_nu.reshape(n_e, n_cz, 3, 3), got_nubar.reshape(n_e, n_cz, 3, 3)
            else:
                got_nu = got_nu.reshape(n_cz, n_e, 3, 3).transpose(1, 0, 2, 3)
                got_nubar = got_nubar.reshape(n_cz, n_e, 3, 3).transpose(1, 0, 2, 3)
            np.testing.assert_allclose(got_nu, g[name + "::prob_nu"], err_msg=name, **AC)
            np.testing.assert_allclose(got_nubar, g[name + "::prob_nubar"], err_msg=name, **AC)


def launch_shape(n_events):
    """a single workgroup, fewer workgroups than containers, the default, many more than CUs and other
    workgroup sizes give the same bits -- for the 16-bit index form, the compact form, the
    reference-order form and the coordinate form"""
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=int(n_events), grid=(40, 30), out_binning="dragon", seed=11)
    p = wl.osc_params(theta23_deg=44.0)
    for kw in (dict(compact=True), dict(compact=True, index16=False), dict(compact=False), dict(indexed=False)):
        ref = None
        for blocks, threads in ((None, None), (1, None), (5, None), (64, None), (1500, None), (None, 256), (300, 512)):
            for key, v in (("PISA_HIP_HIST_BLOCKS", blocks), ("PISA_HIP_HIST_THREADS", threads)):
                if v is None:
                    os.environ.pop(key, None)
                else:
                    os.environ[key] = str(v)
            st = synthetic.DeviceState(wl, **kw)
            st.make_pseudo_data(wl.osc_params(), seed=0)
            llh = float(st.eval(p, "llh").item())
            st.check_status()
            h, s2 = st.finalize()
            got = (h.cpu().numpy().copy(), s2.cpu().numpy().copy(), llh)
            if ref is None:
                ref = got
                assert ref[0].sum() > 0
            else:
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2] == ref[2], \
                    (kw, blocks, threads)


CASES = {"planned_variant": planned_variant, "launch_shape": launch_shape}

if __name__ == "__main__":
    from pisa_amd import _lib

    assert _lib.LIB_PATH.endswith("libpisa_hip_dev.so"), _lib.LIB_PATH
    CASES[sys.argv[1]](*sys.argv[2:])
    print("dev case %s%r OK" % (sys.argv[1], tuple(sys.argv[2:])))
