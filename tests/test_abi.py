"""CPU-side checks of the drop-in boundary: the shared library builds for
gfx950, loads, and exports every symbol include/pisa_hip.h declares.  No
compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from pisa_amd import _lib

    assert os.path.exists(_lib.LIB_PATH)
    return _lib


def test_header_symbols_all_exported(built):
    header = open(os.path.join(ROOT, "include", "pisa_hip.h")).read()
    declared = set(re.findall(r"\b(pisa_hip_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    handle = ctypes.CDLL(built.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), "symbol %s missing from libpisa_hip.so" % name
    # the binding covers the whole header
    assert declared == set(built.EXPORTED_SYMBOLS)
    # and DESIGN.md quotes the count that is true today (it had drifted: "74 entry points" against 84)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    quoted = re.search(r"`include/pisa_hip.h`, (\d+) entry points", design)
    assert quoted and int(quoted.group(1)) == len(declared), (quoted and quoted.group(1), len(declared))


def test_struct_layout_matches_header(built):
    # sizes implied by include/pisa_hip.h (LP64)
    assert ctypes.sizeof(built.Prob3Params) == (9 + 18 + 18 + 18 + 9) * 8 + 8
    assert ctypes.sizeof(built.Earth) == 8 + 8 + 3 * 64 * 8
    assert ctypes.sizeof(built.Binning) == 8 + 3 * 8 * 3
    assert ctypes.sizeof(built.Container) == 8 + 5 * 8 + 3 * 8 + 5 * 8 + 4 + 4 + 8 + 8 + 2 * 8 + 8 + 4 + 4


def test_status_strings_and_no_gpu_behaviour(built):
    lib = built.lib()
    assert lib.pisa_hip_strerror(0) == b"ok"
    assert b"120 layers" in lib.pisa_hip_strerror(-2)
    assert lib.pisa_hip_version() >= 100
    # argument validation happens before any device access
    p = built.Prob3Params()
    assert lib.pisa_hip_propagate_array(p, 1, None, None, None, 10, 200, 1, None, None) == -2
    assert lib.pisa_hip_propagate_array(p, 0, None, None, None, 10, 4, 1, None, None) == -1
    assert lib.pisa_hip_grid_plan_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, built):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libpisa_hip.so")
    with pytest.raises(ImportError):
        built.lib()


def test_header_is_plain_c_and_the_documented_call_sequence_type_checks(tmp_path):
    """include/pisa_hip.h compiles as C99 without any HIP header, and the per-evaluation call
    sequence shown in INTEGRATION.md matches the declared signatures (gcc -fsyntax-only)."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "seq.c"
    src.write_text(r'''
#include "pisa_hip.h"
int one_evaluation(const pisa_hip_earth *earth, const double *d_cz, const double *d_energy, int32_t n_e,
                   int32_t n_cz, int32_t max_layers, double *d_dens, double *d_dist,
                   const pisa_hip_prob3_params *par, const pisa_hip_container *cont, int32_t n_cont,
                   const pisa_hip_binning *calc_grid, const pisa_hip_binning *out_binning, int64_t n_bins,
                   double *d_P_nu, double *d_P_nubar, double *d_pepmu, int64_t *d_limbs, double *d_hist,
                   double *d_sumw2, const double *d_data, double *pinned_llh, int32_t *d_status,
                   int32_t *d_mstatus, void *stream) {
    pisa_hip_grid_plan *plan;
    int rc = pisa_hip_calc_layers(earth, d_cz, n_cz, max_layers, 0, d_dens, d_dist, d_status, stream);
    if (!rc) rc = pisa_hip_grid_plan_create(d_dens, d_dist, n_cz, max_layers, &plan);
    if (!rc) rc = pisa_hip_prob3_grid_planned(par, plan, d_energy, n_e, 1, d_P_nu, d_P_nubar, d_pepmu, stream);
    if (!rc) rc = pisa_hip_reweight_hist_acc(cont, n_cont, calc_grid, d_P_nu, d_P_nubar, d_pepmu, out_binning,
                                             d_limbs, d_status, stream);
    if (!rc) rc = pisa_hip_finalize_metric(d_limbs, n_cont, n_bins, d_hist, d_sumw2, PISA_HIP_METRIC_LLH, d_data,
                                           pinned_llh, d_status, d_mstatus, 1, stream);
    if (rc) (void)pisa_hip_strerror(rc);
    return rc ? rc : pisa_hip_grid_plan_destroy(plan);
}
''')
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
           "-I", os.path.join(root, "include"), str(src)]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_every_struct_of_the_header_has_the_layout_of_its_binding(built, tmp_path):
    """sizeof and every member offset of the POD blocks in include/pisa_hip.h, as gcc lays them out,
    against the ctypes classes of pisa_amd/_lib.py (the same member names on both sides)"""
    import json
    import subprocess

    pairs = [("pisa_hip_prob3_params", built.Prob3Params), ("pisa_hip_earth", built.Earth),
             ("pisa_hip_event_set", built.EventSet), ("pisa_hip_binning", built.Binning),
             ("pisa_hip_container", built.Container), ("pisa_hip_kde_info_t", built.KdeInfo),
             ("pisa_hip_flux_table", built.FluxTable), ("pisa_hip_fold_set", built.FoldSet),
             ("pisa_hip_barr_set", built.BarrSet), ("pisa_hip_barr_fold_set", built.BarrFoldSet),
             ("pisa_hip_kde_job", built.KdeJob), ("pisa_hip_evaluator_desc", built.EvaluatorDesc),
             ("pisa_hip_chain_set", built.ChainSet), ("pisa_hip_pack_set", built.PackSet)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pisa_hip.h"', 'int main(void) {', 'printf("{");']
    for k, (cname, cls) in enumerate(pairs):
        lines.append('printf("%s\\"%s\\": {\\"sizeof\\": %%zu", sizeof(%s));' % (", " if k else "", cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf(", \\"%s\\": %%zu", offsetof(%s, %s));' % (fname, cname, fname))
        lines.append('printf("}");')
    lines += ['printf("}\\n");', 'return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    out = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr      # a member missing on either side fails here
    got = json.loads(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)
    for cname, cls in pairs:
        assert got[cname]["sizeof"] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[cname][fname] == getattr(cls, fname).offset, (cname, fname)


def test_product_library_has_no_environment_switches(built):
    """The product library reads no environment variable of its own and has no switch that makes a kernel do
    less work (round-3 verdict): the names of the development probes reach the binary only in the
    -DPISA_DEV_PROBES build (libpisa_hip_dev.so).  Checked on the bytes of the shipped file: no string that
    IS an environment-variable name of ours, and none of the names the sources hand to PISA_DEV_INT/_LL/_STR.
    (Python-level options that remain and are documented: PISA_HIP_LIB, PISA_HIP_DIRECT_RCCL.)"""
    import glob
    import re

    from pisa_amd import _lib

    blob = open(_lib.LIB_PATH, "rb").read()
    assert not re.search(rb"(?<![A-Za-z0-9_ (*])PISA_HIP_[A-Z0-9_]+\x00", blob)
    names = set()
    src = os.path.join(ROOT, "pisa_amd", "csrc")
    for path in glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.hpp")):
        text = open(path).read()
        names.update(re.findall(r'PISA_DEV_(?:INT|LL|STR)\("([A-Z0-9_]+)"', text))
        if not path.endswith("common.hpp"):
            assert "getenv" not in text, path
    assert len(names) >= 15
    for n in names:
        assert ("PISA_HIP_" + n).encode() not in blob, n
