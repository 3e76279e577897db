"""Import shim that lets THIS build container execute the reference's own
Python kernels (read in place from /root/reference) without numba/pint.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/gen_golden.py`` to produce the
fixtures under ``tests/golden/`` (run in this container, where /root/reference exists; the
tests themselves read only the committed fixtures).  Nothing of the
reference is copied: the modules are imported by path and run as plain Python
after pre-seeding ``sys.modules`` with inert stand-ins for the compiled /
unit-system dependencies (SURVEY.md Appendix B recipe).

What becomes importable:
  pisa.utils.numba_tools, pisa.stages.osc.prob3numba.numba_osc_kernels,
  pisa.stages.osc.layers, osc_params, nsi_params, decay_params, lri_params,
  pisa.stages.flux.barr_simple (+ utils.barr_parameterization),
  pisa.core.translation (lookup_* / find_index bodies), pisa.utils.stats,
  pisa.utils.flux_weights (scipy/FITPACK splines; load_2d_table, calculate_2d_flux_weights).
"""
import importlib
import logging as _logging
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("PISA_REFERENCE_ROOT", "/root/reference")
REF_RESOURCES = os.path.join(REF_ROOT, "pisa_examples", "resources")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "pisa"))


def _decorator_factory(*args, **kwargs):
    """Stands in for numba.jit/njit/guvectorize in all call forms."""
    if len(args) >= 1 and callable(args[0]) and not isinstance(args[0], (list, str)):
        return args[0]

    def deco(func):
        return func

    return deco


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = _mod(name)
    m.__path__ = [path]
    return m


_installed = False


def install():
    """Pre-seed sys.modules; idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)

    # ---- numba ------------------------------------------------------------
    nb = _mod(
        "numba",
        jit=_decorator_factory,
        njit=_decorator_factory,
        guvectorize=_decorator_factory,
        vectorize=_decorator_factory,
        prange=range,
        typeof=type,
        float32=np.float32,
        float64=np.float64,
        complex64=np.complex64,
        complex128=np.complex128,
        int32=np.int32,
        int64=np.int64,
        uint32=np.uint32,
        uint64=np.uint64,
        SmartArray=None,
    )
    nb.cuda = types.SimpleNamespace(
        jit=_decorator_factory, local=types.SimpleNamespace(array=np.empty)
    )
    nb.__version__ = "0.0-stub"
    _mod("numba.cuda", jit=_decorator_factory)

    # ---- pisa package skeleton -------------------------------------------
    p = os.path.join(REF_ROOT, "pisa")
    pisa = _pkg("pisa", p)
    pisa.FTYPE = np.float64
    pisa.CTYPE = np.complex128
    pisa.ITYPE = np.int64
    pisa.TARGET = "cpu"
    pisa.PISA_NUM_THREADS = 1
    pisa.PISA_HIST_THREADING = "off"
    pisa.HASH_SIGFIGS = 12
    pisa.EPSILON = 10 ** (-12)
    pisa.ureg = None
    pisa.CACHE_DIR = "/tmp"
    for sub in ("utils", "stages", "core"):
        _pkg("pisa." + sub, os.path.join(p, sub))
    for sub in ("osc", "flux"):
        _pkg("pisa.stages." + sub, os.path.join(p, "stages", sub))
    _pkg(
        "pisa.stages.osc.prob3numba", os.path.join(p, "stages", "osc", "prob3numba")
    )

    # ---- leaf stubs -------------------------------------------------------
    eps = np.finfo(np.float64).eps

    def isscalar(x):
        return np.isscalar(x)

    def isbarenumeric(x):
        return isinstance(x, (int, float, np.ndarray, np.number))

    def recursive_equality(x, y, allclose_kw=None):
        return np.allclose(x, y)

    _mod(
        "pisa.utils.comparisons",
        ALLCLOSE_KW=dict(rtol=1e-12, atol=eps, equal_nan=True),
        FTYPE_PREC=eps,
        isscalar=isscalar,
        isbarenumeric=isbarenumeric,
        recursiveEquality=recursive_equality,
    )

    lg = _logging.getLogger("pisa_ref_shim")
    lg.trace = lg.debug

    class Levels:
        FATAL, ERROR, WARN, INFO, DEBUG, TRACE = range(6)

    _mod("pisa.utils.log", logging=lg, Levels=Levels, set_verbosity=lambda *a, **k: None)

    def from_file(fname, as_array=False, **kw):
        path = fname if os.path.isabs(fname) else os.path.join(REF_RESOURCES, fname)
        return np.loadtxt(path)

    _mod("pisa.utils.fileio", from_file=from_file)

    def find_resource(fname, fail=True):
        return fname if os.path.isabs(fname) else os.path.join(REF_RESOURCES, fname)

    def open_resource(fname, mode="r"):
        return open(find_resource(fname), mode)

    _mod("pisa.utils.resources", find_resource=find_resource, open_resource=open_resource)
    _mod("pisa.utils.profiler", profile=lambda f: f, line_profile=lambda f: f)
    _mod("pisa.utils.likelihood_functions")

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    _mod("pisa.core.param", Param=_Dummy, ParamSet=_Dummy)
    _mod("pisa.core.stage", Stage=_Dummy)
    _mod("pisa.core.binning", OneDimBinning=_Dummy, MultiDimBinning=_Dummy)
    _mod("fast_histogram")

    unp = types.SimpleNamespace(
        nominal_values=np.asarray, std_devs=lambda x: np.zeros_like(np.asarray(x))
    )
    unc = _mod("uncertainties", unumpy=unp)
    sys.modules["uncertainties.unumpy"] = unp
    _installed = True


def ref_module(name):
    """Import a reference module (by dotted name under ``pisa``)."""
    install()
    return importlib.import_module(name)


def kernels():
    return ref_module("pisa.stages.osc.prob3numba.numba_osc_kernels")


def layers_mod():
    return ref_module("pisa.stages.osc.layers")
