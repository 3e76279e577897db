import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; see oracle/pisa_oracle.c)."""
    from oracle import oracle as orc

    orc.build()
    return orc


BENCH_LINE_LIMIT = 4096


def bench_result(stdout, detail_path):
    """bench.py's contract as the driver reads it: the LAST stdout line is ONE compact strict-JSON object below 4 KB
    (round 5's 20 KB line was not parsed); the full result is in the detail file.  Returns (compact, detail)."""
    import json

    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert lines and lines[-1].startswith("{"), stdout[-2000:]
    assert sum(ln.startswith("{") for ln in lines) == 1, stdout[-2000:]     # (a launcher may print lines of its own before it)
    lines = lines[-1:]
    assert len(lines[0].encode()) < BENCH_LINE_LIMIT, len(lines[0])

    def no_constants(name):   # NaN / Infinity are not JSON
        raise ValueError("non-JSON constant %s in the bench line" % name)

    compact = json.loads(lines[0], parse_constant=no_constants)
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
    assert need <= set(compact), need - set(compact)
    assert {"workload", "events", "calc_grid", "out_bins", "parallelism"} <= set(compact["config"])
    with open(detail_path) as fh:
        detail = json.load(fh)
    assert detail["value"] == compact["value"] and detail["ms_per_step"] == compact["ms_per_step"]
    return compact, detail


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


# reference's own prob3 tolerance (numba_osc_tests.py:82: AC_KW)
PROB3_RTOL = 1e-10
PROB3_ATOL = 1e-14


DEV_LIB = os.path.join(ROOT, "pisa_amd", "libpisa_hip_dev.so")


def run_dev_case(case, *args, timeout=900):
    """one case of tests/dev_cases.py in a process of its own, on the development build of the library
    (-DPISA_DEV_PROBES): the product library has no run-time switches"""
    import subprocess

    src = os.path.join(ROOT, "pisa_amd", "csrc")
    # (built by __graft_entry__.build(); rebuilt here only if missing or older than a source)
    if not os.path.exists(DEV_LIB) or subprocess.call(["make", "-q", "-C", src, "dev"],
                                                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0:
        subprocess.check_call(["make", "-s", "-j8", "-C", src, "dev"])
    env = dict(os.environ, PISA_HIP_LIB=DEV_LIB, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, "-m", "tests.dev_cases", case] + [str(a) for a in args], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return res.stdout
