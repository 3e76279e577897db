"""pisa_amd -- MI355X-native implementation of PISA's per-event hot path
(prob3 oscillation -> reweight -> weighted histogram -> LLH) behind PISA's
Stage / Container / Pipeline interface.  See DESIGN.md.

FTYPE is fixed to float64 (PISA's default, pisa/__init__.py:152-179); the only
compute target is the HIP library (there is no CPU fallback).
"""
import numpy as np

FTYPE = np.float64
CTYPE = np.complex128
ITYPE = np.int64
TARGET = "hip"
HASH_SIGFIGS = 12      # significant figures kept when values are normalised for hashing (pisa/__init__.py:277)

__version__ = "0.1.0"


_WARM = {"thread": None, "done": False, "ms": None, "error": None}


def warm_up(background=False):
    """Pay the runtime's first-use costs now instead of inside the first `Pipeline(cfg)` / `HotPathEngine`: the code
    objects of this library and of the torch kernels the set-up uses are loaded at their first launch, the copy engines'
    staging buffers are made at the first large host-to-device copy -- 0.2-0.5 s on a fresh process against 24 ms for a
    second engine of 1e7 events (bench.py `setup`).  Runs one small engine (12 x 100 000 synthetic events: upload,
    digitisation, resident order, packing, oscillation plan, one evaluation -- the same calls, large enough for the
    sorts to take the kernels they take at full size) and one 32 MB pageable upload.  `background=True`: on a daemon thread, so that it runs beside the caller's
    own start-up work (reading event files, parsing the cfg); `warm_up_wait()` joins it.  Idempotent; needs a HIP device
    (there is no CPU fallback in this package: without a device it raises like everything else)."""
    import threading

    if _WARM["done"] or _WARM["thread"] is not None:
        return

    def work():
        import time

        import torch

        t0 = time.perf_counter()
        try:
            from pisa_amd import synthetic

            dev = torch.device("cuda", torch.cuda.current_device())
            wl = synthetic.Workload(n_events=12 * 100_000, grid=(8, 8), out_binning="dragon", seed=1)
            st = synthetic.DeviceState(wl, compact=True)
            st.make_pseudo_data(wl.osc_params(), seed=0)
            st.eval_host(wl.osc_params(theta23_deg=44.0), "llh")
            st.maps()
            big = torch.from_numpy(np.zeros(4 << 20, dtype=np.float64)).to(dev)     # the pageable-copy path at a real size
            torch.cuda.synchronize()
            del st, big
        except Exception as exc:       # reported by warm_up_wait(); the first real engine raises the real error
            _WARM["error"] = exc
        _WARM["ms"] = 1e3 * (time.perf_counter() - t0)
        _WARM["done"] = True

    if background:
        import torch

        device = torch.cuda.current_device()

        def run():
            torch.cuda.set_device(device)
            work()

        _WARM["thread"] = threading.Thread(target=run, name="pisa_amd-warm-up", daemon=True)
        _WARM["thread"].start()
    else:
        work()
        if _WARM["error"] is not None:
            raise _WARM["error"]


def warm_up_wait():
    """joins a `warm_up(background=True)`; returns the milliseconds it took (None: never started)"""
    t = _WARM["thread"]
    if t is not None:
        t.join()
        _WARM["thread"] = None
    if _WARM["error"] is not None:
        err, _WARM["error"] = _WARM["error"], None
        raise err
    return _WARM["ms"]
