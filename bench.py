#!/usr/bin/env python
"""Headline benchmark: pipeline evals/sec (osc + reweight + hist + LLH) on 1e7
synthetic MC events (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one template evaluation with fresh oscillation parameters (every
stage recomputes, no memo hit; protocol of
pisa/scripts/benchmark_pipeline_performance.py:196-223):
    prob3 on the 200x100 (E, coszen) calc grid for nu and nubar
    fused grid->event lookup + flux*osc*aeff reweight + 8x8x2 histogram (+sumw2)
      over all 12 containers
    [integer all-reduce of the histogram limbs if N > 1]
    fixed point -> fp64 maps, Poisson LLH against pseudo-data
and the host reads the LLH back (a fit loop needs it to choose the next point).

N > 1 (north star): ONE 1e7-event sample, its events sharded over the N GPUs, the int64
histogram limbs all-reduced over RCCL -- strong scaling; `value` is the evaluation rate of that
one sample.  The weak-scaling rate (every rank holds 1e7 events of its own, N samples per step)
is measured in the same run and reported as `weak_value`.

Rank 0 prints, as its LAST stdout line, ONE compact JSON object below 4 KB: the contract's fields, `roofline`,
`cpu_baseline`, the LLH gate and one number per leg (`compact_line`).  The full result goes to `bench_detail.json`
beside this script (`--detail-out`) and, marked `bench_detail `, to stderr.  Besides the headline the full result
carries, at N = 1, the `legs`:
bounded extra measurements of the same hot path (larger-than-L3 sample, the reference-order and
the coordinate-form kernels, a 4 800-bin output binning, flux systematics moving every evaluation
with the flux per event and on the oscillation grid, the evaluation through the Pipeline/cfg
boundary, the published IceCube 3-year analysis through DistributionMaker / metric_total,
event-by-event oscillation for configs C2 / C5, the KDE stage for config C3) and the CPU baseline
at one thread and at all cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
L3_BYTES = 256 * 2 ** 20
ALL_LEGS = ("multi_point", "point_parallel", "fit_c4_engine", "fit_c4", "osc_example_c1", "l3_exceeding", "exact_association", "coordinate_form", "fine_binning",
            "update_flux", "node_flux", "pipeline_boundary", "icecube3y_boundary", "events_c2", "events_c2_decay", "events_c5", "events_c5_full", "kde_c3")
# the legs that also run with N > 1 (every rank takes part: configs C4 and C5, the multi-point sweep)
DIST_LEGS = ("multi_point", "point_parallel", "fit_c4_engine", "fit_c4", "events_c5")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--events", type=float, default=1e7)
    ap.add_argument("--grid", default="200x100", help="calc grid n_E x n_coszen")
    ap.add_argument("--binning", default="dragon", choices=["dragon", "example2d", "fine3d"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the HIP-event measurement of the dominant kernel (use under rocprofv3 --pmc)")
    ap.add_argument("--coordinate-form", action="store_true",
                    help="headline with the SURVEY 8(d)-shaped kernel: event coordinates binned on the fly "
                         "(72 B/event) instead of the pre-digitised index columns")
    ap.add_argument("--exact-association", action="store_true",
                    help="headline with the 40 B/event columns (initial_weights, weighted_aeff, nu_flux kept "
                         "separate, the reference's operation order)")
    ap.add_argument("--wide-index", action="store_true",
                    help="compact form with 32-bit node and bin indices (24 B/event)")
    ap.add_argument("--point-hybrid", action="store_true",
                    help="leg point_parallel: also measure G = N / 2 groups of two shards (sub-groups, per-group RCCL communicators)")
    ap.add_argument("--legs", default="all",
                    help="comma list of extra measurements at N = 1 (%s), 'all' or 'none'" % ", ".join(ALL_LEGS))
    ap.add_argument("--no-batch-probe", action="store_true", help="(accepted for old command lines: the probe is off by default)")
    ap.add_argument("--batch-probe", action="store_true",
                    help="informational: stream-overlapped evaluation of independent points (`eval_batch`, superseded by the "
                         "multi-point sweep `batched_evals_per_s*`); runs last -- its high-priority stream stays alive in "
                         "torch's pool and takes one of the device's four hardware queues")
    ap.add_argument("--no-drop-probe", action="store_true",
                    help="skip the informational second engine without the events outside the binning")
    ap.add_argument("--weak-scaling", action="store_true",
                    help="N > 1: make the weak-scaling rate the headline `value` (default: strong, as the "
                         "north star words it; the other one is reported beside it either way)")
    ap.add_argument("--strong-scaling", action="store_true", help="(default for N > 1; kept for compatibility)")
    ap.add_argument("--force-dist", action="store_true",
                    help="test aid: take the N > 1 code path (RCCL process group, limb all-reduce, barriers, "
                         "max over ranks) with the ranks that are there, e.g. one rank under "
                         "torch.distributed.run on a single-GPU box")
    ap.add_argument("--detail-out", default=None,
                    help="where the full result (every leg, thread scan, LLH referee) is written; default bench_detail.json "
                         "beside this script ('-': nowhere).  The LAST stdout line is the compact contract line either way")
    ap.add_argument("--c4-gate-points", type=int, default=50,
                    help="config C4's seeded (theta23, dm31) points put through the LLH gate against the oracle after the timed "
                         "regions (with the CPU baseline; 0: none)")
    ap.add_argument("--no-warm-up", action="store_true",
                    help="do not call pisa_amd.warm_up() beside the generation of the sample: `setup.first_in_process` then includes "
                         "the runtime's first-use costs (code objects, staging buffers)")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--min-timed-s", type=float, default=0.2,
                    help="repeat the block of --steps timed steps until this much timed work has been seen "
                         "(0: exactly one block)")
    ap.add_argument("--event-order", default="auto", choices=["auto", "node", "bin", "part"],
                    help="resident event order: sorted by calc-grid node, or by (output bin, node)")
    return ap.parse_args(argv)


COMPACT_LIMIT = 4096   # bytes: the LAST stdout line (what the driver parses) stays below this


def _r(x, digits=6):
    """floats to `digits` significant digits (the compact line only; the detail file keeps every bit)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def compact_line(out, detail_path=None):
    """The contract's fields + `roofline` + `cpu_baseline` + one number per leg: what rank 0 prints as its
    LAST stdout line.  Everything else of `out` (the legs in full, the thread scan of the CPU baseline, the
    LLH referee's report) lives in the detail file."""
    legs = out.get("legs") or {}

    def leg(name, *path):
        v = legs.get(name)
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
        return v

    cfg = out["config"]
    rf = out["roofline"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype")}
    line["data"] = "synthetic"
    line["config"] = {"workload": "%d events/12 containers, prob3 %dx%d PREM-12 grid (nu+nubar), fused lookup+reweight+%s hist+sumw2, "
                                  "Poisson LLH read back every eval, %d B/event"
                                  % (cfg["events"], cfg["calc_grid"][0], cfg["calc_grid"][1],
                                     "x".join(str(b) for b in cfg["out_bins"]), rf["bytes_per_event"]),
                      "events": cfg["events"], "calc_grid": cfg["calc_grid"], "out_bins": cfg["out_bins"],
                      "parallelism": "events sharded over %d GPU(s), int64 limb all-reduce (RCCL)" % out["n_gpus"]}
    bt = rf.get("by_kernel_trace") or {}
    line["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_event",
                                               "events_per_launch", "avg_launch_ms", "fits_l3", "traffic", "frac_beyond_l3")}
    line["roofline"]["by_kernel_trace"] = {"frac": bt.get("frac"), "avg_launch_us": bt.get("avg_launch_us")} if bt else None
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "cpu_model", "oracle_llh", "device_llh",
                                                       "llh_rel_diff")}
        line["cpu_baseline"]["sample"] = "all %d events + full grid, nothing scaled" % cfg["events"]
        line["cpu_baseline"]["single_thread"] = (cb.get("single_thread") or {}).get("value")
    gate = out.get("llh_gate")
    if gate:
        line["llh_gate"] = {"pure_1e-10_relative_met": gate.get("pure_1e-10_relative_met"), "applied": gate.get("applied"),
                            "referee_met": (gate.get("referee") or {}).get("met")}
    for k in ("strong_value", "weak_value", "allreduce_ms", "nccl_comm_count", "last_llh", "llh_bits_identical", "setup_ms",
              "topology", "point_parallel_evals_per_s", "batched_evals_per_s3", "batched_evals_per_s9", "timed_blocks",
              "hooks_used"):
        if out.get(k) is not None:
            line[k] = out[k]
    if out.get("llh_bits_per_rank"):
        line["llh_bits"] = out["llh_bits_per_rank"][0]
    line["setup_first_ms"] = ((out.get("setup") or {}).get("first_in_process") or {}).get("wall_ms")
    line["phase_ms"] = {k: v for k, v in (out.get("phase_ms") or {}).items() if k != "events_this_rank" and v is not None}
    summary = {
        "kde_c3_ms": leg("kde_c3", "ms_per_step"), "kde_c3_frac": leg("kde_c3", "roofline", "frac"),
        "kde_c3_launches": leg("kde_c3", "launches_per_evaluation"),
        "kde_c3_default_tol_ms": leg("kde_c3", "default_tol", "ms_per_step"),
        "events_c2_ms": leg("events_c2", "ms_per_step"), "events_c2_frac": leg("events_c2", "roofline", "frac"),
        "events_c2_decay_ms": leg("events_c2_decay", "ms_per_step"),
        "events_c5_ms": leg("events_c5", "ms_per_step"),
        "events_c5_full_ms": leg("events_c5_full", "ms_per_step"), "events_c5_full_frac": leg("events_c5_full", "roofline", "frac"),
        "coordinate_form_frac": leg("coordinate_form", "roofline", "frac"),
        "exact_association_frac": leg("exact_association", "roofline", "frac"),
        "fine_binning_frac": leg("fine_binning", "roofline", "frac"), "fine_binning_ms": leg("fine_binning", "ms_per_step"),
        "pipeline_boundary_evals_per_s": leg("pipeline_boundary", "evals_per_s"),
        "pipeline_boundary_over_engine": leg("pipeline_boundary", "boundary_over_engine"),
        "icecube3y_evals_per_s": leg("icecube3y_boundary", "all_free", "evals_per_s"),
        "osc_example_c1_ms": leg("osc_example_c1", "ms_per_step"),
        "fit_c4_evals_per_s": leg("fit_c4", "stencil_in_one_sweep", "evals_per_s"),
        "fit_c4_same_history": leg("fit_c4", "same_history"),
        "update_flux_ms": leg("update_flux", "ms_per_step"),
        # informational: the same evaluations with the events outside the output binning (which deposit nothing) not resident
        "unbinned_dropped_evals_per_s": (out.get("unbinned_events_dropped") or {}).get("evals_per_s"),
        "unbinned_dropped_events_resident": (out.get("unbinned_events_dropped") or {}).get("events_resident"),
        "c4_llh_gate": None if not out.get("c4_llh_gate") else {k: out["c4_llh_gate"].get(k) for k in
                                                                 ("points", "pure_1e-10_met", "all_met", "max_fp64_rel_diff",
                                                                  "max_maps_rel_diff_extended", "max_device_over_eps_rms")},
    }
    line["legs_summary"] = {k: v for k, v in summary.items() if v is not None}
    line["legs_run"] = sorted(k for k, v in legs.items() if v is not None)
    errs = sorted(k for k, v in legs.items() if isinstance(v, dict) and "error" in v)
    if errs:
        line["legs_failed"] = errs
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
    line = _r(line)
    line["value"], line["ms_per_step"] = out["value"], out["ms_per_step"]          # every bit of the headline and the LLHs
    for k in ("achieved", "frac", "avg_launch_ms"):                                 # (frac == achieved / peak to the last bit)
        v = rf.get(k)
        line["roofline"][k] = v if (isinstance(v, (int, float)) and v == v and abs(v) != float("inf")) else None
    for k in ("last_llh",):
        if k in line:
            line[k] = out[k]
    if cb:
        for k in ("oracle_llh", "device_llh"):
            line["cpu_baseline"][k] = cb.get(k)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= COMPACT_LIMIT:     # never let the line outgrow the driver again: shed the optional parts
        for k in ("legs_summary", "phase_ms", "legs_run", "llh_gate"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < COMPACT_LIMIT:
                break
    return text


HOOKS_ENV = "PISA_BENCH_HOOKS"   # tests only: "module:function" returning the `hooks` dict of main()


def _hooks_from_env():
    spec = os.environ.get(HOOKS_ENV)
    if not spec:
        return None
    import importlib

    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn)()


def launch_ranks(args, argv, standin):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process -- which has NOT
    touched the GPU and never will -- starts `torch.distributed.run` with N fresh rank processes on this
    very command line, relays rank 0's JSON line on stdout (everything else the ranks print goes to
    stderr) and returns the launcher's exit code.  `torch.cuda.device_count()` does not initialise HIP
    on this image; the ranks check their own device again."""
    import socket
    import subprocess

    if not standin:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            print("bench.py: --gpus %d asked for, %d HIP device(s) visible on this node" % (args.gpus, have),
                  file=sys.stderr)
            return 3
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, cwd=ROOT)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks ended without a result line", file=sys.stderr)
        return 4
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def param_list(wl, n, **kw):
    """fixed seeded scan of (theta23, dm31) over the ranges of SURVEY 8d (C4)"""
    import numpy as np

    rs = np.random.RandomState(2024)
    out = []
    for _ in range(n):
        out.append(wl.osc_params(theta23_deg=31.0 + 28.0 * rs.rand(), dm31=1e-3 + 6e-3 * rs.rand(), **kw))
    return out


def param_points(wl, n, **kw):
    """the points of `param_list`, one at a time, each with the matrices it was made from: (params, matrices)"""
    import numpy as np

    rs = np.random.RandomState(2024)
    for _ in range(n):
        p = wl.osc_params(theta23_deg=31.0 + 28.0 * rs.rand(), dm31=1e-3 + 6e-3 * rs.rand(), **kw)
        yield p, dict(wl.last_matrices)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """logical CPUs of the affinity mask, one per physical core (first SMT sibling)"""
    avail = sorted(os.sched_getaffinity(0))
    firsts = set()
    for cpu in avail:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as fh:
                sib = fh.read().strip().replace("-", ",").split(",")[0]
            firsts.add(int(sib))
        except (OSError, ValueError):
            firsts.add(cpu)
    return max(1, len(firsts)), len(avail)


def sockets_of_team(n_threads):
    """physical packages the first `n_threads` places of OMP_PLACES=cores / OMP_PROC_BIND=close fall on
    (places are the cores of the affinity mask in ascending order)"""
    firsts = {}
    for cpu in sorted(os.sched_getaffinity(0)):
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
            with open(base + "thread_siblings_list") as fh:
                first = int(fh.read().strip().replace("-", ",").split(",")[0])
            with open(base + "physical_package_id") as fh:
                firsts.setdefault(first, int(fh.read().strip()))
        except (OSError, ValueError):
            firsts.setdefault(cpu, 0)
    pk = [firsts[c] for c in sorted(firsts)][: max(1, n_threads)]
    return sorted(set(pk))


def llh_gate(data, lam, device_llh, oracle_llh, device_hist=None):
    """Which gate the device LLH of the line's last point meets against the oracle's.  `pure_1e-10_relative_met` is the
    headline pass / fail field: the north star's 1e-10 on the two fp64 numbers.  Where it is not met -- the reference's
    formula  llh = sum_b k ln(lam) - lam - (k ln k - k)  (stats.py:169-253) cancels terms orders of magnitude larger than the
    total, so two correct fp64 evaluations differ by ulps of the TERMS -- the extended-precision referee
    (`oracle/referee.py`, round 5; round 4: a floor of 8 eps sum|terms|) says whether the MAPS agree to 1e-10 (the formula in
    np.longdouble on the device's and the oracle's summed map) and each fp64 value is a correctly rounded evaluation (within
    2 eps sum|terms| of the extended value on its own map)."""
    import numpy as np

    diff = abs(device_llh - oracle_llh)
    pure = diff <= 1e-10 * abs(oracle_llh)
    out = {"abs_diff": diff, "pure_1e-10_relative_met": bool(pure), "applied": "1e-10 relative" if pure else "NONE MET"}
    if device_hist is not None:
        from oracle.referee import llh_referee

        lam_dev = np.asarray(device_hist, dtype=np.float64).reshape(-1, np.asarray(lam).size).sum(axis=0)
        ref = llh_referee(data, lam_dev, lam, device_llh, oracle_llh)
        out["referee"] = ref
        out["applied"] = ref["applied"]
    return out


def cpu_baseline_subprocess(args, n_e, n_cz, data, matrices, device_llh, device_hist=None, c4=None):
    """`cpu_baseline` in a process of its own: its OpenMP team is pinned one thread per core
    (OMP_PLACES=cores, OMP_PROC_BIND=close), which must not reach this process -- the runtime would pin
    the main thread as well and every helper thread the HIP runtime starts afterwards inherits that mask
    (measured: the Pipeline-boundary legs went from 126 to 223 us per evaluation)."""
    import subprocess
    import tempfile

    import numpy as np

    with tempfile.TemporaryDirectory(prefix="pisa_cpu_baseline_") as tmp:
        path = os.path.join(tmp, "in.npz")
        np.savez(path, data=data, device_llh=device_llh, events=int(args.events), grid=[n_e, n_cz],
                 device_hist=np.zeros(0) if device_hist is None else np.asarray(device_hist, dtype=np.float64),
                 cores=list(physical_cores()),     # counted here: the worker's main thread is pinned
                 **{"m_" + k: np.asarray(v) for k, v in matrices.items()},
                 **({} if not c4 else dict(c4_llh=np.asarray(c4["llh"]), c4_lam=np.asarray(c4["lam"]),
                                           **{"c4m_" + k: np.stack([np.asarray(m[k]) for m in c4["matrices"]])
                                              for k in c4["matrices"][0]})))
        env = dict(os.environ, OMP_PLACES="cores", OMP_PROC_BIND="close")
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", path,
                              "--binning", args.binning], env=env, stdout=subprocess.PIPE, check=True)
    return json.loads(res.stdout.decode().strip().splitlines()[-1])


def cpu_baseline_worker(path, binning):
    import numpy as np

    from pisa_amd import synthetic

    z = np.load(path)
    wl = synthetic.Workload(n_events=int(z["events"]), grid=tuple(int(v) for v in z["grid"]), out_binning=binning,
                            seed=0)
    m = {k[2:]: z[k] for k in z.files if k.startswith("m_")}
    m["decay_flag"] = int(m["decay_flag"])
    dev_hist = z["device_hist"] if "device_hist" in z.files and z["device_hist"].size else None
    c4 = None
    if "c4_llh" in z.files:
        keys = [k[4:] for k in z.files if k.startswith("c4m_")]
        mats = []
        for i in range(len(z["c4_llh"])):
            mi = {k: z["c4m_" + k][i] for k in keys}
            mi["decay_flag"] = int(mi["decay_flag"])
            mats.append(mi)
        c4 = {"llh": z["c4_llh"], "lam": z["c4_lam"], "matrices": mats}
    print(json.dumps(cpu_baseline(wl, z["data"], m, float(z["device_llh"]), tuple(int(v) for v in z["cores"]), dev_hist, c4)))


def cpu_baseline(wl, data, matrices, device_llh, cores_logical=None, device_hist=None, c4=None):
    """The oracle (C restatement of the reference algorithms) timed on this box's host cores on the
    WHOLE workload -- full calc grid and all events, nothing scaled from a sample:
      * all physical cores (the reference's TARGET='parallel'): prob3 grid under OpenMP + every
        container's lookup/reweight/histogram chain as one OpenMP loop over its events
        (oracle_container_chain: private histograms per thread, merged in thread order);
      * a thread-count scan of the same code (scaling of the baseline itself);
      * ONE thread, stage by stage as the reference runs it (TARGET='cpu': lookup arrays, weights
        array, one histogram pass for w and one for w^2).
    The all-core evaluation of the bench's last parameter point also gives `oracle_llh`, the LLH the
    device value is compared with."""
    import numpy as np

    from oracle import oracle as orc
    from oracle.pipeline_oracle import oracle_eval, oracle_eval_allcore

    orc.build()
    cores, logical = cores_logical or physical_cores()
    # columns placed for the all-core run: pages first touched by the thread that reads them (two
    # sockets: a column allocated by the main thread sits in ONE NUMA node and caps the event loop at
    # that node's bandwidth -- 64 threads were no faster than 32)
    orc.set_num_threads(cores)
    keep = []

    def placed(a):
        keep.append(orc.PartitionedCopy(a))
        return keep[-1].array

    events = []
    for ev in wl.events:
        d = dict(ev)
        for k in ("true_coszen", "nu_flux", "weighted_aeff", "initial_weights"):
            d[k] = placed(ev[k])
        d["sample"] = [placed(c) for c in ev["sample"]]
        events.append(d)
    ln_e = [placed(np.log(ev["true_energy"])) for ev in wl.events]

    def run(threads, reps):
        oracle_eval_allcore(wl, events, threads=threads, matrices=matrices, ln_energy=ln_e)      # warm-up
        ts, tg = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            res = oracle_eval_allcore(wl, events, threads=threads, matrices=matrices, ln_energy=ln_e)
            ts.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            oracle_eval_allcore(wl, containers=[], threads=threads, matrices=matrices)
            tg.append(time.perf_counter() - t0)
        return float(np.median(ts)), float(np.median(tg)), res

    t_all, g_all, res = run(cores, 7)
    lam = np.asarray(res["hist"]).reshape(len(wl.events), -1).sum(axis=0)
    oracle_llh = float(orc.metric("llh", data, lam)[1])
    scan = {}
    for th in sorted({1, 2, 4, 8, 16, 32, 64, cores}):
        if th <= cores and th != cores:
            t, g, _ = run(th, 3)
            scan[str(th)] = {"evals_per_s": 1.0 / t, "grid_s": g, "events_s": t - g}
    scan[str(cores)] = {"evals_per_s": 1.0 / t_all, "grid_s": g_all, "events_s": t_all - g_all}
    # the box is shared and the container may not own all of its cores: if a smaller team is faster,
    # that is the baseline (the better number for the CPU), with its thread count as `cores`
    best = max(scan, key=lambda k: scan[k]["evals_per_s"])
    if int(best) != cores:
        t_all, g_all, cores_used = 1.0 / scan[best]["evals_per_s"], scan[best]["grid_s"], int(best)
    else:
        cores_used = cores
    all_core_rate = scan[str(cores)]["evals_per_s"]
    collapsed = cores_used != cores and all_core_rate < 0.67 / t_all
    team_note = ""
    if cores_used != cores:
        team_note = ("; the %d-thread team gave %.1f evals/s%s -- the baseline is the best team of the scan, %d threads "
                     "pinned close = package(s) %s" % (cores, all_core_rate,
                                                        " (collapsed: the box's other socket is busy or its memory remote)"
                                                        if collapsed else "", cores_used, sockets_of_team(cores_used)))
    # one thread, stage by stage (what the reference's TARGET='cpu' runs)
    orc.set_num_threads(1)
    oracle_eval(wl, matrices, containers=[])
    t1 = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle_eval(wl, matrices)
        t1.append(time.perf_counter() - t0)
    t_one = float(np.median(t1))
    # config C4's 50 parameter points through the LLH gate (round 6): device LLH + summed device map of every point (made
    # by the caller, outside every timed region) against the all-core oracle on the same matrices
    c4_gate = None
    if c4:
        from oracle.referee import llh_referee

        rows = []
        for mi, llh_dev, lam_dev in zip(c4["matrices"], c4["llh"], c4["lam"]):
            r = oracle_eval_allcore(wl, events, threads=cores_used, matrices=mi, ln_energy=ln_e)
            lam_o = np.asarray(r["hist"]).reshape(len(wl.events), -1).sum(axis=0)
            rows.append(llh_referee(data, lam_dev, lam_o, float(llh_dev), float(orc.metric("llh", data, lam_o)[1])))
        rel = [r["fp64_abs_diff"] / abs(r["oracle_evaluation"]["llh_fp64"]) for r in rows]
        c4_gate = {"points": len(rows), "pure_1e-10_met": int(sum(r["pure_1e-10_relative_met"] for r in rows)),
                   "referee_met": int(sum(r["met"] for r in rows)),
                   "all_met": bool(all(r["pure_1e-10_relative_met"] or r["met"] for r in rows)),
                   "max_fp64_rel_diff": float(max(rel)),
                   "max_maps_rel_diff_extended": float(max(r["maps"]["rel_diff"] for r in rows)),
                   "max_device_over_eps_rms": float(max(r["device_evaluation"]["over_eps_rms"] for r in rows)),
                   "max_oracle_over_eps_rms": float(max(r["oracle_evaluation"]["over_eps_rms"] for r in rows)),
                   "gate_in_eps_rms": 8.0}
    return {
        "c4_llh_gate": c4_gate,
        "value": 1.0 / t_all,
        "unit": "evals/s",
        "cores": cores_used,
        "kind": "port",
        "cpu_model": cpu_model(),
        "physical_cores_available": cores,
        "logical_cpus_available": logical,
        "omp": {k: os.environ.get(k) for k in ("OMP_PLACES", "OMP_PROC_BIND")},
        "sample": "all %d events and the full %dx%dx2-node prob3 grid, nothing scaled: grid %.4f s + events "
                  "%.4f s per evaluation on %d OpenMP threads pinned to cores (events = one loop per container, "
                  "columns first-touched by the reading thread, per-thread private histograms merged in thread "
                  "order); best of the thread scan, medians%s"
                  % (wl.n_events, wl.grid.n_e, wl.grid.n_cz, g_all, t_all - g_all, cores_used, team_note),
        "sockets_used": sockets_of_team(cores_used),
        "all_core_team_collapsed": bool(collapsed),
        "thread_scan": scan,
        "single_thread": {
            "value": 1.0 / t_one, "unit": "evals/s", "cores": 1,
            "sample": "all %d events + full grid, stage by stage as the reference runs it (lookup arrays, "
                      "weights array, two histogram passes): %.3f s per evaluation; median of 3"
                      % (wl.n_events, t_one)},
        "oracle_llh": oracle_llh,
        "device_llh": device_llh,
        "llh_rel_diff": abs(device_llh - oracle_llh) / abs(oracle_llh) if oracle_llh else None,
        "llh_gate": llh_gate(data, lam, device_llh, oracle_llh, device_hist),
    }


def latest_profile(name):
    """newest committed profiles/r*/<name> (json), with the directory it came from"""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(os.path.dirname(f)))])
    if not files:
        return None, None
    with open(files[-1]) as fh:
        return json.load(fh), os.path.relpath(os.path.dirname(files[-1]), ROOT)


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from the newest COMMITTED rocprofv3 PMC passes
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; separate `rocprofv3 --pmc` runs of this same
    command, scripts/profile_round.sh).  Not measured in this run: `traffic_source` says where it
    comes from.  Only valid for the default workload."""
    if (args.coordinate_form or args.exact_association or args.wide_index or int(args.events) != 10000000
            or args.binning != "dragon"):
        return None, None
    d, src = latest_profile("traffic.json")
    if d is None:
        return None, None
    return d.get("hbm_bytes"), "%s/traffic.json (committed rocprofv3 --pmc passes, not this run)" % src


def trace_roofline(args, bytes_per_launch):
    """the dominant kernel's duration by the committed rocprofv3 kernel trace of this very command
    (profiles/<newest>/kernels_by_phase.json, scripts/profile_round.sh) and the roofline fraction that follows
    from it -- beside `roofline.frac`, which is this run's HIP-event time (events around one launch add ~3 us)"""
    if (args.coordinate_form or args.exact_association or args.wide_index or int(args.events) != 10000000
            or args.binning != "dragon" or args.gpus != 1):
        return None
    d, src = latest_profile("kernels_by_phase.json")
    if d is None or "hist_accumulate_kernel" not in d:
        return None
    us = d["hist_accumulate_kernel"]["timed_loop_mean_us"]
    ach = bytes_per_launch / (us * 1e-6) / 1e9
    return {"avg_launch_us": us, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "unit": "GB/s",
            "source": "%s/kernels_by_phase.json + kernel_stats.csv (committed rocprofv3 --kernel-trace --stats of "
                      "this command, not this run)" % src}


def time_fused(st, plist, lib, torch, k=30):
    """average duration of the fused accumulate kernel, HIP events on the launch stream"""
    import numpy as np

    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    torch.cuda.synchronize()
    for (a, b), p in zip(pairs, (plist * (k // len(plist) + 1))[:k]):
        a.record(); b.record()  # materialise the hipEvent handles
        lib.pisa_hip_profile_events(a.cuda_event, b.cuda_event)
        st.eval(p, "llh")
    lib.pisa_hip_profile_events(None, None)
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in pairs])) * 1e-3


def bytes_per_event(st, coordinate_form, compact, d_out):
    if coordinate_form:
        return 8 * (2 + 2 + 1 + 1 + d_out)   # 48 + 8 D = 72 B (D=3), 64 B (D=2)
    if not compact:
        return 4 + 4 + 16 + 8 + 8            # node, bin (int32) + flux(2) + aeff + w0 = 40 B
    return (2 + 2 + 16) if st.index16 else (4 + 4 + 16)


def hbm_leg(synthetic, lib, torch, n_events, n_e, n_cz, binning, steps, coordinate_form=False, compact=True):
    """one engine variant: whole-evaluation rate + roofline of its fused kernel"""
    wl = synthetic.Workload(n_events=int(n_events), grid=(n_e, n_cz), out_binning=binning, seed=0)
    st = synthetic.DeviceState(wl, indexed=not coordinate_form, compact=compact and not coordinate_form)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    plist = param_list(wl, 10 + steps)
    for p in plist[:10]:
        st.eval_host(p, "llh")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in plist[10:]:
        st.eval_host(p, "llh")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    t_k = time_fused(st, plist[:10], lib, torch)
    bpe = bytes_per_event(st, coordinate_form, compact and not coordinate_form, len(wl.ob["nbins"]))
    resident = bpe * st.n_local
    out = {
        "events": wl.n_events, "bytes_per_event": bpe, "resident_column_bytes": resident,
        "exceeds_l3": bool(resident > L3_BYTES), "evals_per_s": 1.0 / dt, "ms_per_step": dt * 1e3,
        "roofline": {"bound": "hbm", "achieved": resident / t_k / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": resident / t_k / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": t_k * 1e3,
                     "kernel": "hist_accumulate_kernel<%d, true>" % (1 if coordinate_form else
                                                                    ((7 if st.index16 else 5) if compact else 3))},
    }
    del st, wl
    torch.cuda.empty_cache()
    return out


def leg_update_flux(synthetic, torch, wl, st, steps):
    """every evaluation a Barr / spectral-index systematic moves as well (as in the 3-y fit):
    barr_simple kernel for the 12 containers + refresh of the folded flux columns + evaluation"""
    import numpy as np

    from pisa_amd import kernels as K

    cols = []
    for ev in wl.events:
        cols.append((K.to_device(ev["true_energy"]), K.to_device(ev["true_coszen"]), K.to_device(ev["nu_flux"]),
                     K.to_device(ev["nu_flux"] * 0.7), ev["nubar"]))
    plist = param_list(wl, 5 + steps)
    rs = np.random.RandomState(5)
    # as the stages do it: one Barr launch for all containers into arrays the stage owns, one fold launch
    outs = [torch.empty((e.numel(), 2), dtype=torch.float64, device=e.device) for e, *_ in cols]
    sets = K.barr_sets([(e, cz, nom, nom_bar, nubar, out) for (e, cz, nom, nom_bar, nubar), out in zip(cols, outs)])
    items = list(enumerate(outs))

    def one(p):
        didx, ratio = 0.1 * (rs.rand() - 0.5), 1.0 + 0.05 * (rs.rand() - 0.5)
        K.barr_simple_multi(sets, ratio, 1.0, didx, 0.0, 0.0)
        st.update_flux_many(items)
        return st.eval_host(p, "llh")

    for p in plist[:5]:
        one(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in plist[5:]:
        one(p)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # ONE pass: the engine keeps the nominal fluxes and the events' parameter-free Barr factors in its
    # resident order and writes the folded column directly (HotPathEngine.update_flux_barr)
    st.enable_barr([(e, cz, nom, nom_bar) for e, cz, nom, nom_bar, _ in cols])

    def one_pass(p):
        didx, ratio = 0.1 * (rs.rand() - 0.5), 1.0 + 0.05 * (rs.rand() - 0.5)
        st.update_flux_barr(ratio, 1.0, didx, 0.0, 0.0)
        return st.eval_host(p, "llh")

    for p in plist[:5]:
        one_pass(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in plist[5:]:
        one_pass(p)
    torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t0) / steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        st.update_flux_barr(1.01, 1.0, 0.02, 0.0, 0.0)
    e1.record()
    torch.cuda.synchronize()
    t_kernel = e0.elapsed_time(e1) / 20 * 1e-3
    bytes_per_event = 2 * 16 + 5 * 8 + 8 + 16     # nominal pairs, factors, static weight; folded pair written
    # restore the nominal columns for whatever runs after this leg
    for i, ev in enumerate(wl.events):
        st.update_flux(i, K.to_device(ev["nu_flux"]))
    st._barr = None
    torch.cuda.empty_cache()
    return {"evals_per_s": 1.0 / dt1, "ms_per_step": dt1 * 1e3,
            "two_pass": {"evals_per_s": 1.0 / dt, "ms_per_step": dt * 1e3},
            "refresh_kernel": {"kernel": "barr_fold_multi_kernel", "avg_launch_ms": t_kernel * 1e3,
                               "bytes_per_event": bytes_per_event,
                               "achieved_GBs": bytes_per_event * st.n_local / t_kernel / 1e9,
                               "frac_of_hbm_peak": bytes_per_event * st.n_local / t_kernel / 1e9 / HBM_PEAK_GBS},
            "what": "flux.barr_simple (nue/numu ratio + spectral index moved) for all events with the flux per event + "
                    "the headline evaluation, every step.  Headline of the leg: ONE pass (resident-order nominal "
                    "fluxes and parameter-free Barr factors -> folded (w0*aeff*flux) column, same bits); two_pass: "
                    "the stage's Barr launch in container order + gather / fold launch"}


def leg_node_flux(synthetic, torch, args, n_e, n_cz, steps):
    """the shape of the IceCube 3-year cfgs: the flux lives on the oscillation grid (flux stages with
    osc.prob3's calc_mode).  Every step: barr_simple on the grid nodes of the 12 containers, new node
    tables, oscillation, flux x probability per node (pisa_hip_flux_prob_tables), fused
    lookup+histogram with the events' static factor, LLH."""
    import numpy as np

    from pisa_amd import kernels as K

    wl = synthetic.Workload(n_events=int(args.events), grid=(n_e, n_cz), out_binning=args.binning, seed=0)
    g = wl.grid
    ee, cc = np.meshgrid(g.energy, g.coszen, indexing="ij")
    f_mu = 1e4 * ee ** -2.7 * (1 + 0.5 * cc ** 2)
    nodes = np.stack([f_mu * (0.5 - 0.2 * cc), f_mu], axis=-1).reshape(-1, 2)
    for ev in wl.events:
        ev["nu_flux_nodes"] = nodes
    st = synthetic.DeviceState(wl, compact=True, node_flux=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    e_d, cz_d = K.to_device(ee.ravel()), K.to_device(cc.ravel())
    nom, nom_bar = K.to_device(nodes), K.to_device(nodes * 0.7)
    nubars = [ev["nubar"] for ev in wl.events]
    plist = param_list(wl, 5 + 2 * steps)
    rs = np.random.RandomState(5)
    barr_sets = K.barr_sets([(e_d, cz_d, nom, nom_bar, nubar, st._node_flux_t[i]) for i, nubar in enumerate(nubars)])

    def one(p, flux_moves):
        if flux_moves:
            didx, ratio = 0.1 * (rs.rand() - 0.5), 1.0 + 0.05 * (rs.rand() - 0.5)
            K.barr_simple_multi(barr_sets, ratio, 1.0, didx, 0.0, 0.0)   # as the stage does: one launch
        return st.eval_host(p, "llh")

    out = {}
    for key, moves, pts in (("osc_only", False, plist[5:5 + steps]), ("flux_moves", True, plist[5 + steps:])):
        for p in plist[:5]:
            one(p, moves)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in pts:
            one(p, moves)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[key] = {"evals_per_s": 1.0 / dt, "ms_per_step": dt * 1e3}
    out["what"] = ("flux on the %d x %d oscillation grid (IceCube 3-y cfg shape): per-node flux x probability tables "
                   "+ 20 B/event fused kernel; flux_moves = flux.barr_simple on the nodes of all containers (one "
                   "launch, pisa_hip_barr_simple_multi) every step as well" % (n_e, n_cz))
    del st, wl
    torch.cuda.empty_cache()
    return out


def _pipeline_cfg(n_events, kde=False, kde_tol=None):
    from collections import OrderedDict

    from pisa_amd.core.config_parser import parse_pipeline_config

    cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
    cfg[("data", "synthetic_events")]["params"].params.n_events.value = n_events
    if not kde:
        return cfg
    out = OrderedDict()
    for k, v in cfg.items():
        if k == ("utils", "hist"):
            out[("utils", "kde")] = OrderedDict(calc_mode="events", apply_mode=v["apply_mode"],
                                                **({} if kde_tol is None else {"tol": kde_tol}))
        else:
            out[k] = v
    out["pipeline"]["output_key"] = "weights"
    return out


def leg_pipeline_boundary(torch, n_events, steps):
    """the same kind of evaluation through the reference's API: cfg text -> Pipeline.get_outputs()
    -> Map.metric_total (what `Analysis._minimizer_callable` does per point)"""
    import numpy as np

    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline(_pipeline_cfg(n_events))
    data = sum(pipe.get_outputs()).fluctuate("poisson", random_state=0)
    rs = np.random.RandomState(2024)
    pts = [(31.0 + 28.0 * rs.rand(), 1e-3 + 6e-3 * rs.rand()) for _ in range(10 + steps)]

    def one(pt):
        pipe.params.theta23.value = pt[0] * ureg.degree
        pipe.params.deltam31.value = pt[1] * ureg.eV ** 2
        return data.metric_total(expected_values=sum(pipe.get_outputs()), metric="llh")

    for pt in pts[:10]:
        one(pt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for pt in pts[10:]:
        llh = one(pt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the same with a flux.barr_simple parameter (flux per event in this cfg) moving every step as well:
    # replayed as the engine's one-pass refresh of the folded flux columns
    rsf = np.random.RandomState(3)

    def one_flux(pt):
        pipe.params.delta_index.value = 0.1 * (rsf.rand() - 0.5) * ureg.dimensionless
        return one(pt)

    for pt in pts[:5]:
        one_flux(pt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_flux = max(20, steps // 4)
    for pt in pts[10:10 + n_flux]:
        one_flux(pt)
    torch.cuda.synchronize()
    dt_flux = (time.perf_counter() - t0) / n_flux
    flux_replayed = bool(pipe._plan is not None and getattr(pipe._plan, "_barr_ready", False))
    pipe.params.delta_index.value = 0.0 * ureg.dimensionless
    pipe.fast_path = False
    pipe._plan = None
    for pt in pts[:3]:
        one(pt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # (>= 300 steps: at 0.24 ms per step a 20-step sample of this host-bound leg read 2 159 .. 4 300 on the round-4 boxes)
    n_slow = 300
    for i in range(n_slow):
        one(pts[10 + i % steps])
    torch.cuda.synchronize()
    dt_slow = (time.perf_counter() - t0) / n_slow
    cm = pipe["prob3"].calc_mode
    out_shape = tuple(pipe.output_binning.shape)
    del pipe
    torch.cuda.empty_cache()
    # the engine alone on the same workload (same events, calc grid and binning): what the
    # boundary costs on top
    from pisa_amd import synthetic

    wl = synthetic.Workload(n_events=int(n_events), grid=tuple(cm.shape), out_binning="example3d", seed=0)
    st = synthetic.DeviceState(wl, compact=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    plist = param_list(wl, 10 + steps)
    for p in plist[:10]:
        st.eval_host(p, "llh")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in plist[10:]:
        st.eval_host(p, "llh")
    torch.cuda.synchronize()
    dt_eng = (time.perf_counter() - t0) / steps
    del st, wl
    torch.cuda.empty_cache()
    return {"evals_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "last_llh": llh,
            "engine_same_workload_evals_per_s": 1.0 / dt_eng, "boundary_over_engine": dt / dt_eng,
            "stage_protocol_every_step_evals_per_s": 1.0 / dt_slow,
            "flux_per_event_moves_evals_per_s": 1.0 / dt_flux, "flux_per_event_moves_replayed": flux_replayed,
            "workload": "settings/pipeline/example_hip.cfg (cfg text): %d events, prob3 on the %s calc grid, "
                        "aeff, hist into %s with sumw2; theta23/dm31 set through pipeline.params every step, "
                        "Pipeline.get_outputs() + Map.metric_total('llh') read back every step"
                        % (int(n_events) // 12 * 12, "x".join(str(n) for n in cm.shape),
                           "x".join(str(n) for n in out_shape))}


def leg_icecube3y(torch, n_events, steps):
    """the published 3-year analysis through its own boundary: the unmodified cfgs
    IceCube_3y_neutrinos.cfg (csv_loader -> honda_ip -> barr_simple -> prob3 -> aeff -> hist ->
    hypersurfaces) + IceCube_3y_muons.cfg in a DistributionMaker, a synthetic stand-in for the MC file,
    `get_outputs(return_sum=True)` + `Map.metric_total('mod_chi2')` against the released data histogram,
    parameters set through `params[...]` every step"""
    import subprocess
    import tempfile

    import numpy as np

    tmp = tempfile.mkdtemp(prefix="pisa_hip_3y_")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synthetic_3y_mc.py"), tmp,
                           str(int(n_events)), "3"], stdout=subprocess.DEVNULL)
    old = os.environ.get("PISA_RESOURCES")
    os.environ["PISA_RESOURCES"] = tmp
    try:
        from pisa_amd.core.distribution_maker import DistributionMaker
        from pisa_amd.core.pipeline import Pipeline

        template = DistributionMaker(["settings/pipeline/IceCube_3y_neutrinos.cfg",
                                      "settings/pipeline/IceCube_3y_muons.cfg"])
        data = Pipeline("settings/pipeline/IceCube_3y_data.cfg").get_outputs()[0]
        free = [p.name for p in template.params.free]
        rs = np.random.RandomState(1)

        def step(which):
            for name in which:
                p = template.params[name]
                lo, hi = ((p.range[0].magnitude, p.range[1].magnitude) if p.range is not None
                          else (p.value.magnitude * 0.9, p.value.magnitude * 1.1))
                p.value = (p.nominal_value.magnitude + 0.05 * (hi - lo) * (rs.rand() - 0.5)) * p.value.units
            total = template.get_outputs(return_sum=True)[0]
            return data.metric_total(expected_values=total, metric="mod_chi2")

        out = {"events": int(n_events), "free_parameters": free}
        osc = [f for f in free if f in ("theta23", "deltam31", "theta13")]
        for key, which in (("all_free", free), ("osc_only", osc)):
            for _ in range(3):
                step(which)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                val = step(which)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[key] = {"evals_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "last_mod_chi2": val}
        nu = template.pipelines[0]
        out["plan"] = bool(nu._plan is not None)
        out["node_flux"] = bool(nu["hist"]._engine.node_flux)
        out["workload"] = ("DistributionMaker(IceCube_3y_neutrinos.cfg + IceCube_3y_muons.cfg, unmodified) on %d "
                           "synthetic MC events; every step moves the listed parameters (oscillation, Barr flux, "
                           "aeff norms, hypersurface detector systematics, muon scale), sums the 13 maps and "
                           "evaluates mod_chi2 against the released data histogram" % int(n_events))
        return out
    finally:
        if old is None:
            os.environ.pop("PISA_RESOURCES", None)
        else:
            os.environ["PISA_RESOURCES"] = old


def leg_events(synthetic, torch, n_events, steps, nsi, rank=0, world=1, share=None, sync=None, reduce_max=None,
               on_device=False, decay=False):
    """configs C2 / C5 (per-GPU share): prob3 EVENT BY EVENT (layers rebuilt per event in-kernel from the
    PREM table in LDS) + fused reweight + 10x10 histogram + LLH.  N > 1 (C5: 1e8 events on 8 GPUs): every
    rank holds `n_events` events of its own (seed = rank), the limbs are all-reduced over the ranks
    (`share` = an engine whose communicator is reused), every rank evaluates the same LLH."""
    import numpy as np

    wl = synthetic.Workload(n_events=int(n_events), grid=(10, 10), out_binning="example2d", seed=rank,
                            on_device=on_device)
    st = synthetic.DeviceState(wl, osc_mode="events", compact=True)
    device_bytes = torch.cuda.memory_allocated()
    if world > 1 or share is not None:
        st.world_size = share.world_size
        st.group, st._rccl = share.group, share._rccl
    mat_pot = None
    if nsi:
        from pisa_amd.stages.osc.nsi_params import StdNSIParams

        n = StdNSIParams()
        n.eps_emu, n.eps_etau, n.eps_mutau = ((0.07, np.deg2rad(340)), (0.06, np.deg2rad(35)),
                                              (0.003, np.deg2rad(175)))  # numba_osc_tests.py:129-136
        mat_pot = np.diag([1.0, 0, 0]).astype(complex) + n.eps_matrix
    dec = 1e-4 if decay else None     # decay_alpha3 in eV^2: the decay instantiation of the event kernel
    st.make_pseudo_data(wl.osc_params(mat_pot=mat_pot, decay_alpha3=dec), seed=0)
    plist = param_list(wl, 3 + steps, mat_pot=mat_pot, decay_alpha3=dec)
    # eval_host: the LLH arrives in pinned host memory (what a fit loop reads), as in the headline loop
    sync = sync or torch.cuda.synchronize
    reduce_max = reduce_max or (lambda x: x)
    for p in plist[:3]:
        st.eval_host(p, "llh")
    sync()
    t0 = time.perf_counter()
    for p in plist[3:]:
        llh = st.eval_host(p, "llh")
    sync()
    dt = reduce_max(time.perf_counter() - t0) / steps
    st.check_status()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for p in plist[3:]:
        st.compute_probs(p)
    e1.record()
    torch.cuda.synchronize()
    t_osc = e0.elapsed_time(e1) / steps * 1e-3
    out = {"events": wl.n_events * world, "events_per_gpu": wl.n_events, "evals_per_s": 1.0 / dt,
           "ms_per_step": dt * 1e3, "event_evals_per_s": wl.n_events * world / dt,
           "prob3_events_kernel_ms": t_osc * 1e3, "last_llh": llh,
           "device_bytes_allocated": device_bytes,
           "algorithmic_resident_bytes_per_event": 16 + 16 + 24,   # (E, coszen) + (P_e, P_mu) pair + 24 B index / folded flux
           "events_generated": "in HBM (torch generator)" if on_device else "on the host (numpy RandomState)",
           "workload": "%d events%s, prob3 event by event (PREM-12%s) + fused reweight + 10x10 hist + LLH"
                       % (wl.n_events * world, (" on %d GPUs (%d each, limbs all-reduced)" % (world, wl.n_events))
                          if world > 1 else "", (", std NSI" if nsi else "") + (", neutrino decay" if decay else ""))}
    if world > 1:
        import torch.distributed as dist

        v = torch.tensor([llh], dtype=torch.float64, device="cuda")
        allv = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(allv, v)
        out["same_llh_bits_on_all_ranks"] = bool(all(bool((a.view(torch.int64) == v.view(torch.int64)).all()) for a in allv))
    st._rccl = None
    # executed fp64 flops per event of prob3_events_kernel from the committed SQ_INSTS_VALU_*_F64
    # counter passes (scripts/profile_round.sh); not measured in this run
    cal, src = latest_profile("events_flops.json")
    key = "decay" if decay else ("nsi" if nsi else "std")
    if cal is not None and key in cal:
        fpe = cal[key]["flop_per_event"]
        ach = fpe * wl.n_events / t_osc / 1e12
        out["roofline"] = {"bound": "fp64 valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / FP64_VALU_PEAK_TFLOPS, "flop_per_event": fpe,
                           "flop_source": "%s/events_flops.json (executed fp64 FMA x2 + ADD + MUL + TRANS lane "
                                          "operations of prob3_events_kernel from committed rocprofv3 "
                                          "SQ_INSTS_VALU_*_F64 passes, not this run)" % src}
    del st, wl
    torch.cuda.empty_cache()
    return out


def leg_multi_point(torch, st, wl, sync, reduce_max):
    """K independent parameter points per sweep of the events (`HotPathEngine.eval_many`: batched prob3,
    one fused launch with K accumulator sets, one tail workgroup per point) on the headline workload:
    evaluations per second for K = 3, 5, 9, beside the point-by-point `value`.  Per point the LLH is
    the same bits as `eval_host` (checked here on the last batch)."""
    out = {}
    serial_last = None
    plist = param_list(wl, 45)
    for k in (3, 5, 9):
        batches = [plist[i:i + k] for i in range(0, len(plist) - k + 1, k)]
        for b in batches[:3]:
            st.eval_many(b, "llh")
        sync()
        t0 = time.perf_counter()
        n = 0
        while True:
            for b in batches:
                vals = st.eval_many(b, "llh")
                n += k
            sync()
            dt = reduce_max(time.perf_counter() - t0)
            if dt > 0.15:
                break
        serial_last = [st.eval_host(p, "llh") for p in batches[-1]]
        out["K%d" % k] = {"evals_per_s": n / dt, "us_per_point": 1e6 * dt / n, "points_per_sweep": k,
                          "same_bits_as_point_by_point": bool(vals == serial_last)}
    st.check_status()
    out["what"] = ("K independent parameter points per sweep of the events (finite-difference stencils, scans): "
                   "prob3 for K points in one pair of launches, ONE pass over the 20 B/event columns with K "
                   "accumulator sets, one tail workgroup per point, K LLH values read back; per point bit-identical "
                   "to the point-by-point evaluation")
    return out


def leg_point_parallel(make_state, st, wl, rank, world, state_kw, sync, reduce_max, hybrid=False):
    """Hybrid point x event parallelism (`engine.PointGroups`): the W ranks as G groups x R shards, the K points of an
    `eval_many` call dealt to the groups, every point computed inside one group (the sample replicated on a group's one
    rank, or sharded over its R ranks with the limb all-reduce inside the group), one all-gather of K doubles.  Measured
    for G = W (sample replicated on every GPU) and, from four ranks on, G = W / 2 (two shards per group): 9 points per
    group and call, evaluations per second of the whole job; the values are compared, bit for bit, with the event-sharded
    engine's `eval_many` of the same points."""
    from pisa_amd.engine import PointGroups

    out = {}
    nominal = wl.osc_params()
    # (the hybrid topology creates sub-groups and per-group RCCL communicators: measured on request -- `--point-hybrid` --
    # and in the tests; the default multi-GPU line keeps to G = W, which needs nothing beyond the world's all-gather)
    for n_groups in ([world, world // 2] if (world >= 4 and hybrid) else [world]):
        pg = PointGroups(rank, world, n_groups)
        stp = make_state(wl, points=pg, **state_kw)
        stp.make_pseudo_data(nominal, seed=0)
        k = 9 * n_groups
        plist = param_list(wl, 3 * k)
        batches = [plist[i:i + k] for i in range(0, len(plist), k)]
        for b in batches:
            vals = stp.eval_many(b, "llh")
        sync()
        t0 = time.perf_counter()
        n = 0
        while True:
            for b in batches:
                vals = stp.eval_many(b, "llh")
                n += k
            sync()
            dt = reduce_max(time.perf_counter() - t0)
            if dt > 0.15:
                break
        want = st.eval_many(batches[-1], "llh")      # the event-sharded engine (one group of W shards), same points
        out[pg.topology] = {"groups": n_groups, "shards_per_group": pg.n_shards, "points_per_call": k,
                            "evals_per_s": n / dt, "us_per_point": 1e6 * dt / n,
                            "same_bits_as_event_sharded": bool(vals == want)}
        stp.check_status()
        if hasattr(stp, "close"):
            stp.close()
        del stp
    out["what"] = ("K = 9 G points per eval_many call dealt to G groups of R ranks (world rank = group R + shard); a group "
                   "holds the whole sample (R = 1: replicated) or shards it (int64 limb all-reduce inside the group); one "
                   "all-gather of K doubles; per point the single-GPU bits")
    return out


def leg_fit_engine(torch, st, wl, sync, reduce_max):
    """The C4 fit loop on the engine itself (no Pipeline / Param layer in between): scipy L-BFGS-B
    (eps 1e-4, ftol 2e-5, gtol 1e-5: the reference's l-bfgs-b settings) over (theta23, deltam31) rescaled
    to [0, 1], -LLH against the headline pseudo-data, several starting points until >= 50 evaluations.
    `point_by_point`: scipy takes its forward differences through `eval_host`; `stencil_in_one_sweep`:
    value and gradient from one `eval_many` of the same n + 1 points (Analysis._forward_stencil).  Same
    trajectory, point for point.  `four_free` repeats it with theta13 and deltacp free as well (stencils
    of five points)."""
    import numpy as np
    from scipy import optimize

    from pisa_amd.analysis.analysis import Analysis

    opts = dict(ftol=2e-5, gtol=1e-5, eps=1e-4, maxiter=200)
    names = ("theta23_deg", "dm31", "theta13_deg", "deltacp_deg")
    lo_all, hi_all = np.array([31.0, 1e-3, 7.0, 0.0]), np.array([59.0, 7e-3, 10.0, 360.0])
    starts_all = [(0.40, 0.24, 0.5, 0.1), (0.25, 0.20, 0.4, 0.3), (0.71, 0.30, 0.6, 0.2), (0.46, 0.33, 0.45, 0.5),
                  (0.64, 0.22, 0.55, 0.4), (0.32, 0.28, 0.5, 0.6)]

    def run(n_free):
        lo, hi = lo_all[:n_free], hi_all[:n_free]
        bounds = [(0.0, 1.0)] * n_free

        def point(x):
            v = lo + (hi - lo) * np.clip(x, 0.0, 1.0)
            return wl.osc_params(**{k: float(a) for k, a in zip(names, v)})

        trace = {}

        def serial(x):
            f = -st.eval_host(point(x), "llh")
            trace["pts"].append((tuple(x), f))
            return f

        def swept(x):
            pts, dx = Analysis._forward_stencil(x, opts["eps"], np.zeros(n_free), np.ones(n_free))
            f = [-v for v in st.eval_many([point(q) for q in pts], "llh")]
            trace["pts"] += [(tuple(q), v) for q, v in zip(pts, f)]
            return f[0], (np.array(f[1:]) - f[0]) / dx

        out, traces = {}, {}
        for key, fun, jac in (("point_by_point", serial, None), ("stencil_in_one_sweep", swept, True)):
            dt = None
            for rep in range(4):   # the first repetition warms up; the best of the other three counts
                trace["pts"] = []
                evals, fits = 0, []
                sync()
                t0 = time.perf_counter()
                for x0 in starts_all:
                    res = optimize.minimize(fun, np.array(x0[:n_free]), jac=jac, bounds=bounds, method="L-BFGS-B",
                                            options=opts)
                    evals = len(trace["pts"])
                    fits.append((float(res.fun), [float(v) for v in lo + (hi - lo) * res.x]))
                    if evals >= 50 and len(fits) >= 2:
                        break
                sync()
                t = reduce_max(time.perf_counter() - t0)
                if rep > 0:
                    dt = t if dt is None else min(dt, t)
            traces[key] = list(trace["pts"])
            out[key] = {"wall_s": dt, "llh_evaluations": evals, "fits": len(fits), "evals_per_s": evals / dt,
                        "best_fit": dict(zip(("neg_llh",) + names[:n_free], [fits[0][0]] + fits[0][1]))}
        out["same_history"] = bool(traces["point_by_point"] == traces["stencil_in_one_sweep"])
        out["speedup"] = out["point_by_point"]["wall_s"] / out["stencil_in_one_sweep"]["wall_s"]
        return out

    out = run(2)
    out["four_free"] = run(4)
    out["workload"] = ("scipy L-BFGS-B (eps 1e-4) on HotPathEngine directly: free theta23, deltam31 (four_free: + "
                       "theta13, deltacp); the headline workload (%d events, 200x100 grid, 8x8x2 bins), -llh against "
                       "its pseudo-data" % wl.n_events)
    st.check_status()
    return out


def leg_fit_c4(torch, n_events, dist_on, sync, reduce_max):
    """BASELINE config C4: the fit loop.  Analysis.fit_hypo (scipy L-BFGS-B, the reference's
    l-bfgs-b_ftol2e-5_gtol1e-5_eps1e-4_maxiter200 settings) over 2 free parameters (theta23, deltam31) on
    the cfg-text pipeline with `n_events` events (sharded over the ranks when N > 1: every rank runs the
    same minimiser on its shard, the limbs are all-reduced per evaluation), pseudo-data = Poisson-fluctuated
    template at an injected truth; repeated from several starting points until >= 50 LLH evaluations have
    been made.  Timed twice: the minimiser taking its finite differences point by point (the reference's
    flow), and with every iterate's stencil in one sweep (`batched_gradient`) -- the same fit, point for
    point (`same_history`)."""
    import numpy as np

    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    dm = DistributionMaker(_pipeline_cfg(n_events))
    for name in dm.params.free.names:
        if name not in ("theta23", "deltam31"):
            dm.params.fix(name)
    dm.params.theta23.value = 47.5 * ureg.degree
    dm.params.deltam31.value = 2.55e-3 * ureg.eV ** 2
    data = dm.get_outputs(return_sum=True).fluctuate("poisson", random_state=0)
    starts = [(42.3, 2.457e-3), (38.0, 2.2e-3), (51.0, 2.8e-3), (44.0, 3.0e-3), (49.0, 2.3e-3), (40.0, 2.7e-3)]
    ana = Analysis()
    out = {}
    hist = {}
    for key, batched in (("point_by_point", False), ("stencil_in_one_sweep", True)):
        dt = None
        for rep in range(4):     # the first round warms every code path up; the best of the other three counts
            evals, fits, hist[key] = 0, [], []
            sync()
            t0 = time.perf_counter()
            for t23, dm31 in starts:
                dm.params.theta23.value = t23 * ureg.degree
                dm.params.deltam31.value = dm31 * ureg.eV ** 2
                res = ana.fit_hypo(data, dm, "llh", reset_free=False, batched_gradient=batched)
                evals += res.num_distributions_generated
                fits.append((res.metric_val, res.params.theta23.value.m_as("deg"),
                             res.params.deltam31.value.m_as("eV**2")))
                hist[key] += res.fit_history
                if evals >= 50 and len(fits) >= 2:
                    break
            sync()
            t = reduce_max(time.perf_counter() - t0)
            if rep > 0:
                dt = t if dt is None else min(dt, t)
        out[key] = {"wall_s": dt, "llh_evaluations": evals, "fits": len(fits), "evals_per_s": evals / dt,
                    "best_fit": {"llh": fits[0][0], "theta23_deg": fits[0][1], "deltam31_eV2": fits[0][2]}}
    out["same_history"] = bool(hist["point_by_point"] == hist["stencil_in_one_sweep"])
    out["speedup"] = out["point_by_point"]["wall_s"] / out["stencil_in_one_sweep"]["wall_s"]
    if dist_on:
        import torch.distributed as dist

        # every rank ran the same minimiser on all-reduced limbs: the LLH bits must agree
        v = torch.tensor([out["stencil_in_one_sweep"]["best_fit"]["llh"]], dtype=torch.float64, device="cuda")
        allv = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(allv, v)
        out["same_llh_bits_on_all_ranks"] = bool(all(bool((a.view(torch.int64) == v.view(torch.int64)).all()) for a in allv))
    out["workload"] = ("Analysis.fit_hypo, L-BFGS-B (eps 1e-4), free: theta23, deltam31; %d events through "
                       "settings/pipeline/example_hip.cfg (cfg text), llh against Poisson pseudo-data at (47.5 deg, "
                       "2.55e-3 eV^2); fits from successive starting points until >= 50 LLH evaluations"
                       % (int(n_events) // 12 * 12))
    del dm
    torch.cuda.empty_cache()
    return out


def leg_osc_example(torch, steps):
    """BASELINE config C1: the unmodified `settings/pipeline/osc_example.cfg` (toy generator on the calc
    grid -> flux.barr_simple -> osc.prob3, 200 x 200 (E, coszen) PREM-12 grid, 12 output maps = the README's
    oscillograms), `Pipeline.get_outputs()` with theta23 changed every step, maps read on the host.  The one
    number the reference publishes for this path is osc.prob3's compute on this grid: mean 0.887 s per call,
    ~11 us per node, unstated CPU, one thread."""
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    pipe = Pipeline("settings/pipeline/osc_example.cfg")
    pipe.get_outputs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = 0.0
    for i in range(steps):
        pipe.params.theta23.value = (40.0 + 0.05 * i) * ureg.degree
        maps = pipe.get_outputs()
        total += float(maps[1].hist[0, 0])     # the maps are on the host
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    osc = pipe["prob3"]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50):
        pipe.params.theta23.value = (41.0 + 0.05 * i) * ureg.degree
        osc.compute()
    e1.record()
    torch.cuda.synchronize()
    t_osc = e0.elapsed_time(e1) / 50 * 1e-3
    n_nodes = 2 * int(osc.calc_mode.size)
    return {"evals_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "maps": len(maps), "map_shape": list(maps[0].hist.shape),
            "prob3_compute_ms": t_osc * 1e3, "prob3_nodes_per_s": n_nodes / t_osc, "prob3_us_per_node": t_osc / n_nodes * 1e6,
            "reference_published": {"prob3_compute_s": 0.887, "us_per_node": 11.0, "hardware": "unstated CPU, 1 thread",
                                    "source": "pisa_examples/IceCube_3y_oscillations_example.ipynb:987 (BASELINE.md)"},
            "workload": "settings/pipeline/osc_example.cfg (unmodified text): prob3 on the 200x200 calc grid for nu and "
                        "nubar (%d nodes), 12 oscillogram maps brought to the host every step" % n_nodes}


def leg_kde(torch, n_events, steps):
    """config C3: the event pipeline with the KDE stage ON (reference defaults: adaptive Silverman
    bandwidths, oversample 10, coszen reflection, pid stacking)"""
    from pisa_amd.core.pipeline import Pipeline
    from pisa_amd.core.units import ureg

    from pisa_amd.stages.utils.kde import KDE_FAST_TOL, KDE_STAGE_TOL

    def run(tol, steps):
        pipe = Pipeline(_pipeline_cfg(n_events, kde=True, kde_tol=tol))
        for i in range(3):   # (the library's worker threads size their workspaces on their first jobs)
            pipe.params.theta23.value = (37.0 + i) * ureg.degree
            pipe.get_outputs()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.params.theta23.value = (40.0 + 0.5 * i) * ureg.degree
            maps = pipe.get_outputs()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, pipe, maps

    # the stage's default cut-off (1e-14: all pairs at rounding level) beside the explicit fast one the leg reports
    dt_default, pipe, maps = run(None, max(4, steps // 2))
    total_default = float(sum(m.hist.sum() for m in maps))
    del pipe, maps
    dt, pipe, maps = run(KDE_FAST_TOL, steps)
    st = pipe["kde"].stats
    work = st["pairs_pilot"] + st["pairs_eval"]
    out = {"events": int(n_events) // 12 * 12, "evals_per_s": 1.0 / dt, "ms_per_step": dt * 1e3,
           "kernel_evaluations_per_step": work, "all_pairs_would_be": st["all_pairs"],
           "total_of_maps": float(sum(m.hist.sum() for m in maps)),
           "tol": KDE_FAST_TOL,
           "default_tol": {"tol": KDE_STAGE_TOL, "ms_per_step": dt_default * 1e3, "total_of_maps": total_default},
           "workload": "settings/pipeline/example_hip.cfg with utils.kde (tol = 1e-12 set in the cfg: every bin of these maps "
                       "within 6.2e-12 of the all-pairs evaluation, scripts/dev/kde_tol_budget.py; the stage's default 1e-14 "
                       "timed beside it) in place of utils.hist: 12 containers x 2 "
                       "pid channels = 24 adaptive 2-D KDEs per evaluation, 150 x 100 evaluation points each; "
                       "theta23 changed every step; KDE core parity unpinned (un-vendored `kde` package)"}
    # executed fp64 flops of ALL kde_* kernels of one evaluation from the committed SQ_INSTS_VALU_*_F64
    # counter pass over scripts/dev/c3_probe.py (scripts/profile_round.sh); not measured in this run.
    # (The map evaluation no longer spends one exponential per kernel value -- Gaussian recurrence along
    # lattice lines -- so a flop count per counted kernel value would mean nothing.)
    cal, src = latest_profile("kde_flops.json")
    if cal is not None and cal.get("events") == out["events"]:
        flop = cal["fp64_flop_per_evaluation"]
        out["roofline"] = {"bound": "fp64 valu (the matrix cores' fp64 peak is the same 78.6 TFLOP/s)", "achieved": flop / dt / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": flop / dt / 1e12 / FP64_VALU_PEAK_TFLOPS,
                           "fp64_flop_per_evaluation": flop,
                           "flop_source": "%s/kde_flops.json (executed fp64 FMA x2 + ADD + MUL + TRANS lane operations "
                                          "and the matrix cores' fp64 operations of all kde_* kernels of one evaluation, "
                                          "committed rocprofv3 pass, not this run)" % src,
                           "note": "whole evaluation wall time (24 estimators on 8 threads / streams of the library: "
                                   "sorts, pilot through local expansions, lattice evaluation, host glue)"}
    return out


def main(argv=None, hooks=None):
    """`hooks` (tests only, tests/test_distributed_cpu.py): run this very control flow -- N > 1 process
    group, barriers, max over ranks, the legs that run on several ranks, the JSON line -- on CPU ranks over
    gloo, with `hooks["device_state"]` in place of the HIP-backed engine.  Nothing in the product or in the
    driver's invocation passes hooks."""
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.cpu_baseline_worker, args.binning)
    if hooks is None:
        hooks = _hooks_from_env()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become one (before anything touches the GPU)
        return launch_ranks(args, argv, standin=hooks is not None)
    import numpy as np
    import torch

    # tests only: hooks = {"share_device": True} runs the REAL engine on N ranks that all use HIP device 0 and exchange over
    # gloo (tests/test_gpu_distributed.py: the N > 1 control flow and the sharded kernels on hardware with one GPU)
    share = bool(hooks and hooks.get("share_device"))
    cuda = hooks is None or share
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (--nproc-per-node %d wanted)"
                         % (args.gpus, world, args.gpus))
    if cuda:
        have = torch.cuda.device_count()
        device_index = 0 if share else local_rank
        if have <= device_index:
            raise SystemExit("bench.py: rank %d wants HIP device %d, %d device(s) visible" % (rank, device_index, have))
        torch.cuda.set_device(device_index)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist

        if cuda and not share:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from pisa_amd import _lib, synthetic

    make_state = synthetic.DeviceState if cuda else hooks["device_state"]
    dev_sync = torch.cuda.synchronize if cuda else (lambda: None)
    if not cuda or share:
        args.no_kernel_timing = args.no_batch_probe = args.no_drop_probe = args.no_cpu_baseline = True

    n_e, n_cz = (int(v) for v in args.grid.split("x"))
    compact = not (args.exact_association or args.coordinate_form)
    index16 = compact and not args.wide_index
    order = True if args.event_order == "auto" else args.event_order

    def barrier():
        if dist_on:
            import torch.distributed as dist

            dist.barrier()
        dev_sync()

    def max_over_ranks(x):
        if not dist_on:
            return x
        import torch.distributed as dist

        t = torch.tensor([x], dtype=torch.float64, device="cuda" if (cuda and not share) else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_loop(st, plist):
        """W warm-up steps, then blocks of EXACTLY K timed steps, each block bracketed by barrier +
        synchronize, MAX over ranks per block.  One block is what the contract asks for; with a small
        K a block lasts a millisecond or two, so blocks are repeated (same K points) until
        `--min-timed-s` of timed work has been seen and the mean block time is reported
        (`timed_blocks` in the line)."""
        llh = 0.0
        for p in plist[: args.warmup]:
            llh = st.eval_host(p, "llh")
        st.check_status()
        total, blocks = 0.0, 0
        while True:
            barrier()
            t0 = time.perf_counter()
            for p in plist[args.warmup:]:
                llh = st.eval_host(p, "llh")
            barrier()
            total += max_over_ranks(time.perf_counter() - t0)   # the same number on every rank
            blocks += 1
            if total >= args.min_timed_s or blocks >= 10000:
                break
        st.check_status()
        timed_loop.blocks = blocks
        return total / blocks, llh

    # ---- headline: ONE sample of --events events, sharded over the ranks (strong scaling)
    # (the runtime's first-use costs -- code objects, staging buffers -- are paid on a background thread while the host
    #  generates the sample, as a user's program would call pisa_amd.warm_up() beside reading its event files)
    warm_ms = None
    if cuda and not args.no_warm_up:
        import pisa_amd

        pisa_amd.warm_up(background=True)
    wl = synthetic.Workload(n_events=int(args.events), grid=(n_e, n_cz), out_binning=args.binning, seed=0)
    if cuda and not args.no_warm_up:
        try:
            warm_ms = pisa_amd.warm_up_wait()
        except Exception as exc:     # the warm-up is a convenience: whatever stopped it stops the real set-up below, with its own message
            print("bench.py: pisa_amd.warm_up() failed (%s: %s)" % (type(exc).__name__, exc), file=sys.stderr)
    t_setup0 = time.perf_counter()
    st = make_state(wl, rank=rank, world_size=world, indexed=not args.coordinate_form,
                    sort_events=order, compact=compact, index16=index16, **({"time_setup": True} if cuda else {}))
    dev_sync()
    setup_first = {"wall_ms": 1e3 * (time.perf_counter() - t_setup0), "phases_ms": getattr(st, "setup_ms", None),
                   "after_warm_up": warm_ms is not None, "warm_up_ms": warm_ms}
    if args.force_dist and world == 1:
        st.world_size = 2   # one rank, but through the collective
    nominal = wl.osc_params()
    st.make_pseudo_data(nominal, seed=0)
    plist = param_list(wl, args.warmup + args.steps)
    mats_last = dict(wl.last_matrices)     # of plist[-1], the point whose LLH the line reports
    dt, llh = timed_loop(st, plist)
    headline_blocks = timed_loop.blocks
    # set-up a second time in the warm process (allocator grown, code objects loaded): host columns -> engine ready
    setup_warm = None
    if cuda and not dist_on:
        t_setup0 = time.perf_counter()
        st_again = make_state(wl, rank=rank, world_size=world, indexed=not args.coordinate_form,
                              sort_events=order, compact=compact, index16=index16, time_setup=True)
        dev_sync()
        setup_warm = {"wall_ms": 1e3 * (time.perf_counter() - t_setup0), "phases_ms": st_again.setup_ms}
        del st_again
        torch.cuda.empty_cache()
    # the device's maps of the last timed point (12 x n_bins doubles): what the LLH referee compares with the oracle's
    dev_hist_last = st.maps()[0] if (cuda and not args.no_cpu_baseline and world == 1) else None
    lib = _lib.lib() if cuda else None
    d_out = len(wl.ob["nbins"])
    bpe = bytes_per_event(st, args.coordinate_form, compact, d_out)
    fused_avg_s = float("nan") if args.no_kernel_timing else time_fused(st, plist[args.warmup:], lib, torch,
                                                                         k=max(10, min(args.steps, 50)))
    achieved = bpe * st.n_local / fused_avg_s / 1e9

    # per-phase device times (extra information, not part of the contract)
    def time_phase(fn, n=30):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))

    def tail():
        st._maps_valid = False
        st._tail("llh", st.metric_out)

    t_prob3 = time_phase(lambda: st.compute_probs(nominal)) if cuda else None
    t_tail = time_phase(tail) if cuda else None
    t_allreduce = time_phase(st.allreduce) if (dist_on and cuda) else None
    n_comm = None
    if dist_on and st._rccl:
        n_comm = st._rccl.count()
        if n_comm != world:
            raise SystemExit("bench.py: the RCCL communicator spans %d rank(s), %d launched" % (n_comm, world))
    # the LLH of the last timed point as every rank holds it: integer limbs summed over the ranks, the tail
    # replicated -- the bits must be the same everywhere (north star: bit-reproducible across GPU counts)
    llh_bits = None
    if dist_on:
        import torch.distributed as dist

        mine = torch.tensor([int(np.float64(llh).view(np.int64))], dtype=torch.int64,
                            device="cuda" if (cuda and not share) else "cpu")
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        llh_bits = ["%016x" % (int(t.item()) & 0xFFFFFFFFFFFFFFFF) for t in every]
        if len(set(llh_bits)) != 1:
            raise SystemExit("bench.py: the ranks hold different LLH bits: %s" % llh_bits)

    # ---- weak scaling beside it: every rank a sample of its own, limbs all-reduced, N samples per step
    weak = None
    if dist_on:
        wl_w = synthetic.Workload(n_events=int(args.events), grid=(n_e, n_cz), out_binning=args.binning, seed=rank)
        st_w = make_state(wl_w, rank=0, world_size=1, indexed=not args.coordinate_form,
                          sort_events=order, compact=compact, index16=index16)
        st_w.world_size = max(world, 2) if args.force_dist else world
        st_w.group, st_w._rccl = st.group, st._rccl          # same communicator
        st_w.make_pseudo_data(nominal, seed=0)
        dt_w, _ = timed_loop(st_w, plist)
        weak = {"value": args.steps / dt_w * world, "ms_per_step": 1e3 * dt_w / args.steps,
                "events_per_gpu": wl_w.n_events, "samples_per_step": world}
        st_w._rccl = None
        del st_w, wl_w

    # for information: the same evaluations with the events that can never land in a bin
    # (static reco coordinates outside the output binning) not kept resident
    dropped = None
    if not dist_on and not args.coordinate_form and not args.no_drop_probe:
        st2 = synthetic.DeviceState(wl, sort_events=order, drop_unbinned=True, compact=compact, index16=index16)
        st2.set_data(st.data.cpu().numpy())
        for p in plist[: args.warmup]:
            st2.eval_host(p, "llh")
        torch.cuda.synchronize()
        t0d = time.perf_counter()
        for p in plist[args.warmup:]:
            llh2 = st2.eval_host(p, "llh")
        dtd = time.perf_counter() - t0d
        dropped = {"evals_per_s": args.steps / dtd, "events_resident": st2.n_local, "same_llh": bool(llh2 == llh)}
        del st2

    # ---- legs (N = 1 only)
    legs = {}
    want = [] if args.legs == "none" else (list(ALL_LEGS) if args.legs == "all" else
                                           [x.strip() for x in args.legs.split(",") if x.strip()])
    if dist_on:
        want = [x for x in want if x in DIST_LEGS]
    if hooks is not None:
        want = [x for x in want if x in hooks.get("legs", ("multi_point",))]
    leg_steps = max(20, min(args.steps, 200))
    for name in want:
        t0 = time.perf_counter()
        try:
            if name == "multi_point":
                legs[name] = leg_multi_point(torch, st, wl, barrier, max_over_ranks) \
                    if (compact and index16 and not args.coordinate_form) else None
            elif name == "point_parallel":
                legs[name] = leg_point_parallel(make_state, st, wl, rank, world,
                                                dict(indexed=not args.coordinate_form, sort_events=order, compact=compact,
                                                     index16=index16), barrier, max_over_ranks,
                                                hybrid=bool(args.point_hybrid or hooks is not None)) \
                    if (dist_on and world > 1 and compact and index16 and not args.coordinate_form) else None
            elif name == "fit_c4_engine":
                legs[name] = leg_fit_engine(torch, st, wl, barrier, max_over_ranks)
            elif name == "fit_c4":
                legs[name] = leg_fit_c4(torch, args.events, dist_on, barrier, max_over_ranks)
            elif name == "osc_example_c1":
                legs[name] = leg_osc_example(torch, leg_steps)
            elif name == "l3_exceeding":
                legs[name] = hbm_leg(synthetic, lib, torch, 4 * args.events, n_e, n_cz, args.binning, leg_steps // 2)
            elif name == "exact_association":
                legs[name] = hbm_leg(synthetic, lib, torch, args.events, n_e, n_cz, args.binning, leg_steps,
                                     compact=False)
            elif name == "coordinate_form":
                legs[name] = hbm_leg(synthetic, lib, torch, args.events, n_e, n_cz, args.binning, leg_steps,
                                     coordinate_form=True)
            elif name == "fine_binning":
                # 40 x 40 x 3 = 4 800 output bins: beyond the LDS accumulators, LDS-window path
                legs[name] = hbm_leg(synthetic, lib, torch, args.events, n_e, n_cz, "fine3d", leg_steps)
            elif name == "update_flux":
                legs[name] = leg_update_flux(synthetic, torch, wl, st, leg_steps) if compact else None
            elif name == "node_flux":
                legs[name] = leg_node_flux(synthetic, torch, args, n_e, n_cz, leg_steps)
            elif name == "pipeline_boundary":
                legs[name] = leg_pipeline_boundary(torch, args.events, leg_steps)
            elif name == "icecube3y_boundary":
                legs[name] = leg_icecube3y(torch, 2e5, leg_steps)
            elif name == "events_c2":
                legs[name] = leg_events(synthetic, torch, 1e6, 20, nsi=False)
            elif name == "events_c5":
                legs[name] = leg_events(synthetic, torch, 1.25e7, 6, nsi=True, rank=rank, world=world,
                                        share=st if dist_on else None, sync=barrier, reduce_max=max_over_ranks)
            elif name == "events_c2_decay":
                legs[name] = leg_events(synthetic, torch, 1e6, 20, nsi=False, decay=True)
            elif name == "events_c5_full":
                # C5 at the size BASELINE.json states, on ONE device: 1e8 std-NSI events, generated in HBM
                legs[name] = leg_events(synthetic, torch, 1e8, 3, nsi=True, on_device=True)
            elif name == "kde_c3":
                legs[name] = leg_kde(torch, args.events, 16)
            else:
                raise ValueError("unknown leg %r" % name)
        except Exception as exc:  # a leg must not take the headline down with it
            if dist_on:
                raise   # the other ranks are inside a collective: better the launcher ends them all
            legs[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if isinstance(legs.get(name), dict):
            legs[name]["leg_wall_s"] = time.perf_counter() - t0

    # stream-overlapped evaluation of independent points (e.g. finite-difference gradient
    # stencils): prob3 of point k+1 runs beside the fused kernel of point k.  Run AFTER the legs: its
    # high-priority stream (torch keeps pooled streams alive) takes one of the device's four hardware queues for
    # the rest of the process, and the KDE leg's eight streams then share three (13.3 -> 15.8 ms per evaluation)
    bsz = 10
    pipelined = None
    if args.batch_probe and not args.no_batch_probe and not dist_on:
        try:
            st.eval_batch(plist[:bsz]).cpu()
            torch.cuda.synchronize()
            t0b = time.perf_counter()
            nb = 0
            for i in range(args.warmup, args.warmup + args.steps - bsz + 1, bsz):
                st.eval_batch(plist[i:i + bsz]).cpu()
                nb += bsz
            torch.cuda.synchronize()
            pipelined = nb / (time.perf_counter() - t0b) if nb else None
        except Exception:   # informational only
            pipelined = None

    if rank == 0:
        evals_per_s = args.steps / dt
        headline_weak = dist_on and args.weak_scaling
        traffic, traffic_src = pmc_traffic(args) if not dist_on else (None, None)
        out = {
            "metric": "pipeline evals/sec (osc+reweight+hist+LLH) on 1e7 MC events",
            "value": weak["value"] if headline_weak else evals_per_s,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": weak["ms_per_step"] if headline_weak else 1e3 * dt / args.steps,
            "higher_is_better": True,
            # ONE sample of --events events whatever N (its events sharded over the ranks): total work
            # fixed, so the series N = 1, 2, 4, 8 is a strong-scaling series and the N = 1 point carries
            # the same label as the others
            "scaling": "weak" if headline_weak else "strong",
            "timed_blocks": headline_blocks,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (toy_event_generator-style E/coszen, builder-defined reco/flux/aeff; see pisa_amd/synthetic.py)",
            "config": {
                "workload": "%d events in 12 containers%s, prob3 on %dx%d (E,coszen) PREM-12 calc grid (nu+nubar), fused "
                            "lookup+reweight+%s hist with sumw2, Poisson LLH; theta23/dm31 changed every eval, LLH "
                            "read back every eval; event columns %d B/event (%s)"
                            % (wl.n_events, (", sharded over %d GPUs" % world) if world > 1 else "", n_e, n_cz,
                               "x".join(str(b) for b in wl.ob["nbins"]), bpe,
                               ("static factors initial_weights*weighted_aeff folded into the flux pair"
                                + (", 16-bit node and bin indices" if st.index16 else "")) if compact
                               else ("coordinates binned on the fly, SURVEY 8(d) form" if args.coordinate_form
                                     else "reference operation order")),
                "events": wl.n_events,
                "calc_grid": [n_e, n_cz],
                "out_bins": wl.ob["nbins"],
                "parallelism": "events sharded over %d GPU(s), int64 limb all-reduce over RCCL, prob3 grid and "
                               "metric replicated" % world,
            },
            "event_evals_per_s": evals_per_s * wl.n_events,
            "strong_value": evals_per_s,
            "weak_value": weak["value"] if weak else None,
            "weak": weak,
            "allreduce_ms": t_allreduce,
            "nccl_comm_count": n_comm,
            "last_llh": llh,
            "llh_bits_per_rank": llh_bits,
            "llh_bits_identical": None if llh_bits is None else len(set(llh_bits)) == 1,
            # host columns -> first evaluation ready (HotPathEngine's constructor; device-synchronised phases: what the launch
            # stream waits for the PCIe copies -- they run beside the previous container's phases --, digitisation,
            # resident order, packing, oscillation plan): the first construction of the process and one in the warm process
            "setup_ms": None if setup_warm is None else setup_warm["wall_ms"],
            "setup": {"first_in_process": setup_first, "warm_process": setup_warm,
                      "evaluations_worth": None if setup_warm is None else setup_warm["wall_ms"] / (1e3 * dt / args.steps),
                      "host_bytes_uploaded": 80 * wl.n_events,
                      "note": "80 B/event of host columns cross PCIe once (E, ln E, coszen, flux pair, aeff, w0, three reco "
                              "columns): at ~30 GB/s from pageable memory that alone is ~2.7 ms per 1e6 events"},
            # test-only stand-ins (PISA_BENCH_HOOKS / hooks=): a line produced with them says so
            "hooks_used": hooks is not None,
            "share_device": share,
            "pipelined_evals_per_s": pipelined,
            "batched_evals_per_s3": (legs.get("multi_point") or {}).get("K3", {}).get("evals_per_s"),
            "batched_evals_per_s5": (legs.get("multi_point") or {}).get("K5", {}).get("evals_per_s"),
            "batched_evals_per_s9": (legs.get("multi_point") or {}).get("K9", {}).get("evals_per_s"),
            # hybrid point x event parallelism (engine.PointGroups): the stencil's points dealt to groups of ranks
            "topology": "%dx1" % world if (legs.get("point_parallel") or {}).get("%dx1" % world) else "1x%d" % world,
            "point_parallel_evals_per_s": ((legs.get("point_parallel") or {}).get("%dx1" % world) or {}).get("evals_per_s"),
            "unbinned_events_dropped": dropped,
            "phase_ms": {"prob3_grid": t_prob3, "fused_reweight_hist": 1e3 * fused_avg_s, "finalize_metric": t_tail,
                         "allreduce": t_allreduce, "events_this_rank": st.n_local},
            "roofline": {
                "bound": "hbm",
                "kernel": "hist_accumulate_kernel<%d, true>" % (1 if args.coordinate_form else
                                                               ((7 if st.index16 else 5) if compact else 3)),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "bytes_per_event": bpe,
                "events_per_launch": st.n_local,
                "avg_launch_ms": 1e3 * fused_avg_s,
                "resident_column_bytes": bpe * st.n_local,
                "fits_l3": bool(bpe * st.n_local <= L3_BYTES),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "timing": "HIP events recorded on the launch stream around the kernel, inside the library",
                "by_kernel_trace": trace_roofline(args, bpe * st.n_local) if not dist_on else None,
                # the same kernel on 4 x the events (800 MB of columns: beyond the 256 MiB Infinity Cache, every byte from HBM):
                # the HBM-true fraction beside `frac` (whose 200 MB of columns stay inside the Infinity Cache between launches)
                "frac_beyond_l3": ((legs.get("l3_exceeding") or {}).get("roofline") or {}).get("frac"),
                "beyond_l3": {k: ((legs.get("l3_exceeding") or {}).get("roofline") or {}).get(k) for k in ("achieved", "avg_launch_ms")}
                if legs.get("l3_exceeding") else None,
            },
            "legs": legs,
        }
        if not args.no_cpu_baseline and world == 1:
            c4 = None
            if cuda and args.c4_gate_points > 0:
                c4 = {"llh": [], "lam": [], "matrices": []}
                for p, mats in param_points(wl, args.c4_gate_points):
                    c4["matrices"].append(mats)
                    c4["llh"].append(st.eval_host(p, "llh"))
                    c4["lam"].append(np.asarray(st.maps()[0]).sum(axis=0))
            cb = cpu_baseline_subprocess(args, n_e, n_cz, st.data.cpu().numpy(), mats_last, llh, dev_hist_last, c4)
            out["cpu_baseline"] = cb
            out["c4_llh_gate"] = cb.pop("c4_llh_gate", None)
            # the bench's last headline point against the oracle on identical inputs (north star: <= 1e-10)
            out["oracle_llh"], out["llh_rel_diff"] = cb["oracle_llh"], cb["llh_rel_diff"]
            out["llh_gate"] = cb["llh_gate"]
        detail_path = None
        if args.detail_out != "-":
            detail_path = os.path.abspath(args.detail_out or os.path.join(ROOT, "bench_detail.json"))
            try:
                with open(detail_path, "w") as fh:
                    json.dump(out, fh)
                    fh.write("\n")
            except OSError as exc:
                print("bench.py: detail file not written (%s)" % exc, file=sys.stderr)
                detail_path = None
        # the full result also on stderr (one line, marked), the compact contract line LAST on stdout
        print("bench_detail " + json.dumps(out), file=sys.stderr, flush=True)
        print(compact_line(out, detail_path), flush=True)
        if hooks is not None and "result" in hooks:
            hooks["result"](out)
    if dist_on:
        import torch.distributed as dist

        st.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
