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
N > 1: every rank holds 1e7 events of its own and the int64 histogram limbs are all-reduced
(weak scaling; `value` counts 1e7-event evaluation units, N per step); `--strong-scaling` shards
one 1e7-event sample instead.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
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
                    help="bin event coordinates on the fly (72 B/event) instead of the pre-digitised "
                         "index columns (40 B/event)")
    ap.add_argument("--cpu-sample-events", type=float, default=2.4e6)
    ap.add_argument("--exact-association", action="store_true",
                    help="stream the 40 B/event columns (initial_weights, weighted_aeff, nu_flux kept separate, "
                         "the reference's operation order) instead of the compact form in which the "
                         "static per-event factors are folded into the flux pair once")
    ap.add_argument("--wide-index", action="store_true",
                    help="compact form with 32-bit node and bin indices (24 B/event) instead of the "
                         "16-bit ones (20 B/event; same weights, same arithmetic, identical results)")
    ap.add_argument("--no-batch-probe", action="store_true",
                    help="skip the informational stream-overlapped batch evaluation (keeps a rocprofv3 "
                         "kernel average free of launches that share the chip with another stream)")
    ap.add_argument("--no-drop-probe", action="store_true",
                    help="skip the informational second engine without the events outside the binning")
    ap.add_argument("--strong-scaling", action="store_true",
                    help="N > 1: shard ONE sample of --events events over the ranks (fixed total work).  The "
                         "default for N > 1 is weak scaling: every rank holds --events events of its own "
                         "(seed = rank), the histograms of all ranks are all-reduced, and `value` counts "
                         "evaluations of --events-sized units (N per step) -- one MI355X already evaluates "
                         "1e7 events in the time of a few kernel launches, so only the per-GPU-constant regime "
                         "has anything to scale (DESIGN.md section 6)")
    ap.add_argument("--weak-scaling", action="store_true", help="(default for N > 1; kept for compatibility)")
    ap.add_argument("--force-dist", action="store_true",
                    help="test aid: take the N > 1 code path (RCCL process group, limb all-reduce, barriers, "
                         "max over ranks) with the ranks that are there, e.g. one rank under "
                         "torch.distributed.run on a single-GPU box")
    ap.add_argument("--event-order", default="auto", choices=["auto", "node", "bin", "part"],
                    help="resident event order: sorted by calc-grid node, or by (output bin, node)")
    return ap.parse_args()


def param_list(wl, n):
    """fixed seeded scan of (theta23, dm31) over the ranges of SURVEY 8d (C4)"""
    import numpy as np

    rs = np.random.RandomState(2024)
    out = []
    for _ in range(n):
        out.append(wl.osc_params(theta23_deg=31.0 + 28.0 * rs.rand(), dm31=1e-3 + 6e-3 * rs.rand()))
    return out


def cpu_baseline(wl, sample_events):
    """The oracle (C restatement of the reference algorithms) timed on this
    box's host cores on a bounded sample: the full calc grid + a subsample of
    the events, scaled to the full event count."""
    import numpy as np

    from oracle import oracle as orc
    from oracle.pipeline_oracle import oracle_eval

    orc.build()
    # one thread per physical core of one socket at most: the event loops are
    # memory bound and the histogram loop is sequential, more threads only add
    # OpenMP overhead (measured: 256 SMT threads ran 10x slower than 8)
    cores = min(len(os.sched_getaffinity(0)), 64)
    orc.set_num_threads(cores)
    n_per = max(1, int(sample_events) // len(wl.events))
    sub = []
    for ev in wl.events:
        d = dict(ev)
        for k in ("true_energy", "true_coszen", "nu_flux", "weighted_aeff", "initial_weights"):
            d[k] = ev[k][:n_per]
        d["sample"] = [s[:n_per] for s in ev["sample"]]
        sub.append(d)
    wl.osc_params()
    oracle_eval(wl, containers=[])  # warm-up (loads the library, touches the grid)
    # median of five repetitions each (the host is shared and the sample is short)
    t_all, t_grid = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        ref = oracle_eval(wl, containers=sub)
        t_all.append(time.perf_counter() - t0)
        # split: grid part is independent of the number of events
        t0 = time.perf_counter()
        oracle_eval(wl, containers=[])
        t_grid.append(time.perf_counter() - t0)
    t_all, t_grid = float(np.median(t_all)), float(np.median(t_grid))
    t_events = max(t_all - t_grid, 1e-9)
    n_sub = n_per * len(sub)
    t_full = t_grid + t_events * (wl.n_events / n_sub)
    orc.metric("llh", ref["hist"].sum(axis=0) + 1, ref["hist"].sum(axis=0) + 1)
    return {
        "value": 1.0 / t_full,
        "unit": "evals/s",
        "cores": cores,
        "kind": "port",
        "sample": "full %dx%dx2-node prob3 grid (%.3f s) + %d of %d events through "
                  "lookup/reweight/hist (%.3f s), event part scaled to all events; "
                  "OpenMP over %d threads (histogram loop sequential); medians of 5 repetitions"
                  % (wl.grid.n_e, wl.grid.n_cz, t_grid, n_sub, wl.n_events, t_events, cores),
    }


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3
    PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; collected by
    separate `rocprofv3 --pmc` runs of this same command, see profiles/*/traffic.json).
    Only valid for the default workload."""
    if args.coordinate_form or args.exact_association or args.wide_index or int(args.events) != 10000000 or args.binning != "dragon":
        return None
    import glob

    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(os.path.dirname(f)))])
    if not files:
        return None
    with open(files[-1]) as fh:
        return json.load(fh).get("hbm_bytes")


def main():
    args = parse()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from pisa_amd import _lib, synthetic

    n_e, n_cz = (int(v) for v in args.grid.split("x"))
    weak = dist_on and not args.strong_scaling
    compact = not (args.exact_association or args.coordinate_form)
    index16 = compact and not args.wide_index
    wl = synthetic.Workload(n_events=int(args.events), grid=(n_e, n_cz), out_binning=args.binning,
                            seed=rank if weak else 0)
    st = synthetic.DeviceState(wl, rank=0 if weak else rank, world_size=1 if weak else world,
                               indexed=not args.coordinate_form,
                               sort_events=True if args.event_order == "auto" else args.event_order,
                               compact=compact, index16=index16)
    if weak:
        # whole local sample per rank; the limb all-reduce still spans all ranks (>= 2 so that
        # --force-dist on one rank goes through the collective as well)
        st.world_size = max(world, 2) if args.force_dist else world
    nominal = wl.osc_params()
    st.make_pseudo_data(nominal, seed=0)
    plist = param_list(wl, args.warmup + args.steps)

    def barrier():
        if dist_on:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    llh = 0.0
    for p in plist[: args.warmup]:
        llh = st.eval_host(p, "llh")
    st.check_status()

    # ---- timed region: exactly K evaluations, LLH read back every time
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    fused_ms = []
    barrier()
    t0 = time.perf_counter()
    for p in plist[args.warmup:]:
        llh = st.eval_host(p, "llh")
    barrier()
    dt = time.perf_counter() - t0
    st.check_status()

    # ---- dominant kernel, measured live with HIP events on the launch stream
    lib = _lib.lib()
    k_meas = max(10, min(args.steps, 50))
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(k_meas)]
    torch.cuda.synchronize()
    d_out = len(wl.ob["nbins"])
    if args.coordinate_form:
        bytes_per_event = 8 * (2 + 2 + 1 + 1 + d_out)  # 72 B (D=3), 64 B (D=2)
    else:
        bytes_per_event = 4 + 4 + 16 + 8 + 8  # node, bin (int32) + flux(2) + aeff + w0 = 40 B
        if compact:
            bytes_per_event = 4 + 4 + 16  # node, bin + (w0*aeff*f_e, w0*aeff*f_mu) = 24 B
            if st.index16:
                bytes_per_event = 2 + 2 + 16  # both indices in 16 bits = 20 B
    if args.no_kernel_timing:
        fused_avg_s = float("nan")
    else:
        for (a, b), p in zip(pairs, plist[args.warmup:] + plist):
            a.record(); b.record()  # materialise the hipEvent handles
            lib.pisa_hip_profile_events(a.cuda_event, b.cuda_event)
            st.eval(p, "llh")
        lib.pisa_hip_profile_events(None, None)
        torch.cuda.synchronize()
        fused_ms = [a.elapsed_time(b) for a, b in pairs]
        fused_avg_s = float(np.mean(fused_ms)) * 1e-3
    achieved = bytes_per_event * st.n_local / fused_avg_s / 1e9

    # per-phase device times (extra information, not part of the contract)
    def time_phase(fn, n=30):
        # median of individually timed calls: a single slow call (the chip dropping its clocks
        # after the host-side pause between phases) does not distort it
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

    t_prob3 = time_phase(lambda: st.compute_probs(nominal))
    def tail():
        st._maps_valid = False
        st._tail("llh", st.metric_out)

    t_tail = time_phase(tail)

    # stream-overlapped evaluation of independent points (e.g. finite-difference
    # gradient stencils): prob3 of point k+1 runs beside the fused kernel of point k
    bsz = 10
    pipelined = None
    if not args.no_batch_probe:
        st.eval_batch(plist[:bsz]).cpu()
        barrier()
        t0b = time.perf_counter()
        nb = 0
        for i in range(args.warmup, args.warmup + args.steps - bsz + 1, bsz):
            st.eval_batch(plist[i:i + bsz]).cpu()
            nb += bsz
        barrier()
        dtb = time.perf_counter() - t0b
        pipelined = nb / dtb if nb else None

    # for information: the same evaluations with the events that can never land in a bin
    # (static reco coordinates outside the output binning) not kept resident
    dropped = None
    if world == 1 and not args.coordinate_form and not args.no_drop_probe:
        st2 = synthetic.DeviceState(wl, sort_events=True if args.event_order == "auto" else args.event_order,
                                    drop_unbinned=True, compact=compact, index16=index16)
        st2.set_data(st.data.cpu().numpy())
        for p in plist[: args.warmup]:
            st2.eval_host(p, "llh")
        torch.cuda.synchronize()
        t0d = time.perf_counter()
        for p in plist[args.warmup:]:
            llh2 = st2.eval_host(p, "llh")
        dtd = time.perf_counter() - t0d
        dropped = {"evals_per_s": args.steps / dtd, "events_resident": st2.n_local,
                   "same_llh": bool(llh2 == llh)}
        del st2

    # max over ranks
    if dist_on:
        import torch.distributed as dist

        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        evals_per_s = args.steps / dt
        units = world if weak else 1  # weak scaling: one step evaluates `world` samples of --events events
        out = {
            "metric": "pipeline evals/sec (osc+reweight+hist+LLH) on 1e7 MC events",
            "value": evals_per_s * units,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.strong_scaling else "weak",  # identical workloads at N = 1
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (toy_event_generator-style E/coszen, builder-defined reco/flux/aeff; see pisa_amd/synthetic.py)",
            "config": {
                "workload": ("weak scaling, PER GPU: " if weak else "") + "%d events in 12 containers, prob3 on %dx%d (E,coszen) PREM-12 calc grid "
                            "(nu+nubar), fused lookup+reweight+%s hist with sumw2, Poisson LLH; "
                            "theta23/dm31 changed every eval, LLH read back every eval; %s"
                            % (wl.n_events, n_e, n_cz, "x".join(str(b) for b in wl.ob["nbins"]),
                               "event columns %d B/event (%s)" % (
                                   bytes_per_event,
                                   ("static factors initial_weights*weighted_aeff folded into the flux pair"
                                    + (", 16-bit node and bin indices" if st.index16 else ""))
                                   if compact else "reference operation order")),
                "events": wl.n_events,
                "calc_grid": [n_e, n_cz],
                "out_bins": wl.ob["nbins"],
                "parallelism": ("%d GPU(s) x %d events each, int64 limb all-reduce" % (world, wl.n_events)) if weak
                               else "events sharded over %d GPU(s), int64 limb all-reduce" % world,
            },
            "event_evals_per_s": evals_per_s * wl.n_events * units,
            "last_llh": llh,
            "pipelined_evals_per_s": pipelined,
            "unbinned_events_dropped": dropped,
            "phase_ms": {"prob3_grid": t_prob3, "fused_reweight_hist": 1e3 * fused_avg_s,
                         "finalize_metric": t_tail},
            "roofline": {
                "bound": "hbm",
                "kernel": "hist_accumulate_kernel<%d, true>" % (1 if args.coordinate_form else
                                                               ((7 if st.index16 else 5) if compact else 3)),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "bytes_per_event": bytes_per_event,
                "events_per_launch": st.n_local,
                "avg_launch_ms": 1e3 * fused_avg_s,
                "traffic": pmc_traffic(args) if world == 1 else None,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample_events)
        print(json.dumps(out))
    if dist_on:
        import torch.distributed as dist

        st.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
