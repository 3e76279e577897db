"""Sharded evaluation with N ranks on ONE GPU over gloo against the single-rank engine, random workloads (development
tool, GPU box): every trial draws a sample (down to fewer events than ranks per container), a calc grid, an output binning
(LDS accumulators or windows), an engine layout, a world size of 2-6, a metric and a few parameter points; every rank builds
`synthetic.DeviceState(wl, rank, world)` on HIP device 0, evaluates the points (single point through the one-call
evaluator or the three calls, several points through one sweep) with the limb all-reduce going through `torch.distributed`
-- and rank 0 compares every value, bit for bit, with the engine that holds the whole sample.
usage: fuzz_ranks.py [trials] [seed]"""
import os
import pickle
import socket
import sys
import tempfile

import numpy as np

sys.path.insert(0, ".")


def worker(rank, world, port, spec_path, out_path):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pisa_amd import synthetic

    spec = pickle.load(open(spec_path, "rb"))
    synthetic.BINNINGS["fuzz"] = spec["binning"]
    wl = synthetic.Workload(n_events=spec["n_events"], grid=spec["grid"], out_binning="fuzz", seed=spec["seed"])
    st = synthetic.DeviceState(wl, rank=rank, world_size=world, **spec["kw"])
    st.one_call = spec["one_call"]
    st.set_data(spec["data"])
    pts = [wl.osc_params(**p) for p in spec["points"]]
    vals = [float(st.eval_host(p, spec["kind"])) for p in pts]
    many = [float(v) for v in st.eval_many(pts, spec["kind"])] if spec["sweep"] else None
    st.check_status()
    h, s2 = (x.cpu().numpy() for x in st.finalize())
    if rank == 0:
        pickle.dump(dict(vals=vals, many=many, hist=h, sumw2=s2), open(out_path, "wb"))
    st.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    from pisa_amd import synthetic

    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    tmp = tempfile.mkdtemp()
    for trial in range(trials):
        dims = int(rs.randint(1, 4))
        target = int(10 ** rs.uniform(0, 3.6))
        per = max(1, int(round(target ** (1.0 / dims))))
        nb = [max(1, int(per * rs.uniform(0.5, 1.6))) for _ in range(dims)]
        if dims == 3:
            nb[2] = int(rs.randint(1, 4))
        binning = dict(mins=[np.log(5.0), -1.0, -1000.0][:dims], maxs=[np.log(100.0), 1.0, 1000.0][:dims], nbins=nb,
                       log=[True, False, False][:dims])
        world = int(rs.randint(2, 7))
        n_events = int(12 * max(1, int(10 ** rs.uniform(0, 4.0))))
        form = ["reference", "compact", "compact16"][rs.randint(3)]
        kw = dict(sort_events=[True, False, "node", "bin", "part"][rs.randint(5)])
        if form != "reference":
            kw.update(compact=True, index16=form == "compact16")
        points = [dict(theta23_deg=float(rs.uniform(31, 59)), dm31=float(rs.uniform(1e-3, 7e-3))) for _ in range(int(rs.randint(1, 5)))]
        kind = ["llh", "chi2", "mod_chi2", "poisson_llh"][rs.randint(4)]
        synthetic.BINNINGS["fuzz"] = binning
        seed = int(rs.randint(1 << 30))
        wl = synthetic.Workload(n_events=n_events, grid=(int(rs.randint(3, 50)), int(rs.randint(3, 40))), out_binning="fuzz", seed=seed)
        grid = (wl.grid.n_e, wl.grid.n_cz) if hasattr(wl.grid, "n_e") else None
        one = synthetic.DeviceState(wl, **kw)
        one.make_pseudo_data(wl.osc_params(), seed=1)
        data = one.data.cpu().numpy()
        pts = [wl.osc_params(**p) for p in points]
        want = [float(one.eval_host(p, kind)) for p in pts]
        wh, ws2 = (x.cpu().numpy() for x in one.finalize())
        sweep = form == "compact16" and len(points) > 1
        want_many = [float(v) for v in one.eval_many(pts, kind)] if sweep else None
        spec = dict(binning=binning, n_events=n_events, grid=(one.grid.n_e, one.grid.n_cz), seed=seed, kw=kw, data=data, points=points, kind=kind,
                    one_call=bool(rs.rand() < 0.5), sweep=sweep)
        one.close()
        spec_path, out_path = os.path.join(tmp, "spec.pkl"), os.path.join(tmp, "out.pkl")
        pickle.dump(spec, open(spec_path, "wb"))
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        tag = "trial %d: %d ranks, %d events, bins %s, %s %s, %s, %d point(s)%s" % (trial, world, wl.n_events, nb, form, kw, kind, len(points),
                                                                                 " + sweep" if sweep else "")
        try:
            mp.spawn(worker, args=(world, port, spec_path, out_path), nprocs=world, join=True)
            got = pickle.load(open(out_path, "rb"))
            eq = lambda a, b: a == b or (a != a and b != b)  # noqa: E731
            ok = all(eq(a, b) for a, b in zip(got["vals"], want)) and np.array_equal(got["hist"], wh) and np.array_equal(got["sumw2"], ws2)
            if sweep:
                ok = ok and all(eq(a, b) for a, b in zip(got["many"], want_many))
            if not ok:
                bad += 1
                print("MISMATCH", tag, got["vals"], want, flush=True)
        except Exception as e:  # pylint: disable=broad-except
            bad += 1
            print("ERROR", tag, type(e).__name__, str(e)[:300], flush=True)
    print("fuzz_ranks: %d trials, %d bad" % (trials, bad))
    sys.exit(1 if bad else 0)
