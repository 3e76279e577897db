"""Multi-rank cases that need real processes (one per rank), started by the tests through `torch.distributed.run`:

    python -m torch.distributed.run --nproc-per-node N ... tests/dist_cases.py <case> <out_dir> [args]

`fit_point_groups <out_dir> <n_groups>`: every rank on HIP device 0, exchanging over gloo (the one-GPU stand-in of one rank
per GPU, as in tests/test_gpu_distributed.py): `Analysis.fit_hypo(batched_gradient=True)` on the cfg-text pipeline under
`engine.configure_point_groups(n_groups)` -- the stencil of every iterate dealt to the groups by the engine the hist stage
builds for itself -- and, on rank 0, the fit history for the test to compare with the single-rank fit.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def fit_point_groups(out_dir, n_groups, n_events="1.2e5"):
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from pisa_amd import engine
    from pisa_amd.analysis.analysis import Analysis
    from pisa_amd.core.config_parser import parse_pipeline_config
    from pisa_amd.core.distribution_maker import DistributionMaker
    from pisa_amd.core.units import ureg

    engine.configure_point_groups(int(n_groups))
    cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
    cfg[("data", "synthetic_events")]["params"].params.n_events.value = float(n_events)
    dm = DistributionMaker(cfg)
    for name in dm.params.free.names:
        if name not in ("theta23", "deltam31"):
            dm.params.fix(name)
    dm.params.theta23.value = 47.5 * ureg.degree
    dm.params.deltam31.value = 2.55e-3 * ureg.eV ** 2
    data = dm.get_outputs(return_sum=True).fluctuate("poisson", random_state=0)
    dm.params.theta23.value = 42.3 * ureg.degree
    dm.params.deltam31.value = 2.457e-3 * ureg.eV ** 2
    res = Analysis().fit_hypo(data, dm, "llh", reset_free=False, batched_gradient=True)
    eng = dm.pipelines[0]["hist"]._engine
    pg = eng.points
    out = {"rank": rank, "world": world, "topology": None if pg is None else pg.topology,
           "engine_world": eng.world_size, "sweeps": getattr(eng, "sweeps", None),
           "history": [[float(v) for v in row] for row in res.fit_history], "metric_val": float(res.metric_val),
           "evaluations": int(res.num_distributions_generated)}
    with open(os.path.join(out_dir, "fit_r%d.json" % rank), "w") as fh:
        json.dump(out, fh)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    {"fit_point_groups": fit_point_groups}[sys.argv[1]](*sys.argv[2:])
