"""the numbers DESIGN.md / README.md quote from profiles/<tag> (default r2_v3), in one place"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2_v3"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", tag)
d = json.load(open(os.path.join(root, "bench.json")))
print({k: d[k] for k in ("value", "ms_per_step", "last_llh", "pipelined_evals_per_s", "phase_ms")})
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["traffic"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread"]["value"])
print(d["unbinned_events_dropped"])
keep = ("evals_per_s", "ms_per_step", "prob3_events_kernel_ms", "boundary_over_engine", "engine_same_workload_evals_per_s",
        "stage_protocol_every_step_evals_per_s", "osc_only", "flux_moves", "all_free")
for k, v in d["legs"].items():
    r = v.get("roofline")
    print(k, {kk: vv for kk, vv in v.items() if kk in keep},
          r and {kk: r[kk] for kk in ("achieved", "frac", "avg_launch_ms", "flop_per_event") if kk in r})
print("events flop", json.load(open(os.path.join(root, "events_flops.json")))["std"])
for f in ("traffic.json", "traffic_l3_exceeding.json", "traffic_coordinate_form.json"):
    t = json.load(open(os.path.join(root, f)))
    print(f, t["hbm_bytes"], t["ratio_to_algorithmic"])
k = json.load(open(os.path.join(root, "kernels_by_phase.json")))
print({n: v["timed_loop_mean_us"] for n, v in k.items() if isinstance(v, dict)})
kf = json.load(open(os.path.join(root, "kde_flops.json")))
print("kde flop", kf["fp64_flop_per_evaluation"], {n: v for n, v in list(kf["per_kernel_flop_per_evaluation"].items())[:5]})
