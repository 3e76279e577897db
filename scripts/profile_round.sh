# Full measurement set of the default bench for profiles/<tag> (run on the GPU box):
#   bench.json                  plain bench.py run (the judged line, with all legs)
#   kernel_stats.csv            timeout 600 rocprofv3 --kernel-trace --stats of the headline loop (bench.py --legs none)
#   kernels_by_phase.json       per-kernel mean over the timed loop from the same trace
#   pmc_FETCH_SIZE.csv / pmc_WRITE_SIZE.csv   separate --pmc passes (bench.py --steps 3 --no-kernel-timing)
#   traffic.json                HBM bytes per launch of the dominant kernel derived from them
#   traffic_l3_exceeding.json / traffic_coordinate_form.json   the same for two of the legs
#   events_flops.json           executed fp64 lane operations per event of prob3_events_kernel
#                               (SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 pass over scripts/bench_events.py)
#   kde_kernel_stats.csv        kernel trace statistics of the C3 (KDE on) pipeline
#   kde_flops.json              executed fp64 flops of all kde_* kernels of one C3 evaluation (counter pass)
#   kde_sq_counters.json        SQ counters of the lattice / pilot / translation / coefficient kernels (scripts/dev/kde_pmc.sh)
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.stderr
tail -c 400 $OUT/bench.json
cp bench_detail.json $OUT/bench_detail.json
# every profiled run below: set-up uploads on the calling thread (counter-collecting profilers serialise dispatches and have
# been seen to stall with several submitting host threads) -- set in this script's environment, never as a hop after `--`
export PISA_HIP_UPLOAD_THREADS=0
LEAN="--no-cpu-baseline --no-drop-probe --no-batch-probe --legs none"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py $LEAN > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp $OUT/stats/bench_kernel_stats.csv $OUT/kernel_stats.csv
pmc_pair () {  # name, extra bench args
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$1_$c -o p -- python3 bench.py --steps 3 --warmup 1 --no-kernel-timing $LEAN $2 > /dev/null 2> $OUT/pmc_$1_$c.log
    cp $OUT/pmc_$1_$c/p_counter_collection.csv $OUT/pmc_$1_$c.csv
  done
}
pmc_pair main ""
pmc_pair l3 "--events 4e7"
pmc_pair coord "--coordinate-form"
pmc_pair fine "--binning fine3d"
for v in std nsi decay; do
  F=""; [ $v = nsi ] && F="--nsi"; [ $v = decay ] && F="--decay"
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_events_$v -o p -- python3 scripts/bench_events.py --events 1e6 --steps 3 --warmup 1 $F > $OUT/events_$v.json 2> $OUT/pmc_events_$v.log
  cp $OUT/pmc_events_$v/p_counter_collection.csv $OUT/pmc_events_$v.csv
done
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_kde -o p -- python3 scripts/bench_kde.py 2e5 > $OUT/kde_pmc_run.json 2> $OUT/pmc_kde.log
cp $OUT/pmc_kde/p_counter_collection.csv $OUT/pmc_kde.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kde_stats -o kde -- python3 scripts/dev/c3_probe.py 1e7 > $OUT/c3_probe.log 2> $OUT/kde_stats.log
cp $OUT/kde_stats/kde_kernel_stats.csv $OUT/kde_kernel_stats.csv
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $OUT/pmc_c3 -o p -- python3 scripts/dev/c3_probe.py 1e7 > $OUT/c3_pmc_run.log 2> $OUT/pmc_c3.log
cp $OUT/pmc_c3/p_counter_collection.csv $OUT/pmc_c3_full.csv
# SQ counters of the KDE kernels (VALU occupancy, wavefront lifetimes, LDS conflicts): kde_sq_counters.json
bash scripts/dev/kde_pmc.sh > $OUT/kde_pmc.log 2>&1; cp gpurun_out/kde_pmc/kde_sq_counters.json $OUT/kde_sq_counters.json
# ---- round 3: kernel-trace durations for every leg's kernel (the legs' roofline fractions were HIP-event
# numbers without a trace counterpart), the event kernel (C2 / C5 sizes, default and decay instantiation)
# with its SQ counters, the multi-point kernels, the one-pass flux refresh
trace () {  # name, command...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$name -o t -- "$@" > $OUT/run_$name.json 2> $OUT/tr_$name.log
  cp $OUT/tr_$name/t_kernel_stats.csv $OUT/kernel_stats_$name.csv
  rm -rf $OUT/tr_$name
}
trace l3 python3 bench.py $LEAN --no-kernel-timing --steps 100 --events 4e7
trace exact python3 bench.py $LEAN --no-kernel-timing --steps 200 --exact-association
trace coord python3 bench.py $LEAN --no-kernel-timing --steps 100 --coordinate-form
trace fine python3 bench.py $LEAN --no-kernel-timing --steps 200 --binning fine3d
trace update_flux python3 bench.py --no-cpu-baseline --no-drop-probe --no-batch-probe --no-kernel-timing --steps 50 --legs update_flux
trace events_c2 python3 scripts/bench_events.py --events 1e6 --steps 20
trace events_c5 python3 scripts/bench_events.py --events 1.25e7 --nsi --steps 6
trace events_c5_full python3 scripts/bench_events.py --events 1e8 --nsi --steps 3 --on-device
trace events_c2_decay python3 scripts/bench_events.py --events 1e6 --steps 20 --decay
trace events_c5_decay python3 scripts/bench_events.py --events 1.25e7 --nsi --steps 6 --decay
for K in 3 5 9; do trace multi_K$K python3 scripts/dev/multi_probe.py 1e7 $K; done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_events_sq -o p -- python3 scripts/bench_events.py --events 1.25e7 --nsi --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_events_sq.log
cp $OUT/pmc_events_sq/p_counter_collection.csv $OUT/pmc_events_sq.csv
rm -rf $OUT/pmc_events_sq
python3 - <<PY
import csv, json
OUT = "$OUT"
# SQ counters of prob3_events_kernel (C5 size): issue utilisation of the kernel
acc = {}
for r in csv.DictReader(open(OUT + "/pmc_events_sq.csv")):
    if "prob3_events_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
sq = {k: sum(v) / len(v) for k, v in acc.items()}
if sq:
    sq["launches"] = len(next(iter(acc.values())))
    # SQ_WAVE_CYCLES / SQ_BUSY_CYCLES are in units of 4 clocks per the counter definitions of this tool; ratios are unit free
    sq["valu_issue_fraction_of_wave_cycles"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
    sq["waiting_fraction_of_wave_cycles"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
    sq["valu_instructions_per_wave"] = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"]
    sq["method"] = ("timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY "
                    "over scripts/bench_events.py --events 1.25e7 --nsi; mean per launch of prob3_events_kernel")
json.dump(sq, open(OUT + "/events_sq_counters.json", "w"), indent=1)
print("events SQ", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sq.items() if k != "method"})
# one table of trace durations: kernel -> mean us, per run
table = {}
import glob, os
for f in sorted(glob.glob(OUT + "/kernel_stats_*.csv")):
    name = os.path.basename(f)[len("kernel_stats_"):-4]
    rows = [r for r in csv.DictReader(open(f)) if "pisa::" in r["Name"]]
    table[name] = {r["Name"].split("(")[0].replace("void ", ""): {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
                   for r in rows}
json.dump(table, open(OUT + "/trace_durations.json", "w"), indent=1)
PY
python3 - <<PY
import csv, json
OUT = "$OUT"
def per_launch(path, kernel, name):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
         if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return (sum(v) / len(v), len(v)) if v else (float("nan"), 0)
METHOD = ("timeout 600 rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 3 --no-kernel-timing); "
          "FETCH_SIZE is in KiB and on gfx950 reports half of the bytes of 16-B/lane coalesced streams "
          "(MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact")
def traffic(tag, label, alg, fname):
    f, nf = per_launch("%s/pmc_%s_FETCH_SIZE.csv" % (OUT, tag), "hist_accumulate_kernel", "FETCH_SIZE")
    w, nw = per_launch("%s/pmc_%s_WRITE_SIZE.csv" % (OUT, tag), "hist_accumulate_kernel", "WRITE_SIZE")
    d = {"kernel": label, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": [nf, nw],
         "fetch_bytes_corrected": f * 1024 * 2, "write_bytes": w * 1024, "hbm_bytes": f * 1024 * 2 + w * 1024,
         "method": METHOD, "algorithmic_bytes": alg, "ratio_to_algorithmic": (f * 1024 * 2 + w * 1024) / alg}
    json.dump(d, open("%s/%s" % (OUT, fname), "w"), indent=1)
    print(fname, d["hbm_bytes"], d["ratio_to_algorithmic"], d["launches"])
traffic("main", "hist_accumulate_kernel<7,true> (16-bit index layout, 20 B/event), 9999996 events", 20 * 9999996, "traffic.json")
traffic("l3", "hist_accumulate_kernel<7,true>, 39999996 events (800 MB resident: beyond the 256 MiB Infinity Cache)", 20 * 39999996, "traffic_l3_exceeding.json")
traffic("coord", "hist_accumulate_kernel<1,true> (coordinate form, SURVEY 8(d) 72 B/event), 9999996 events", 72 * 9999996, "traffic_coordinate_form.json")
traffic("fine", "hist_accumulate_kernel<7,true>, 40x40x3 = 4 800 output bins (partitioned LDS-window order), 9999996 events", 20 * 9999996, "traffic_fine_binning.json")
# executed fp64 lane operations of prob3_events_kernel.  One evaluation = one launch per sign, each over
# the events of that sign: the counters are summed over all launches and divided by the summed grid
# sizes (threads = events, padded to whole wavefronts), i.e. per event of the launch it belongs to.
ev = {}
for v in ("std", "nsi", "decay"):
    tot, threads, launches = {}, {}, set()
    for r in csv.DictReader(open("%s/pmc_events_%s.csv" % (OUT, v))):
        if "prob3_events" not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        threads[r["Counter_Name"]] = threads.get(r["Counter_Name"], 0.0) + float(r["Grid_Size"])
        launches.add(r["Dispatch_Id"])
    per = {name: 64.0 * tot[name] / threads[name] for name in tot}   # per event and lane
    flop = (2 * per["SQ_INSTS_VALU_FMA_F64"] + per["SQ_INSTS_VALU_ADD_F64"] + per["SQ_INSTS_VALU_MUL_F64"]
            + per["SQ_INSTS_VALU_TRANS_F64"])
    ev[v] = {"instructions_per_event_lane": per, "launches": len(launches), "flop_per_event": flop,
             "valu_instructions_per_event_lane": per["SQ_INSTS_VALU"]}
ev["method"] = ("timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 SQ_INSTS_VALU over scripts/bench_events.py --events 1e6; "
                "wave-level instruction counts x 64 lanes summed over the launches, divided by the summed grid sizes "
                "(each launch covers the events of one sign); inactive lanes of partially filled or divergent waves "
                "are counted: an upper bound of the useful lane operations; FMA = 2 flop")
json.dump(ev, open(OUT + "/events_flops.json", "w"), indent=1)
print({k: (v["flop_per_event"] if isinstance(v, dict) else None) for k, v in ev.items()})
# calibration of the same counters on a kernel with a known instruction mix: kde_pairs_kernel
k = {}
for name in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU"):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(OUT + "/pmc_kde.csv"))
            if "kde_pairs_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    k[name] = sum(vals)
# executed fp64 flops of one C3 evaluation (all kde_* kernels; the probe runs 4 evaluations of 24 estimators)
rows = list(csv.DictReader(open(OUT + "/pmc_c3_full.csv")))
per_kernel = {}
for r in rows:
    name = r["Kernel_Name"]
    if "kde_" not in name:
        continue
    short = name.split("(")[0].replace("void ", "").replace("pisa::", "")
    wgt = {"SQ_INSTS_VALU_FMA_F64": 2.0, "SQ_INSTS_VALU_ADD_F64": 1.0, "SQ_INSTS_VALU_MUL_F64": 1.0,
           "SQ_INSTS_VALU_TRANS_F64": 1.0, "SQ_INSTS_VALU_MFMA_MOPS_F64": 512.0 / 64.0}.get(r["Counter_Name"])
    if wgt is None:
        continue
    d = per_kernel.setdefault(short, {"flop": 0.0, "dispatches": set(), "matrix_flop": 0.0})
    d["flop"] += 64.0 * wgt * float(r["Counter_Value"])
    if "MFMA" in r["Counter_Name"]:
        d["matrix_flop"] += 512.0 * float(r["Counter_Value"])
    d["dispatches"].add(r["Dispatch_Id"])
n_eval = 4
tot = sum(d["flop"] for d in per_kernel.values())
json.dump({"events": 9999996, "evaluations_in_run": n_eval, "fp64_flop_per_evaluation": tot / n_eval,
           "per_kernel_flop_per_evaluation": {k_: v["flop"] / n_eval for k_, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["flop"])},
           "launches_per_evaluation": {k_: len(v["dispatches"]) / n_eval for k_, v in per_kernel.items()},
           "matrix_core_flop_per_evaluation": sum(v["matrix_flop"] for v in per_kernel.values()) / n_eval,
           "method": "timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 over scripts/dev/c3_probe.py 1e7 "
                     "(4 evaluations of the 1e7-event pipeline with utils.kde); wave-level counts x 64 lanes, FMA = 2 flop, the matrix cores' "
                     "count x 512 flop (one v_mfma_f64_16x16x4_f64 = 2 048 flop = 4 counts), summed over every kernel whose name contains kde_"}, open(OUT + "/kde_flops.json", "w"), indent=1)
print("kde flop per evaluation %.3e" % (tot / n_eval))
# the raw rows of the KDE kernels only
with open(OUT + "/pmc_c3.csv", "w", newline="") as f:
    wr = csv.writer(f)
    wr.writerow(list(rows[0].keys()))
    for r in rows:
        if any(x in r["Kernel_Name"] for x in ("kde_lattice_kernel", "kde_local_pilot", "kde_h2l", "kde_hermite_coef")):
            wr.writerow(list(r.values()))
json.dump({"kde_pairs_kernel_totals": k, "run": open(OUT + "/kde_pmc_run.json").read()[-1500:]}, open(OUT + "/kde_counter_check.json", "w"), indent=1)
for r in list(csv.reader(open(OUT + "/kernel_stats.csv")))[:8]:
    print(r[0][:60].ljust(60), r[1:5])
rows = sorted(csv.DictReader(open(OUT + "/stats/bench_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
phase = {}
for key in ("hist_accumulate_kernel", "prob3_terms_kernel", "prob3_chain_kernel", "finalize_metric_kernel"):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]]
    loop = d[21:521]
    phase[key] = {"timed_loop_mean_us": sum(loop) / len(loop) / 1e3, "all_launches_mean_us": sum(d) / len(d) / 1e3,
                  "launches": len(d)}
phase["note"] = ("from bench_kernel_trace.csv of the rocprofv3 run of bench.py --no-cpu-baseline --no-drop-probe "
                 "--no-batch-probe --legs none; timed loop = launches 22..521 (after pseudo-data and 20 warm-up evaluations)")
json.dump(phase, open(OUT + "/kernels_by_phase.json", "w"), indent=1)
print({k: round(v["timed_loop_mean_us"], 2) for k, v in phase.items() if k != "note"})
PY
rm -rf $OUT/pmc_c3 $OUT/pmc_c3_full.csv $OUT/stats $OUT/pmc_*_FETCH_SIZE $OUT/pmc_*_WRITE_SIZE $OUT/pmc_events_std $OUT/pmc_events_nsi $OUT/pmc_kde $OUT/kde_stats
ls $OUT
# keep the merged output small (gpurun copies back at most 64 MiB): logs of the profiler runs are not evidence
find $OUT -name "*.log" -size +256k -delete
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kl $GRAFT_REPO_ROOT/gpurun_out/kde_pmc/pmc_*.csv
du -sk $GRAFT_REPO_ROOT/gpurun_out/* | sort -n | tail -4
du -sk $OUT/* | sort -n | tail -6
