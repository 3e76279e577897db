# Full measurement set of the default bench for profiles/<tag> (run on the GPU box):
#   bench.json                  plain bench.py run (the judged line)
#   kernel_stats.csv            rocprofv3 --kernel-trace --stats of the same command
#   bench_under_rocprof.json    the bench line printed under the profiler
#   pmc_FETCH_SIZE.csv / pmc_WRITE_SIZE.csv   separate --pmc passes (bench.py --steps 3 --no-kernel-timing)
#   traffic.json                HBM bytes per launch of the dominant kernel derived from them
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.stderr
tail -c 600 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --no-drop-probe --no-batch-probe > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp $OUT/stats/bench_kernel_stats.csv $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-drop-probe --no-batch-probe > /dev/null 2> $OUT/pmc_$c.log
  cp $OUT/pmc_$c/p_counter_collection.csv $OUT/pmc_$c.csv
done
python3 - <<PY
import csv, json
def per_launch(path, name):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
         if "hist_accumulate_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)
f, nf = per_launch("$OUT/pmc_FETCH_SIZE.csv", "FETCH_SIZE")
w, nw = per_launch("$OUT/pmc_WRITE_SIZE.csv", "WRITE_SIZE")
d = {"kernel": "hist_accumulate_kernel<7,true> (16-bit index layout, 20 B/event)", "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": [nf, nw],
     "fetch_bytes_corrected": f * 1024 * 2, "write_bytes": w * 1024,
     "hbm_bytes": f * 1024 * 2 + w * 1024,
     "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 3 --no-kernel-timing); "
               "FETCH_SIZE is in KiB and on gfx950 reports half of the bytes of 16-B/lane coalesced streams "
               "(MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact",
     "algorithmic_bytes": 20 * 9999996}
json.dump(d, open("$OUT/traffic.json", "w"), indent=1)
print(d["hbm_bytes"], d["launches"])
for r in list(csv.reader(open("$OUT/kernel_stats.csv")))[:8]:
    print(r[0][:60].ljust(60), r[1:5])
# per-kernel mean over the timed loop only (launches after the pseudo-data evaluation and the
# 20 warm-up evaluations), from the kernel trace of the same rocprofv3 run
rows = sorted(csv.DictReader(open("$OUT/stats/bench_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
phase = {}
for key in ("hist_accumulate_kernel", "prob3_terms_kernel", "prob3_chain_kernel", "finalize_metric_kernel"):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]]
    loop = d[21:521]
    phase[key] = {"timed_loop_mean_us": sum(loop) / len(loop) / 1e3, "all_launches_mean_us": sum(d) / len(d) / 1e3,
                  "launches": len(d)}
phase["note"] = ("from bench_kernel_trace.csv of the rocprofv3 run of bench.py --no-cpu-baseline --no-drop-probe "
                 "--no-batch-probe; timed loop = launches 22..521 (after pseudo-data and 20 warm-up evaluations)")
json.dump(phase, open("$OUT/kernels_by_phase.json", "w"), indent=1)
print({k: round(v["timed_loop_mean_us"], 2) for k, v in phase.items() if k != "note"})
PY
rm -rf $OUT/stats/*.db $OUT/pmc_*/ 2>/dev/null
