# R6-3: A/B of the chain -> accumulate hand-over on one box, kernel-trace timelines of both modes, a 1e5-evaluation hang probe
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/overlap; rm -rf $OUT; mkdir -p $OUT
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
for rep in 1 2; do for m in 0 1 2; do
  PISA_HIP_EVAL_OVERLAP=$m timeout 300 python3 scripts/dev/overlap_ab.py 8 500 | tail -1 | tee -a $OUT/ab.jsonl
done; done
for m in 0 1 2; do
  export PISA_HIP_EVAL_OVERLAP=$m
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$m -o t -- python3 scripts/dev/overlap_ab.py 1 300 > $OUT/trace$m.log 2>&1
  python3 - $OUT/tr$m $m <<'PY' | tee -a $OUT/timeline.txt
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
keys = ("prob3_terms", "prob3_chain", "hist_accumulate", "finalize_metric")
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in keys)]
steps, cur = [], {}
for r in rows:
    k = next(k for k in keys if k in r["Kernel_Name"])
    if k == "prob3_terms" and cur:
        cur = {}
    cur[k] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    if len(cur) == 4 and k == "finalize_metric":
        steps.append(cur); cur = {}
steps = steps[len(steps) // 3:]
def med(fn): return statistics.median(fn(s) for s in steps) / 1e3
print("mode %s: %d steps" % (sys.argv[2], len(steps)))
for name, fn in (("terms", lambda s: s["prob3_terms"][1] - s["prob3_terms"][0]),
                 ("chain", lambda s: s["prob3_chain"][1] - s["prob3_chain"][0]),
                 ("accumulate", lambda s: s["hist_accumulate"][1] - s["hist_accumulate"][0]),
                 ("tail", lambda s: s["finalize_metric"][1] - s["finalize_metric"][0]),
                 ("accumulate start - chain start", lambda s: s["hist_accumulate"][0] - s["prob3_chain"][0]),
                 ("accumulate start - chain end", lambda s: s["hist_accumulate"][0] - s["prob3_chain"][1]),
                 ("accumulate end - chain end", lambda s: s["hist_accumulate"][1] - s["prob3_chain"][1]),
                 ("tail start - accumulate end", lambda s: s["finalize_metric"][0] - s["hist_accumulate"][1]),
                 ("terms start -> tail end", lambda s: s["finalize_metric"][1] - s["prob3_terms"][0])):
    print("  %-32s median %7.2f us" % (name, med(fn)))
PY
  cp $(ls $OUT/tr$m/*kernel_trace.csv | head -1) $OUT/kernel_trace_mode$m.csv; gzip -f $OUT/kernel_trace_mode$m.csv
  rm -rf $OUT/tr$m
done
unset PISA_HIP_EVAL_OVERLAP
# hang probe: 1e5 evaluations in the better overlap mode
best=$(python3 -c "
import json
r={}
for l in open('$OUT/ab.jsonl'):
    d=json.loads(l); r.setdefault(d['mode'],[]).append(d['us_per_eval'])
print(min(('1','2'), key=lambda m: min(r[m])))")
PISA_HIP_EVAL_OVERLAP=$best timeout 600 python3 scripts/dev/overlap_ab.py 200 500 | tail -1 | tee $OUT/hang_probe_mode$best.json
