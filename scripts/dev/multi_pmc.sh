#!/bin/bash
# SQ counters of the multi-point fused kernel (and the single-point one beside it): K points per sweep
cd /tmp && export TMPDIR=/tmp
K=${1:-5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/multi_pmc; mkdir -p $OUT
pass () {  # tag, counters...
  tag=$1; shift
  rm -rf /tmp/mp_$tag
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/mp_$tag -o p -- python3 $GRAFT_REPO_ROOT/scripts/dev/multi_probe.py 1e7 $K > /tmp/mp_$tag.log 2>&1
  cp /tmp/mp_$tag/p_counter_collection.csv $OUT/pmc_${tag}_K$K$SUFFIX.csv
}
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
python3 - $OUT $K "$SUFFIX" <<'PY'
import csv, sys, collections
out, k, suf = sys.argv[1], sys.argv[2], sys.argv[3]
for tag in "ab":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("%s/pmc_%s_K%s%s.csv" % (out, tag, k, suf))):
        name = r["Kernel_Name"]
        if "hist_accumulate" not in name:
            continue
        key = "multi" if "multi" in name else "single"
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, d in acc.items():
        print(key, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
