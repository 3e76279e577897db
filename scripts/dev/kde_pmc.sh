#!/bin/bash
# SQ counters of the KDE kernels of one C3 evaluation (kde_facts.py): where the lattice kernel's cycles go
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kde_pmc; mkdir -p $OUT
pass () {  # tag, counters...
  tag=$1; shift
  rm -rf /tmp/kp_$tag
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/kp_$tag -o p -- python3 $GRAFT_REPO_ROOT/scripts/dev/kde_facts.py 1e7 ${NC:-3} > /tmp/kp_$tag.log 2>&1
  cp /tmp/kp_$tag/p_counter_collection.csv $OUT/pmc_$tag.csv
}
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pass c SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass d SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
python3 - $OUT <<'PY'
import csv, sys, collections, json
out = sys.argv[1]
keys = ("kde_lattice_kernel", "kde_local_pilot", "kde_h2l_mfma_kernel<0>", "kde_h2l_mfma_kernel<1>", "kde_hermite_coef")
res = collections.defaultdict(dict)
for tag in "abcd":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        rows = list(csv.DictReader(open("%s/pmc_%s.csv" % (out, tag))))
    except OSError:          # (a counter set this image does not have)
        rows = []
    for r in rows:
        for key in keys:
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if tag == "a" and r["Counter_Name"] == "SQ_WAVES":
                    acc[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key, d in acc.items():
        for c, v in d.items():
            res[key][c.lstrip("_")] = sum(v) / len(v)
for key, d in res.items():
    busy = d["SQ_BUSY_CYCLES"] / 32
    d["derived"] = {"valu_busy_fraction_of_launch": d["SQ_ACTIVE_INST_VALU"] / 1024 * 4 / busy,
                    "mean_wavefront_lifetime_fraction_of_launch": d["SQ_WAVE_CYCLES"] / d["SQ_WAVES"] * 4 / busy,
                    "clock_GHz": busy / d["dur_us"] / 1e3,
                    "lds_bank_conflict_cycles_per_lds_instruction": d.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(d.get("SQ_INSTS_LDS", 0.0), 1.0)}
    print(key, {k: round(v, 3) for k, v in d["derived"].items()}, "dur_us %.1f" % d["dur_us"])
json.dump({"per_launch_means": res,
           "method": "scripts/dev/kde_pmc.sh: four separate rocprofv3 --pmc passes (with --kernel-trace) over scripts/dev/kde_facts.py 1e7 3 "
                     "(the estimators of 3 containers of the C3 workload, one stream); SQ_BUSY_CYCLES / 32 = launch length in cycles, "
                     "SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES in quad-cycles, 1 024 SIMDs"}, open(out + "/kde_sq_counters.json", "w"), indent=1)
PY
