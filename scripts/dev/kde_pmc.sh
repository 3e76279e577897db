#!/bin/bash
# SQ counters of the KDE kernels of one C3 evaluation (kde_facts.py): where the lattice kernel's cycles go
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kde_pmc; mkdir -p $OUT
pass () {  # tag, counters...
  tag=$1; shift
  rm -rf /tmp/kp_$tag
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/kp_$tag -o p -- python3 $GRAFT_REPO_ROOT/scripts/dev/kde_facts.py 1e7 ${NC:-3} > /tmp/kp_$tag.log 2>&1
  cp /tmp/kp_$tag/p_counter_collection.csv $OUT/pmc_$tag.csv
}
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pass c SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python3 - $OUT <<'PY'
import csv, sys, collections
out = sys.argv[1]
for tag in "abc":
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("%s/pmc_%s.csv" % (out, tag))):
        name = r["Kernel_Name"]
        for key in ("kde_lattice_kernel", "kde_local_pilot", "kde_h2l_kernel<20, 0>", "kde_h2l_kernel<20, 1>", "kde_hermite_coef"):
            if key in name:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, d in acc.items():
        print(tag, key, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
    if tag == "a":
        for key, d in acc.items():
            n = len(d["SQ_WAVES"])
            valu = sum(d["SQ_ACTIVE_INST_VALU"]) / n / 1024 * 4; busy = sum(d["SQ_BUSY_CYCLES"]) / n / 32
            print("   ", key, "VALU busy fraction %.2f, wait fraction of wave cycles %.2f, mean wave lifetime / launch %.2f" % (
                valu / busy, sum(d["SQ_WAIT_INST_ANY"]) / sum(d["SQ_WAVE_CYCLES"]), sum(d["SQ_WAVE_CYCLES"]) / sum(d["SQ_WAVES"]) * 4 / busy))
PY
