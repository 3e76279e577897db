# LDS counters of the fused kernel with and without the bank-aware event order (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
for o in 0 1; do
  export PISA_LDS_ORDER=$o
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$o; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/scripts/dev/dev_pmc_target.py > /dev/null 2> $OUT/log.txt
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "hist_accumulate" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("lds_order=$o", {k: sum(v) / len(v) for k, v in acc.items()})
PY
done
