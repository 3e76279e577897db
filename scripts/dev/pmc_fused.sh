# SQ-level counters of hist_accumulate_kernel (run on the GPU box); one pass per group
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fused; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" \
           "SQ_INST_LEVEL_VMEM TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $GRAFT_REPO_ROOT/scripts/dev/dev_pmc_target.py > /dev/null 2> $OUT/p$i.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/p_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "hist_accumulate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # several rows per dispatch (per dimension) may exist: sum per dispatch = total / n_dispatch
        print(f.split("/")[-2], k, "total", sum(v), "rows", len(v))
PY
