export PYTHONPATH=.
for ch in 1 2 3 4 8; do echo -n "ch=$ch: "; PISA_HIP_PROB3_CH=$ch python scripts/dev/dev_probe8.py; done
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_prob3; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 scripts/dev/dev_probe8.py > /dev/null 2> $OUT/stderr.log
python3 - <<PY
import csv
for r in list(csv.reader(open("$OUT/p_kernel_stats.csv")))[:4]:
    print(r[0][:60].ljust(60), r[1:5])
PY
