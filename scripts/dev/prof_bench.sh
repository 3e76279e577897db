# rocprofv3 kernel stats of the default bench (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 bench.py --no-cpu-baseline ${@:2} > $OUT/bench_under_rocprof.json 2> $OUT/stderr.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.reader(open("$OUT/kernel_stats.csv")))[:14]:
    print(r[0][:70].ljust(70), r[1:5])
PY
