cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/coord_traffic; rm -rf $OUT; mkdir -p $OUT
LEAN="--no-cpu-baseline --no-drop-probe --no-batch-probe --legs none"
python3 bench.py $LEAN --coordinate-form --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coord evals/s', round(d['value']), 'roofline', {k: d['roofline'][k] for k in ('achieved','frac','avg_launch_ms')})"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 bench.py --steps 3 --warmup 1 --no-kernel-timing $LEAN --coordinate-form > /dev/null 2> $OUT/pmc_$c.log
  cp $OUT/pmc_$c/p_counter_collection.csv $OUT/pmc_coord_$c.csv
done
python3 - <<'PY'
import csv
OUT="gpurun_out/coord_traffic"
def per_launch(path, kernel, name):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)
f,nf = per_launch(OUT+"/pmc_coord_FETCH_SIZE.csv","hist_accumulate_kernel","FETCH_SIZE")
w,nw = per_launch(OUT+"/pmc_coord_WRITE_SIZE.csv","hist_accumulate_kernel","WRITE_SIZE")
alg = 72*9999996
print("coordinate form: fetch %.1f MB (corrected x2) write %.2f MB; ratio to algorithmic %.4f; launches %d %d" % (f*1024*2/1e6, w*1024/1e6, (f*1024*2+w*1024)/alg, nf, nw))
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k headline 2>&1 | tail -2
