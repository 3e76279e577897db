export PYTHONPATH=.
for t in 1024 512; do for b in 512 768 1024 1536 2048; do
echo "threads=$t blocks=$b"; PISA_HIP_HIST_THREADS=$t PISA_HIP_HIST_BLOCKS=$b python scripts/dev_probe5.py 1e7 node 2>&1 | grep order
done; done
