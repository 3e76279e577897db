export PYTHONPATH=.
for ch in 1 2 4 8 16 64; do echo "ch=$ch"; PISA_HIP_PROB3_CH=$ch python scripts/dev_probe8.py; done
