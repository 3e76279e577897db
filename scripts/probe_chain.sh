export PYTHONPATH=.
for f in 0 1; do for g in 1 2 4; do echo -n "two-sided G=$g fma=$f: "; PISA_HIP_CHAIN_FMA=$f PISA_HIP_CHAIN_TWO_SIDED=1 PISA_HIP_CHAIN_GROUPS=$g python scripts/dev_probe8.py; done; done
PISA_HIP_CHAIN_FMA=1 PISA_HIP_CHAIN_TWO_SIDED=1 PISA_HIP_CHAIN_GROUPS=2 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
