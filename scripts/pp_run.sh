export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "attention_kernel" 2>&1 | tail -6
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "fp16_plans or graph or scorer" 2>&1 | tail -3
timeout 200 python scripts/bench_igemm.py 2>&1 | grep "HipEnc\|RefinePlan\|attention\|   fp_"
FP_ATTENTION=torch timeout 200 python scripts/bench_igemm.py 2>&1 | grep "RefinePlan"
