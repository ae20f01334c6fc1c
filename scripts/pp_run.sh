export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "attention_kernel or fp16_plans" 2>&1 | tail -3
timeout 200 python scripts/bench_igemm.py 2>&1 | grep "RefinePlan\|attention"
