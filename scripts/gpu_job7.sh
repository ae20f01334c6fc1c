export TMPDIR=/tmp
for t in 128x2 256x3 256x2; do echo "== $t"; FP_IGEMM_TILE=$t timeout 200 python scripts/bench_igemm.py 2>&1 | grep -E "igemm conv|igemm linear|HipEncoder|RefinePlan"; done
timeout 300 python -m pytest tests -m gpu -x -q -k "igemm or hip_encoder or fp16_plans" 2>&1 | tail -2
