export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "igemm or hip_encoder or fp16_plans" 2>&1 | tail -3
for t in 128x128 256x256 256x128 512x128; do echo "== $t"; FP_IGEMM_TILE=$t timeout 200 python scripts/bench_igemm.py 2>&1 | grep -E "igemm conv.*res=0|igemm linear|HipEncoder|RefinePlan"; done
echo "== auto"; timeout 200 python scripts/bench_igemm.py 2>&1 | grep -E "HipEncoder|RefinePlan"
