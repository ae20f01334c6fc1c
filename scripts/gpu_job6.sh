export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "igemm or hip_encoder or fp16_plans" 2>&1 | tail -3
for t in 128x2 128x3 256x2 256x3; do echo "== $t"; FP_IGEMM_TILE=$t timeout 200 python scripts/bench_igemm.py 2>&1 | grep -E "igemm|HipEncoder|RefinePlan"; done
for t in 128x3 256x2; do echo "== tests $t"; FP_IGEMM_TILE=$t timeout 300 python -m pytest tests -m gpu -x -q -k "igemm or hip_encoder" 2>&1 | tail -2; done
