export TMPDIR=/tmp
for t in 128x128k32x3 128x128k32x4; do echo "== tests $t"; FP_IGEMM_TILE=$t timeout 300 python -m pytest tests -m gpu -x -q -k "igemm or hip_encoder" 2>&1 | tail -2; done
for t in 128x128 128x128k32x3 128x128k32x4; do echo "== $t"; FP_IGEMM_TILE=$t timeout 200 python scripts/bench_igemm.py 2>&1 | grep -E "igemm conv.*res=0|igemm linear|HipEncoder|RefinePlan"; done
