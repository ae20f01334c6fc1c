export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "igemm or hip_encoder or fp16_plans" 2>&1 | tail -6
FP_CONV3X3_BN=128 timeout 300 python -m pytest tests -m gpu -x -q -k "igemm or hip_encoder" 2>&1 | tail -3
for v in "FP_CONV3X3=0" "FP_CONV3X3_BN=128" "FP_CONV3X3_BN=256"; do echo "== $v"; for l in 128 256 512; do echo -n "layer $l: "; env $v FP_LAYER=$l python scripts/bench_one.py 2>&1 | grep TFLOP; done; done
timeout 200 python scripts/bench_igemm.py 2>&1 | grep -E "HipEncoder|RefinePlan"
