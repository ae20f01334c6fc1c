export TMPDIR=/tmp
FP_IGEMM_TILE=pps256x256 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "igemm or hip_encoder or fp16_plans" 2>&1 | tail -4
for t in auto pps256x256; do
  echo "== $t"
  FP_IGEMM_TILE=$t timeout 200 python scripts/bench_igemm.py 2>&1 | grep "igemm\|HipEnc\|RefinePlan"
done
