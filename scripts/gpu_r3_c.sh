# round 3, call C: (1) A/B of the lock-step kernels, (2) the parity tests of this round (per-test timeout)
export TMPDIR=/tmp
bash scripts/gpu_r3_b.sh 2>&1 | tail -60
T0=$(date +%s)
timeout 1300 python -m pytest tests/test_geometry_vs_reference_golden.py tests/test_gpu_parity.py tests/test_gpu_amp.py -m gpu -q --timeout 420 --durations=12 \
  -k "guess or pose_update or shim or wide or fp32_matches or eight or rasteriser_is_exact or sub_batches or three_way or free_running or small_batches" \
  > gpurun_out/r3c_pytest.log 2>&1; tail -30 gpurun_out/r3c_pytest.log | cut -c1-300
echo "pytest seconds: $(( $(date +%s) - T0 ))"
