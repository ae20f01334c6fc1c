# round 3, call A: parity work of this round on the GPU (targeted tests) + a baseline bench line on this box
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_geometry_vs_reference_golden.py tests/test_gpu_parity.py tests/test_gpu_amp.py -m gpu -q --durations=15 \
  -k "guess or pose_update or zbuffer or fp16_output or full_frame or degenerate or large_mesh or shim or wide or fp32_matches or eight or rasteriser_is_exact or sub_batches or three_way or free_running or small_batches" \
  > gpurun_out/r3a_pytest.log 2>&1; tail -25 gpurun_out/r3a_pytest.log
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; tail -2 gpurun_out/r3a_bench.err; cut -c1-600 gpurun_out/r3a_bench.json
echo "total seconds: $(( $(date +%s) - T0 ))"
