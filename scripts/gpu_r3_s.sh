# round 3, call S: the fused out_proj + residual + LayerNorm kernel with prefetched residual rows -- bit-identity test, the plan /
# predictor tests with the fused path switched on, then the bench without / with it
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 50 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 40 -k "linear_layernorm" > gpurun_out/r3s_pytest.log 2>&1; tail -3 gpurun_out/r3s_pytest.log | cut -c1-300
FP_AMD_FUSED_LN=1 timeout 80 python -m pytest tests -m gpu -q --timeout 60 -k "plans_match or hip_encoder or estimator_api or sub_batches or shared_observed or graphed_tracker" > gpurun_out/r3s_pytest_fused.log 2>&1; tail -3 gpurun_out/r3s_pytest_fused.log | cut -c1-300
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 40 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r3s_bench_plain.json 2> /dev/null; cut -c1-330 gpurun_out/r3s_bench_plain.json
FP_AMD_FUSED_LN=1 timeout 40 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r3s_bench_fused_ln.json 2> /dev/null; cut -c1-330 gpurun_out/r3s_bench_fused_ln.json
echo "total seconds: $(( $(date +%s) - T0 ))"
