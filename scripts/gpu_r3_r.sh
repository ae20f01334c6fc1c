# round 3, call R: the fused out_proj + residual + LayerNorm kernel -- bit-identity test, then the bench without / with it
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 50 -k "linear_layernorm" > gpurun_out/r3r_pytest.log 2>&1; tail -15 gpurun_out/r3r_pytest.log | cut -c1-300
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 40 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r3r_bench_plain.json 2> /dev/null; cut -c1-330 gpurun_out/r3r_bench_plain.json
FP_AMD_FUSED_LN=1 timeout 40 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r3r_bench_fused_ln.json 2> gpurun_out/r3r_bench_fused_ln.err; cut -c1-330 gpurun_out/r3r_bench_fused_ln.json; tail -3 gpurun_out/r3r_bench_fused_ln.err | cut -c1-300
echo "total seconds: $(( $(date +%s) - T0 ))"
