# round 2, GPU call A: new kernels first (short timeouts), then per-layer GEMM A/B, bench, BASELINE-size parity, old suite
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== kernels (conv / linear)"; timeout 400 python -m pytest tests/test_gpu_amp.py -q -k "conv3x3 or conv7x7 or linear_policy" > gpurun_out/a1_kernels.log 2>&1; tail -25 gpurun_out/a1_kernels.log
echo "== rowops / attention / encoder / plans"; timeout 400 python -m pytest tests/test_gpu_amp.py -q -k "rowops or attention or encoder or plans" > gpurun_out/a2_nets.log 2>&1; tail -25 gpurun_out/a2_nets.log
echo "== per-layer igemm: product"; timeout 200 python scripts/bench_igemm.py > gpurun_out/a3_igemm_sw.log 2>&1; cat gpurun_out/a3_igemm_sw.log | tail -16
echo "== per-layer igemm: generic (profile lib)"; FP_AMD_LIB=$PWD/foundationpose_amd/csrc/libfp_amd_profile.so FP_IGEMM_TILE=generic timeout 200 python scripts/bench_igemm.py > gpurun_out/a3_igemm_generic.log 2>&1; tail -16 gpurun_out/a3_igemm_generic.log
echo "== bench"; timeout 400 python bench.py --no-cpu-baseline > gpurun_out/a4_bench.json 2> gpurun_out/a4_bench.err; tail -3 gpurun_out/a4_bench.err; cut -c1-600 gpurun_out/a4_bench.json
echo "== 252"; timeout 900 python -m pytest tests/test_gpu_amp.py -q -k "252" > gpurun_out/a5_252.log 2>&1; tail -30 gpurun_out/a5_252.log
echo "== old suite"; timeout 600 python -m pytest tests/test_gpu_parity.py -q > gpurun_out/a6_parity.log 2>&1; tail -30 gpurun_out/a6_parity.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
