# round 3, call G: attention with the V scatter under the second product: correctness + A/B against the round-2 kernel + bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_amp.py tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "attention or plans_match or write_only or scorer_single" > gpurun_out/r3g_pytest_att.log 2>&1; tail -4 gpurun_out/r3g_pytest_att.log | cut -c1-250
for rep in 1 2; do
  timeout 100 python scripts/bench_attention.py
  FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_prevatt.so timeout 100 python scripts/bench_attention.py
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3g_attention_ab.log | cut -c1-200
for rep in 1 2 3; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['clock']['sclk_MHz_mean'])"
  FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_prevatt.so timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r02-attention', d['ms_per_step'], d['clock']['sclk_MHz_mean'])"
done
