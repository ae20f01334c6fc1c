# round 3, call F: the two-query-tiles-per-wave attention kernel: correctness (product default QT=2) and A/B against QT=1
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_amp.py tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "attention or plans_match or write_only" > gpurun_out/r3f_pytest_att.log 2>&1; tail -8 gpurun_out/r3f_pytest_att.log | cut -c1-250
P=foundationpose_amd/csrc/libfp_amd_profile.so
for rep in 1 2; do
  FP_AMD_LIB=$P FP_ATT_QT=1 timeout 100 python scripts/bench_attention.py | sed 's/libfp_amd_profile.so/QT1/'
  FP_AMD_LIB=$P FP_ATT_QT=2 timeout 100 python scripts/bench_attention.py | sed 's/libfp_amd_profile.so/QT2/'
done 2>&1 | tee gpurun_out/r3f_attention_ab.log | cut -c1-200
for rep in 1 2; do
  FP_AMD_LIB=$P FP_ATT_QT=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('QT1', d['ms_per_step'])"
  FP_AMD_LIB=$P FP_ATT_QT=2 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('QT2', d['ms_per_step'])"
done
echo "total seconds: $(( $(date +%s) - T0 ))"
