# round 3, call E: attention A/B, the full GPU suite, smoke, the bench lines and the kernel statistics that go to profiles/
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
for rep in 1 2; do
  timeout 100 python scripts/bench_attention.py
  FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_prevatt.so timeout 100 python scripts/bench_attention.py
done 2>&1 | tee gpurun_out/r3e_attention_ab.log | cut -c1-200
echo "attention seconds: $(( $(date +%s) - T0 ))"
timeout 1150 python -m pytest tests -m gpu -q --timeout 420 --durations=10 > gpurun_out/r3e_pytest_gpu.log 2>&1; tail -22 gpurun_out/r3e_pytest_gpu.log | cut -c1-250
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err; tail -2 gpurun_out/r3e_bench.err; cut -c1-400 gpurun_out/r3e_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3e_prof_ser -o bench -- python bench.py --steps 5 --warmup 2 --serialize --no-kernel-table --no-cpu-baseline > gpurun_out/r3e_bench_serialize.json 2> /dev/null
head -14 gpurun_out/r3e_prof_ser/bench_kernel_stats.csv | cut -c1-170
rm -f gpurun_out/r3e_prof_ser/bench_kernel_trace.csv gpurun_out/r3e_prof_ser/*agent_info.csv
timeout 200 python bench.py --precision torch_amp --steps 2 --warmup 1 --no-kernel-table --no-cpu-baseline > gpurun_out/r3e_bench_torch_amp.json 2> /dev/null; cut -c1-330 gpurun_out/r3e_bench_torch_amp.json
timeout 200 python scripts/bench_track.py > gpurun_out/r3e_track_config5.json 2> gpurun_out/r3e_track.err; tail -c 600 gpurun_out/r3e_track_config5.json
echo "total seconds: $(( $(date +%s) - T0 ))"
