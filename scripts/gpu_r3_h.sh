# round 3, call H: validation at HEAD -- full GPU suite, smoke, bench (default, incl. cpu baseline), marker trace of the default command
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
# a box whose GPU faults on a plain torch op (seen once this round: "Memory access fault by GPU node-2" in torch.as_tensor) is not worth the minutes
timeout 120 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); y = (x * 2).sum().item(); assert y == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE: giving up on this box"; exit 7; }
timeout 1150 python -m pytest tests -m gpu -q --timeout 420 --durations=6 > gpurun_out/r3h_pytest_gpu.log 2>&1; tail -14 gpurun_out/r3h_pytest_gpu.log | cut -c1-250
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err; tail -1 gpurun_out/r3h_bench.err; cut -c1-330 gpurun_out/r3h_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3h_prof -o bench -- python bench.py --steps 5 --warmup 3 --trace-markers --no-kernel-table --no-cpu-baseline > gpurun_out/r3h_bench_traced.json 2> /dev/null
python scripts/concurrent_roofline.py gpurun_out/r3h_prof/bench_kernel_trace.csv gpurun_out/r3h_bench_traced.json > gpurun_out/r3h_concurrent_roofline.json; grep -E "region_ms_per_step|idle_frac|achieved|frac\"|busy_share" gpurun_out/r3h_concurrent_roofline.json
rm -f gpurun_out/r3h_prof/bench_kernel_trace.csv gpurun_out/r3h_prof/*agent_info.csv
echo "total seconds: $(( $(date +%s) - T0 ))"
