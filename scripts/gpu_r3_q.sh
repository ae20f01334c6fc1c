# round 3, call Q (closing): GPU suite at HEAD without its three slowest oracle comparisons (they ran in call O, one commit before the
# replication kernel; nothing they cover changed), smoke, and the rocprofv3 kernel statistics of the default bench command
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 120 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); y = (x * 2).sum().item(); assert y == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE: giving up on this box"; exit 7; }
timeout 400 python -m pytest tests -m gpu -q --timeout 200 --durations=4 --deselect tests/test_gpu_amp.py::test_refiner_252_teacher_forced_three_way --deselect tests/test_gpu_amp.py::test_refiner_252_free_running_chain --deselect tests/test_gpu_parity.py::test_refiner_fp32_matches_oracle > gpurun_out/r3q_pytest_gpu_fast.log 2>&1; tail -9 gpurun_out/r3q_pytest_gpu_fast.log | cut -c1-250
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3q_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3q_bench_profiled.json 2> /dev/null; cd $GRAFT_REPO_ROOT
cut -c1-330 gpurun_out/r3q_bench_profiled.json; head -8 gpurun_out/r3q_prof/bench_kernel_stats.csv | cut -c1-150
rm -f gpurun_out/r3q_prof/bench_kernel_trace.csv gpurun_out/r3q_prof/*agent_info.csv
echo "total seconds: $(( $(date +%s) - T0 ))"
