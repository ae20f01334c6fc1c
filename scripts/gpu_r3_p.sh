# round 3, call P: the replication kernel of the shared observed crop -- its tests, the bench without / with --shared-crop, kernel stats
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 120 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); y = (x * 2).sum().item(); assert y == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE: giving up on this box"; exit 7; }
timeout 300 python -m pytest tests -m gpu -q --timeout 200 -k "replicate or shared_observed or sub_batches or estimator_api or small_batches" > gpurun_out/r3p_pytest.log 2>&1; tail -4 gpurun_out/r3p_pytest.log | cut -c1-250
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r3p_bench_plain.json 2> /dev/null; cut -c1-330 gpurun_out/r3p_bench_plain.json
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --shared-crop > gpurun_out/r3p_bench_shared_crop.json 2> /dev/null; cut -c1-330 gpurun_out/r3p_bench_shared_crop.json
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3p_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table --shared-crop > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
head -14 gpurun_out/r3p_prof/bench_kernel_stats.csv | cut -c1-150
rm -f gpurun_out/r3p_prof/bench_kernel_trace.csv gpurun_out/r3p_prof/*agent_info.csv
echo "total seconds: $(( $(date +%s) - T0 ))"
