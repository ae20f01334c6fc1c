# round 3, call I: (1) bench under torch.distributed on one rank (both modes), (2) side-stream priority experiment, (3) per-layer GEMM table at HEAD,
# (4) FETCH_SIZE / WRITE_SIZE passes over scripts/run_kernels.py for profiles/traffic.json
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 120 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
FP_BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3i_bench_rccl_world1.json 2> gpurun_out/r3i_rccl.err; tail -1 gpurun_out/r3i_rccl.err; cut -c1-200 gpurun_out/r3i_bench_rccl_world1.json
FP_BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode hypothesis > gpurun_out/r3i_bench_hypothesis_world1.json 2> gpurun_out/r3i_hyp.err; tail -1 gpurun_out/r3i_hyp.err; cut -c1-200 gpurun_out/r3i_bench_hypothesis_world1.json
for rep in 1 2; do for pr in 0 -1 1; do
  FP_AMD_SIDE_PRIORITY=$pr timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side priority $pr', d['ms_per_step'], d['concurrency']['sub_batches'], d['clock'].get('sclk_MHz_mean'), d['clock'].get('power_W_mean'), d['clock'].get('source_matches_torch_device_pci'))" || echo "priority $pr failed"
done; done
for n in 252 126; do FP_N=$n timeout 120 python scripts/bench_igemm.py; done > gpurun_out/r3i_igemm_layers.log 2>&1; grep -c TFLOPs gpurun_out/r3i_igemm_layers.log
timeout 220 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3i_pmc_fetch -o k -- python scripts/run_kernels.py > /dev/null 2>&1
timeout 220 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3i_pmc_write -o k -- python scripts/run_kernels.py > /dev/null 2>&1
python scripts/pmc_traffic.py gpurun_out/r3i_pmc_fetch/k_counter_collection.csv gpurun_out/r3i_pmc_write/k_counter_collection.csv gpurun_out/r3i_traffic.json | tail -15
echo "total seconds: $(( $(date +%s) - T0 ))"
