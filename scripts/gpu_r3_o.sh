# round 3, call O: validation at HEAD (full GPU suite, smoke, bench) + the bench with the observed crop of iteration 0 shared
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 120 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); y = (x * 2).sum().item(); assert y == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE: giving up on this box"; exit 7; }
timeout 1150 python -m pytest tests -m gpu -q --timeout 420 --durations=6 > gpurun_out/r3o_pytest_gpu.log 2>&1; tail -14 gpurun_out/r3o_pytest_gpu.log | cut -c1-250
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py > gpurun_out/r3o_bench.json 2> gpurun_out/r3o_bench.err; tail -1 gpurun_out/r3o_bench.err; cut -c1-330 gpurun_out/r3o_bench.json
python scripts/show_bench_kernels.py gpurun_out/r3o_bench.json
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > gpurun_out/r3o_bench_plain.json 2> /dev/null; cut -c1-330 gpurun_out/r3o_bench_plain.json
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --shared-crop > gpurun_out/r3o_bench_shared_crop.json 2> /dev/null; cut -c1-330 gpurun_out/r3o_bench_shared_crop.json
echo "total seconds: $(( $(date +%s) - T0 ))"
