# round 3, call N: A/B of the rasteriser variants, then validation at HEAD (full GPU suite, smoke, bench)
export TMPDIR=/tmp
mkdir -p gpurun_out
T0=$(date +%s)
timeout 120 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); y = (x * 2).sum().item(); assert y == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE: giving up on this box"; exit 7; }
timeout 420 python scripts/raster_variants.py > gpurun_out/r3n_raster_variants.log 2>&1; cut -c1-260 gpurun_out/r3n_raster_variants.log
echo "variants seconds: $(( $(date +%s) - T0 ))"
timeout 1150 python -m pytest tests -m gpu -q --timeout 420 --durations=6 > gpurun_out/r3n_pytest_gpu.log 2>&1; tail -14 gpurun_out/r3n_pytest_gpu.log | cut -c1-250
echo "pytest seconds: $(( $(date +%s) - T0 ))"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py > gpurun_out/r3n_bench.json 2> gpurun_out/r3n_bench.err; tail -1 gpurun_out/r3n_bench.err; cut -c1-330 gpurun_out/r3n_bench.json
python scripts/show_bench_kernels.py gpurun_out/r3n_bench.json
echo "total seconds: $(( $(date +%s) - T0 ))"
