export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "render or golden or nvdiffrast" 2>&1 | tail -4
python scripts/raster_phases.py 2>&1 | tail -4
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof12 -o k -- python scripts/run_kernels.py > /dev/null 2>&1; grep -E "k_vertex|k_bin|k_raster" gpurun_out/prof12/k_kernel_stats.csv | cut -c1-120
