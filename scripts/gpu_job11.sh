export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "render or golden or refiner or scorer or estimator or nvdiffrast" 2>&1 | tail -12
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench11.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:(v['calls'],round(v['avg_ms'],4)) for k,v in d['kernels'].items()}); print(d['roofline'])"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof11 -o k -- python scripts/run_kernels.py > /dev/null 2>&1; grep -E "k_vertex|k_bin|k_raster|k_warp" gpurun_out/prof11/k_kernel_stats.csv | cut -c1-160
