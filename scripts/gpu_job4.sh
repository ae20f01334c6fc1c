export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "conv7x7 or hip_encoder or warp or fp16" > gpurun_out/pytest_gpu4.log 2>&1; tail -25 gpurun_out/pytest_gpu4.log
timeout 300 python scripts/bench_igemm.py 2>&1 | tail -8
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench4.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(json.dumps(d['kernels'], indent=0)[:1500])"
