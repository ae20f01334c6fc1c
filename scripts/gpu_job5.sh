export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu5.log 2>&1; tail -15 gpurun_out/pytest_gpu5.log
timeout 300 python scripts/bench_igemm.py 2>&1 | tail -8
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench5.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:(v['calls'],round(v['avg_ms'],4)) for k,v in d['kernels'].items()})"
