export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench18.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:(v['calls'],round(v['avg_ms'],4)) for k,v in d['kernels'].items()}); print(d['roofline']); print(d['network_mfma'])"
