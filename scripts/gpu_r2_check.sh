export TMPDIR=/tmp
mkdir -p gpurun_out
if ! timeout 200 python -c "import torch; assert torch.ones(8,device='cuda').sum().item()==8" >/dev/null 2>&1; then echo "BOX BROKEN"; exit 7; fi
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_h_pytest_gpu.log 2>&1; tail -3 gpurun_out/r02_h_pytest_gpu.log | cut -c1-300
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== bench"; timeout 600 python bench.py > gpurun_out/r02_h_bench.json 2> gpurun_out/r02_h_bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r02_h_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('full_batch_launches', {}).get('frac'), d['network_mfma']['achieved_TFLOPs'], d['cpu_baseline']['value'])
PY
