export TMPDIR=/tmp
mkdir -p gpurun_out
T=r02i
if ! timeout 200 python -c "import torch; assert torch.ones(8,device='cuda').sum().item()==8" >/dev/null 2>&1; then echo "BOX BROKEN"; exit 7; fi
timeout 100 python scripts/dbg_raster.py matmul f16 1 rasterfirst 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
timeout 300 python scripts/dbg_streams.py 75 3 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python scripts/dbg_streams.py 252 2 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python scripts/dbg_streams2.py 75 render,conv1,encoder,heads,ln,colmean,attn 40 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-200
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_amp.py -q -x -k "not 252" > gpurun_out/${T}_kernels.log 2>&1; tail -4 gpurun_out/${T}_kernels.log | cut -c1-300
echo "== parity tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/${T}_parity.log 2>&1; tail -4 gpurun_out/${T}_parity.log | cut -c1-300
echo "== per-layer igemm"; timeout 200 python scripts/bench_igemm.py > gpurun_out/${T}_igemm.log 2>&1; grep -E "c2|stem|joint|j2|linear" gpurun_out/${T}_igemm.log | grep -v "false" | cut -c1-200
for S in 1 2; do
  echo "== bench --streams $S"; timeout 300 python bench.py --streams $S --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_bench_s$S.json 2> gpurun_out/${T}_bench_s$S.err; cut -c100-215 gpurun_out/${T}_bench_s$S.json
done
echo "== bench (2 streams, kernel table)"; timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r02i_bench.json'))
print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['launches_timed'])
for k, v in d['kernels'].items(): print(k, v['calls'], v['avg_ms'])
PY
