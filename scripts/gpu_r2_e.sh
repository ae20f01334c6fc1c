# round 2, GPU call E: hypothesis sub-batches on concurrent streams (overlap.py): equality tests, bench A/B, config 5
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r02e
if ! timeout 200 python -c "import torch; assert torch.ones(8,device='cuda').sum().item()==8" >/dev/null 2>&1; then echo "BOX BROKEN: plain torch fails"; exit 7; fi
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sub_batches or graph or tracker or estimator or small_batches or scorer" > gpurun_out/${T}_tests.log 2>&1; tail -5 gpurun_out/${T}_tests.log | cut -c1-400
for S in 1 2 3 4; do
  echo "== bench --streams $S"; timeout 300 python bench.py --streams $S --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_bench_s$S.json 2> gpurun_out/${T}_bench_s$S.err; cut -c100-215 gpurun_out/${T}_bench_s$S.json
done
echo "== bench (2 streams, kernel table)"; timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r02e_bench.json'))
print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['launches_timed'])
for k, v in d['kernels'].items(): print(k, v['calls'], v['avg_ms'])
PY
echo "== config 5"; timeout 400 python scripts/bench_track.py > gpurun_out/${T}_track.json 2> gpurun_out/${T}_track.err; cat gpurun_out/${T}_track.json | cut -c1-900
echo "== trace (2 streams)"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_prof.log 2>&1; ls gpurun_out/${T}_prof | head -3
