export TMPDIR=/tmp
mkdir -p gpurun_out
T=r02h
if ! timeout 200 python -c "import torch; assert torch.ones(8,device='cuda').sum().item()==8" >/dev/null 2>&1; then echo "BOX BROKEN"; exit 7; fi
timeout 300 python scripts/dbg_streams.py 75 3 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python scripts/dbg_streams.py 252 2 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python scripts/dbg_streams.py 64 2 2>&1 | grep -v amdgpu.ids | tail -3
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sub_batches or graph or tracker or estimator or small_batches or scorer" > gpurun_out/${T}_tests.log 2>&1; tail -5 gpurun_out/${T}_tests.log | cut -c1-400
echo "== config 5"; timeout 400 python scripts/bench_track.py > gpurun_out/${T}_track.json 2> gpurun_out/${T}_track.err; cat gpurun_out/${T}_track.json | cut -c400-1100
