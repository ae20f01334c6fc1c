# round 2, closing validation: full GPU test suite, smoke, bench (+cpu baseline), rocprofv3 kernel stats of the bench command,
# PMC traffic, config 5, multi-GPU launch behaviour on a 1-GPU box.  T = evidence prefix.
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r02_final}
if ! timeout 200 python -c "import torch; assert torch.ones(8,device='cuda').sum().item()==8" >/dev/null 2>&1; then echo "BOX BROKEN"; exit 7; fi
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${T}_pytest_gpu.log | cut -c1-300
cp gpurun_out/parity_amp.json gpurun_out/${T}_parity_amp.json 2>/dev/null
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400
echo "== bench"; timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -1 gpurun_out/${T}_bench.err; cut -c1-260 gpurun_out/${T}_bench.json
echo "== bench --streams 1"; timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/${T}_bench_streams1.json 2> /dev/null; cut -c100-260 gpurun_out/${T}_bench_streams1.json
echo "== rocprof kernel stats (bench, default = 2 sub-batch streams)"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_prof.log 2>&1
head -12 gpurun_out/${T}_prof/bench_kernel_stats.csv | cut -c1-170
echo "== rocprof kernel stats (bench --serialize: the sub-batch launches on one stream throughout)"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_profs -o bench -- python bench.py --serialize --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_serialize.json 2> gpurun_out/${T}_profs.log
head -6 gpurun_out/${T}_profs/bench_kernel_stats.csv | cut -c1-170
echo "== rocprof kernel stats (bench --streams 1)"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof1 -o bench -- python bench.py --streams 1 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_prof1.log 2>&1
head -8 gpurun_out/${T}_prof1/bench_kernel_stats.csv | cut -c1-170
echo "== per-layer igemm"; timeout 200 python scripts/bench_igemm.py > gpurun_out/${T}_igemm_layers.log 2>&1; grep -c TFLOPs gpurun_out/${T}_igemm_layers.log
python scripts/bench_conv1.py 2>/dev/null > gpurun_out/${T}_conv1.log; cat gpurun_out/${T}_conv1.log
echo "== PMC traffic"; timeout 220 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}_pmc_fetch -o k -- python scripts/run_kernels.py > /dev/null 2>&1
timeout 220 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}_pmc_write -o k -- python scripts/run_kernels.py > /dev/null 2>&1
python scripts/pmc_traffic.py gpurun_out/${T}_pmc_fetch/k_counter_collection.csv gpurun_out/${T}_pmc_write/k_counter_collection.csv gpurun_out/${T}_traffic.json | tr -d '\n' | cut -c1-1200; echo
echo "== config 5"; timeout 400 python scripts/bench_track.py > gpurun_out/${T}_track_config5.json 2> gpurun_out/${T}_track.err; tail -1 gpurun_out/${T}_track.err; cut -c380-1100 gpurun_out/${T}_track_config5.json
echo "== multi-GPU launch on a 1-GPU box"; python bench.py --gpus 2 > gpurun_out/${T}_bench_gpus2_on_1gpu_box.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/${T}_bench_gpus2_on_1gpu_box.log
FP_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_bench_rccl_world1.json 2> /dev/null; echo "rccl rc=$?"; cut -c100-250 gpurun_out/${T}_bench_rccl_world1.json
FP_BENCH_FORCE_DIST=1 timeout 300 python bench.py --mode hypothesis --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_bench_hypothesis_mode_world1.json 2> /dev/null; echo "hyp rc=$?"; cut -c100-250 gpurun_out/${T}_bench_hypothesis_mode_world1.json
