# round 2, GPU call D: low-priority side stream A/B, graph tests, kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r02d
echo "== sanity (plain torch on the GPU)"; timeout 200 python -c "import torch; x=torch.ones(1<<20,device='cuda'); y=(x*2).sum().item(); print('torch ok', y)" 2>&1 | tail -2
if ! timeout 100 python -c "import torch; assert torch.ones(8,device='cuda').sum().item()==8" >/dev/null 2>&1; then echo "BOX BROKEN: plain torch fails"; exit 7; fi
echo "== kernels"; timeout 600 python -m pytest tests/test_gpu_amp.py -q -k "not 252" > gpurun_out/${T}_kernels.log 2>&1; tail -6 gpurun_out/${T}_kernels.log
echo "== graph / raster tests"; timeout 400 python -m pytest tests/test_gpu_parity.py -q -k "render or raster or warp or golden or pose_update or graph or tracker" > gpurun_out/${T}_raster.log 2>&1; tail -4 gpurun_out/${T}_raster.log | cut -c1-300
echo "== per-layer igemm (split, low-priority side stream)"; timeout 200 python scripts/bench_igemm.py > gpurun_out/${T}_igemm.log 2>&1; grep -E "stem|joint|linear" gpurun_out/${T}_igemm.log | grep -v "false"
echo "== per-layer igemm (profile lib, nosplit)"; FP_AMD_LIB=$PWD/foundationpose_amd/csrc/libfp_amd_profile.so FP_IGEMM_TILE=nosplit timeout 200 python scripts/bench_igemm.py > gpurun_out/${T}_igemm_nosplit.log 2>&1; grep -E "stem|joint|linear" gpurun_out/${T}_igemm_nosplit.log | grep -v "false"
echo "== bench (split)"; timeout 500 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; cut -c100-260 gpurun_out/${T}_bench.json
echo "== bench (nosplit, profile lib)"; FP_AMD_LIB=$PWD/foundationpose_amd/csrc/libfp_amd_profile.so FP_IGEMM_TILE=nosplit timeout 500 python bench.py --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_bench_nosplit.json 2> gpurun_out/${T}_bench_nosplit.err; cut -c100-260 gpurun_out/${T}_bench_nosplit.json
echo "== trace (split)"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_prof.log 2>&1; ls gpurun_out/${T}_prof | head -3
