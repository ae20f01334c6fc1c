# round 2, GPU call C: split launches + rewritten rasteriser: kernel tests, raster/parity tests, per-layer A/B, bench
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r02c
echo "== kernels"; timeout 600 python -m pytest tests/test_gpu_amp.py -q -k "not 252" > gpurun_out/${T}_kernels.log 2>&1; tail -12 gpurun_out/${T}_kernels.log
echo "== raster + image-space parity"; timeout 400 python -m pytest tests/test_gpu_parity.py -q -k "render or raster or warp or golden or pose_update or graph or tracker" > gpurun_out/${T}_raster.log 2>&1; tail -8 gpurun_out/${T}_raster.log
echo "== per-layer igemm (split)"; timeout 200 python scripts/bench_igemm.py > gpurun_out/${T}_igemm.log 2>&1; tail -14 gpurun_out/${T}_igemm.log
echo "== per-layer igemm (profile lib, nosplit)"; FP_AMD_LIB=$PWD/foundationpose_amd/csrc/libfp_amd_profile.so FP_IGEMM_TILE=nosplit timeout 200 python scripts/bench_igemm.py > gpurun_out/${T}_igemm_nosplit.log 2>&1; tail -14 gpurun_out/${T}_igemm_nosplit.log
echo "== bench"; timeout 500 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -2 gpurun_out/${T}_bench.err; cut -c1-300 gpurun_out/${T}_bench.json
echo "== rocprof kernel stats"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-table > gpurun_out/${T}_prof.log 2>&1
head -16 gpurun_out/${T}_prof/bench_kernel_stats.csv | cut -c1-170
