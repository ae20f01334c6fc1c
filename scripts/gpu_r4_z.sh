# round 4, call Z: the row-owning fused kernels after the operand-path rewrite (weights fragment-packed from L2 into registers, A tile
# resident in LDS): equality tests, phase timers, micro-bench, then the whole bench
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "linear_layernorm or ffn_layernorm or sub_batches or graphed_predict" 2>&1 | tail -5
FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so timeout 200 python scripts/dbg_linear_ln.py 2>&1 | grep "N=" | tee gpurun_out/r4z_linear_ln_phases.log
timeout 200 python scripts/bench_linear_ln.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4z_bench_linear_ln.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err; python scripts/show_bench_kernels.py gpurun_out/r4z_bench.json | head -30
FP_AMD_FUSED_LN=0 FP_AMD_FUSED_FFN=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table > gpurun_out/r4z_bench_unfused.json 2>/dev/null; cut -c1-200 gpurun_out/r4z_bench_unfused.json
