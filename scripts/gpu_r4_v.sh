export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "ffn_layernorm or sub_batches or graphed_predict or eight_hypothesis or shared_observed" 2>&1 | tail -5 | cut -c1-300
FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so timeout 120 python scripts/bench_linear_ln_mean.py 2>&1 | grep -v amdgpu | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_amp.py -m gpu -q --timeout 600 -k "contraction or plans_match" 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | cut -c150-230; done
