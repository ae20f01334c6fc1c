export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
P=foundationpose_amd/csrc/libfp_amd_profile.so
{ timeout 100 python scripts/bench_attention.py; FP_AMD_LIB=$P FP_ATT_WAVES=4 timeout 100 python scripts/bench_attention.py; FP_AMD_LIB=$P FP_ATT_PP=1 timeout 100 python scripts/bench_attention.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r4j_attention_ab.log; cat gpurun_out/r4j_attention_ab.log | cut -c1-220
