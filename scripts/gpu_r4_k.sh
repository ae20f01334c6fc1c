export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
P=foundationpose_amd/csrc/libfp_amd_profile.so
for d in 0 1 2 3 4 8 16 24 28 31; do echo "FP_ATT_DBG=$d $(FP_AMD_LIB=$P FP_ATT_DBG=$d timeout 100 python scripts/bench_attention.py 2>/dev/null | head -1 | cut -c1-140)"; done > gpurun_out/r4k_attention_isolation.log; cat gpurun_out/r4k_attention_isolation.log
