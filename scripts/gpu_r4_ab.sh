# round 4, call AB: k_conv_sw with a prefetch of the next tile's first patch chunk into L2 (libfp_amd_alt.so built with -DSW_PREFETCH_NEXT=1)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
for rep in 1 2; do
for lib in libfp_amd.so libfp_amd_alt.so; do
  FP_N=126 FP_AMD_LIB=foundationpose_amd/csrc/$lib timeout 200 python scripts/bench_igemm.py 2>&1 | grep -E "stem 128|joint 256->256|joint 512|weighted" | tee -a gpurun_out/r4ab_prefetch.log
done; done
FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_alt.so timeout 300 python -m pytest tests/test_gpu_amp.py -m gpu -q -x -k "policy or plans_match" 2>&1 | tail -3
for lib in libfp_amd.so libfp_amd_alt.so libfp_amd.so libfp_amd_alt.so; do
  FP_AMD_LIB=foundationpose_amd/csrc/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib', d['ms_per_step'], d['clock']['sclk_MHz_mean'])" | tee -a gpurun_out/r4ab_prefetch.log
done
