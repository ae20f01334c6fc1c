# round 4, call Y: the vendor GEMM at the layers' GEMM-equivalent shapes beside fp_igemm_f16_fwd on the same box; kernel names via rocprofv3
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
for n in 252 126; do
  FP_N=$n timeout 300 python scripts/bench_lib_gemm.py 2>&1 | tee -a gpurun_out/r4y_lib_gemm.log
  FP_N=$n timeout 300 python scripts/bench_igemm.py 2>&1 | tee -a gpurun_out/r4y_lib_gemm.log
done
FP_N=126 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4y_prof -o g -- python scripts/bench_lib_gemm.py > /dev/null 2>&1
python - <<'PY' | tee -a gpurun_out/r4y_lib_gemm.log
import csv, glob
f = glob.glob("gpurun_out/r4y_prof/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print(r["Name"][:150], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/r4y_prof
