export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
{ timeout 100 python scripts/bench_warp.py; FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so FP_WARP_V=1 timeout 100 python scripts/bench_warp.py; FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so timeout 100 python scripts/bench_warp.py; } 2>&1 | grep "^WARP" > gpurun_out/r4w_warp_ab.log; cat gpurun_out/r4w_warp_ab.log | cut -c1-700
python - <<'PY'
import json
rows = [json.loads(l[5:]) for l in open("gpurun_out/r4w_warp_ab.log")]
ref = rows[1]
for r in rows:
    diff = [k for k in r if k not in ("lib",) and not k.startswith("us_") and r[k] != ref[k]]
    print(r["lib"], "us", r.get("us_refine"), r.get("us_score"), "differs from k_warp on:", diff or "nothing")
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -x -k "warp or golden or estimator or graphed_predict" 2>&1 | tail -3 | cut -c1-300
