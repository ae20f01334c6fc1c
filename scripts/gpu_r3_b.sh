# round 3, call B: A/B of the lock-step two-workgroups-per-CU shifted-window kernel (libfp_amd_alt.so) against the ping-pong one
export TMPDIR=/tmp
mkdir -p gpurun_out
ALT=foundationpose_amd/csrc/libfp_amd_alt.so
T0=$(date +%s)
FP_AMD_LIB=$ALT timeout 900 python -m pytest tests/test_gpu_amp.py -m gpu -q -k "igemm_conv3x3_policy or encoder_matches or plans_match or channel_concat" > gpurun_out/r3b_pytest_alt.log 2>&1; tail -5 gpurun_out/r3b_pytest_alt.log
echo "pytest seconds: $(( $(date +%s) - T0 ))"
: > gpurun_out/r3b_igemm_ab.log
for rep in 1 2; do for n in 252 126; do
  FP_N=$n timeout 120 python scripts/bench_igemm.py >> gpurun_out/r3b_igemm_ab.log 2>&1
  FP_N=$n FP_AMD_LIB=$ALT timeout 120 python scripts/bench_igemm.py >> gpurun_out/r3b_igemm_ab.log 2>&1
  FP_N=$n FP_AMD_LIB=foundationpose_amd/csrc/libfp_amd_profile.so FP_IGEMM_TILE=ls256x128 timeout 120 python scripts/bench_igemm.py >> gpurun_out/r3b_igemm_ab.log 2>&1
done; done
python - <<'PY'
import json,collections
rows=collections.defaultdict(dict)
for l in open('gpurun_out/r3b_igemm_ab.log'):
    try: d=json.loads(l)
    except Exception: continue
    key=(d['layer'].split(' M=')[0] if 'linear' in d['layer'] else d['layer'], d['layer'].split('N=')[-1] if 'linear' in d['layer'] else '', d.get('residual'), d['lib'].split('/')[-1])
    rows[key].setdefault('/'.join(d['lib'].split('/')[:2]), []).append(d['TFLOPs'])
for k in sorted(rows, key=str): print(k, {a: v for a, v in rows[k].items()})
PY
for rep in 1 2; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pp', d['ms_per_step'], d['clock'])"
  FP_AMD_LIB=$ALT timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ls', d['ms_per_step'], d['clock'])"
done
echo "total seconds: $(( $(date +%s) - T0 ))"
