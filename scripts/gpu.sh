#!/bin/bash
# ONE parameterised GPU-box script (replaces the fifty single-use scripts/gpu_r*_*.sh of rounds 2-4):
#   gpurun --timeout 1500 -- 'TAG=r05_a bash scripts/gpu.sh sane tests bench prof pmc track smoke'
#   (round 6: + trace, abengine, trained, render, convprobe; the round's closing records = sane tests smoke bench prof pmc trace)
# Every task writes gpurun_out/${TAG}_<task>.*; summaries worth keeping are copied into profiles/ by hand afterwards.
export TMPDIR=/tmp
TAG=${TAG:-r05}
O=gpurun_out
mkdir -p $O
T0=$(date +%s)
el() { echo "== $1 (t=$(( $(date +%s) - T0 )) s)"; }
CS=foundationpose_amd/csrc
BENCH_FAST="--no-cpu-baseline --no-kernel-table --no-extras"

t_sane() {
  timeout 200 python -c "import torch; x = torch.ones(1 << 22, device='cuda'); assert (x * 2).sum().item() == 2 * (1 << 22); print('gpu sane')" || { echo "GPU NOT SANE"; exit 7; }
}
t_tests() {   # the whole GPU suite
  timeout 1500 python -m pytest tests/ -m gpu -q --timeout 900 --durations=8 > $O/${TAG}_pytest_gpu.log 2>&1; tail -16 $O/${TAG}_pytest_gpu.log | cut -c1-300
}
t_tests_fast() {   # everything but the three long parity chains
  timeout 900 python -m pytest tests/ -m gpu -q --timeout 600 -x -k "not teacher_forced and not fp32_matches_oracle and not scorer_252_vs_exact" > $O/${TAG}_pytest_gpu_fast.log 2>&1; tail -8 $O/${TAG}_pytest_gpu_fast.log | cut -c1-300
}
t_bench() {   # the driver's command
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; python scripts/show_bench_kernels.py $O/${TAG}_bench.json
}
bench_line() {  # name, env/lib, extra flags: one short line per variant (A/B runs on the same box)
  local name=$1 lib=$2; shift 2
  FP_AMD_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 $BENCH_FAST "$@" > $O/${TAG}_bench_$name.json 2> $O/${TAG}_bench_$name.err
  python - $O/${TAG}_bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("clock", {})
    print(f"   {sys.argv[2]:28s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} hyp/s  sclk {c.get('sclk_MHz_mean') or 0:.0f} MHz  {c.get('power_W_mean') or 0:.0f} W")
except Exception as e:
    print("   ", sys.argv[2], "FAILED", e)
PY
}
t_ab() {      # same box, back to back: product, lock-step two-per-CU conv (make ab), 3 sub-batch streams; product again
  bench_line product $PWD/$CS/libfp_amd.so
  [ -f $CS/libfp_amd_alt.so ] && bench_line conv_sw_ls $PWD/$CS/libfp_amd_alt.so
  bench_line streams3 $PWD/$CS/libfp_amd.so --streams 3
  bench_line product_again $PWD/$CS/libfp_amd.so
}
t_libs() {    # LIBS="product dmac1 ...": per build the output hashes of the 3x3 convs, per-layer rates at N = 126 / 252, the step
  for name in $LIBS; do
    local lib=$PWD/$CS/libfp_amd_$name.so; [ $name = product ] && lib=$PWD/$CS/libfp_amd.so
    echo "-- $name"
    FP_AMD_LIB=$lib timeout 120 python scripts/cmp_conv_sw.py 2> /dev/null | sha1sum | cut -c1-16 | sed 's/^/   outputs sha1 /'
    for n in 126 252; do
      FP_N=$n FP_AMD_LIB=$lib timeout 120 python scripts/bench_igemm.py 2> /dev/null > $O/${TAG}_igemm_${name}_$n.log
      python - $O/${TAG}_igemm_${name}_$n.log $n <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
pick = lambda name, res: next((r["TFLOPs"] for r in rows if r["layer"] == name and r.get("residual", res) == res), 0)
print(f"   N={sys.argv[2]}: 128->128 {pick('stem 128->128', False):.0f}/{pick('stem 128->128', True):.0f}  256->256 {pick('joint 256->256', False):.0f}/{pick('joint 256->256', True):.0f}  "
      f"512->512 {pick('joint 512->512', False):.0f}/{pick('joint 512->512', True):.0f}  weighted {rows[-3]['TFLOPs'] if len(rows) > 3 else 0:.0f} TFLOP/s (no residual / residual)")
PY
    done
    bench_line $name $lib
  done
}
t_wtiles() {  # tile-packed conv weights on / off with ONE library: output digests, per-layer rates, the step
  for v in 0 1 0 1; do
    echo "-- w_tiles=$v"
    FP_W_TILES=$v timeout 120 python scripts/cmp_conv_sw.py 2> /dev/null | sha1sum | cut -c1-16 | sed 's/^/   outputs sha1 /'
    for n in 126 252; do
      FP_W_TILES=$v FP_N=$n timeout 120 python scripts/bench_igemm.py 2> /dev/null > $O/${TAG}_igemm_wt${v}_$n.log
      python - $O/${TAG}_igemm_wt${v}_$n.log $n <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
pick = lambda name, res: next((r["TFLOPs"] for r in rows if r["layer"] == name and r.get("residual", res) == res), 0)
print(f"   N={sys.argv[2]}: 128->128 {pick('stem 128->128', False):.0f}/{pick('stem 128->128', True):.0f}  256->256 {pick('joint 256->256', False):.0f}/{pick('joint 256->256', True):.0f}  "
      f"512->512 {pick('joint 512->512', False):.0f}/{pick('joint 512->512', True):.0f}  weighted {rows[-3]['TFLOPs'] if len(rows) > 3 else 0:.0f} TFLOP/s (no residual / residual)")
PY
    done
    FP_BENCH_ENGINE="PACKED_CONV_TILES=$v" bench_line wtiles$v $PWD/$CS/libfp_amd.so
  done
}
t_probe() {   # MFMA / VALU co-issue probe (scripts/mfma_valu_overlap)
  hipcc --offload-arch=gfx950 -O3 -pthread -o /tmp/mvprobe scripts/mfma_valu_overlap/probe.hip && timeout 300 /tmp/mvprobe > $O/${TAG}_mfma_valu_probe.log 2>&1; cat $O/${TAG}_mfma_valu_probe.log
}
t_amp_fast() {   # the kernel-level policy tests (bit-identity / flip-rate gates of the GEMM family, plans against the oracle)
  timeout 600 python -m pytest tests/test_gpu_amp.py -m gpu -q --timeout 600 -x -k "policy or encoder or plans" 2>&1 | tail -4 | cut -c1-300
}
t_conv_phases() {
  FP_AMD_LIB=$PWD/$CS/libfp_amd_profile.so timeout 300 python scripts/dbg_conv_sw_step.py 2> /dev/null | tee $O/${TAG}_conv_sw_step_phases.log
  FP_AMD_LIB=$PWD/$CS/libfp_amd_profile.so timeout 300 python scripts/dbg_conv_sw.py 2> /dev/null | tee $O/${TAG}_conv_sw_phases.log
}
t_prof() {    # bench --serialize under rocprofv3 --stats
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o bench -- python bench.py --serialize --no-graph --steps 5 --warmup 2 $BENCH_FAST > $O/${TAG}_bench_serialize.json 2> /dev/null
  python - $O $TAG <<'PY'
import csv, glob, sys
O, TAG = sys.argv[1:3]
f = glob.glob(f"{O}/{TAG}_prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
w = csv.DictWriter(open(f"{O}/{TAG}_bench_serialize_kernel_stats.csv", "w"), fieldnames=list(rows[0].keys()))
w.writeheader()
for r in rows[:40]:
    r["Name"] = r["Name"][:90]
    w.writerow(r)
for r in rows[:12]:
    print("  ", r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
  rm -rf $O/${TAG}_prof
}
t_pmc() {     # PMC passes over one serialized step (separate passes: FETCH_SIZE, WRITE_SIZE, MFMA busy)
  local B1="python bench.py --serialize --no-graph --steps 1 --warmup 1 --trace-markers $BENCH_FAST"
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -o k -- $B1 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -o k -- $B1 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_pmc_mfma -o k -- $B1 > /dev/null 2>&1
  local F=$(ls $O/${TAG}_pmc_fetch/*counter_collection.csv | head -1) W=$(ls $O/${TAG}_pmc_write/*counter_collection.csv | head -1) M=$(ls $O/${TAG}_pmc_mfma/*counter_collection.csv | head -1)
  python scripts/pmc_step_traffic.py "$F" "$W" $O/${TAG}_step_traffic.json > /dev/null; python scripts/pmc_step_mfma.py "$M" $O/${TAG}_step_mfma_busy.json | head -8
  rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_mfma
}
t_track() {
  timeout 400 python scripts/bench_track.py > $O/${TAG}_track_config5.json 2> /dev/null; cut -c1-600 $O/${TAG}_track_config5.json
}
t_smoke() {
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-400
}
t_trace() {   # kernel trace of the default (two-stream, graph-replay) command -> scripts/concurrent_roofline.py
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace -o bench -- python bench.py --trace-markers --no-kernel-table --no-cpu-baseline --no-extras > $O/${TAG}_bench_traced.json 2> /dev/null
  local T=$(ls $O/${TAG}_trace/*kernel_trace.csv $O/${TAG}_trace/*/*kernel_trace.csv 2>/dev/null | head -1)
  python scripts/concurrent_roofline.py "$T" $O/${TAG}_bench_traced.json > $O/${TAG}_concurrent_roofline.json 2> /dev/null; python -c "import json; d = json.load(open('$O/${TAG}_concurrent_roofline.json')); print(d['roofline_concurrent'], d['idle_frac'])"
  rm -rf $O/${TAG}_trace
}
t_abengine() {   # ENGINE_AB="name:SWITCH=0,SWITCH=0 name2: ...": same-box A/B of engine.py switches (bench.py FP_BENCH_ENGINE)
  for v in $ENGINE_AB; do
    local name=${v%%:*} eng=${v#*:}
    FP_BENCH_ENGINE="$eng" bench_line $name $PWD/$CS/libfp_amd.so
  done
}
t_trained() {    # the trained stand-in against its exactly-rounded chain + the one-hypothesis small-call path
  FP_PARITY_REPORT=${TAG}_parity_trained.json timeout 1500 python -m pytest tests/test_gpu_amp.py -q --timeout 1200 -k "trained_standin or track_one_small_call" 2>&1 | tail -3 | cut -c1-300
}
t_render() {     # fp_render_crops alone (LIBS="product ..."): us per launch + output digests
  for name in ${LIBS:-product}; do
    local lib=$PWD/$CS/libfp_amd_$name.so; [ $name = product ] && lib=$PWD/$CS/libfp_amd.so
    echo "-- $name"; FP_AMD_LIB=$lib timeout 120 python scripts/bench_render.py 2>&1 | tail -4
  done
}
t_convprobe() {  # stand-alone replica of the conv main loop (scripts/conv_loop_probe): short (clocks per k-step), long with random operands (power cap)
  hipcc --offload-arch=gfx950 -O3 -o /tmp/cvprobe scripts/conv_loop_probe/probe.hip && { timeout 200 /tmp/cvprobe; timeout 300 /tmp/cvprobe 2000; timeout 300 /tmp/cvprobe 2000 random; } | tee $O/${TAG}_conv_loop_probe.log
}
for task in "$@"; do
  el "$task"
  if declare -f "t_$task" > /dev/null; then "t_$task"; else echo "unknown task $task"; fi
done
el done
