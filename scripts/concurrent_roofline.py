"""Per-kernel busy time of bench.py's TIMED region, in the execution mode that is timed (hypothesis sub-batches on concurrent
streams), from a rocprofv3 kernel trace of the default command:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -o bench -- python bench.py --trace-markers --no-kernel-table --no-cpu-baseline > bench.json
    python scripts/concurrent_roofline.py gpurun_out/prof/bench_kernel_trace.csv bench.json > profiles/rNN_concurrent_roofline.json

The region is cut out by the two marker launches of `--trace-markers` (k_depth_to_xyz with a 64 x 4 grid).  For every kernel
family: number of launches, sum of launch durations, and BUSY time = length of the union of its launches' [start, end]
intervals (two launches that overlap on the chip count once).  `roofline_concurrent` = the network's GEMM arithmetic of the
region (bench.json: algorithmic TFLOP of the convolutions and projections per step x steps) / the busy time of the GEMM
family, next to the share of the region's wall time during which at least one GEMM kernel was running."""
import csv
import json
import re
import sys

FAMILIES = [("gemm", r"k_conv_sw|k_igemm|k_rows512|k_linear512|k_splitk"),      # round 4/5: the row-owning projection kernels and the split-K pair belong to the GEMM arithmetic
             ("attention", r"k_attention"), ("patch_embed", r"k_conv7x7"), ("rowops", r"k_layernorm|k_colmean|k_rows_linear|k_add_pe|k_ln_mean_finish"),
            ("raster", r"k_raster|k_bin|k_vertex"), ("warp", r"k_warp"), ("pose", r"k_crop_windows|k_pose_update")]
MFMA_PEAK = 2500.0
# GEMM-family arithmetic per hypothesis-pass (SURVEY 8(d), 2 x MAC): 15 3x3 convs + the 512-wide projections of the heads;
# the patch-embed conv and attention are their own families
ENC_3X3_GF, REFINE_PROJ_GF, SCORE_PROJ_GF = 20.772 - 2 * 0.2408, 2 * (0.629 + 3 * 0.2097), 0.629


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for a, b in iv:
        if cs is None:
            cs, ce = a, b
        elif a <= ce:
            ce = max(ce, b)
        else:
            tot += ce - cs
            cs, ce = a, b
    return tot + (ce - cs if cs is not None else 0)


def main(trace, bench_json):
    rows = list(csv.DictReader(open(trace)))
    marks = [r for r in rows if "k_depth_to_xyz" in r["Kernel_Name"] and r["Grid_Size_X"] == "64" and r["Grid_Size_Y"] == "4"]
    if len(marks) < 2:
        sys.exit("no marker pair in the trace: run bench.py with --trace-markers")
    t0, t1 = int(marks[-2]["End_Timestamp"]), int(marks[-1]["Start_Timestamp"])
    inside = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1]
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    steps, N = bench["steps"], bench["config"]["hypotheses_per_gpu"]
    R = bench["config"]["refine_iterations"]
    fam = {}
    for name, pat in FAMILIES + [("other", None)]:
        sel = [r for r in inside if (re.search(pat, r["Kernel_Name"]) if pat else not any(re.search(p, r["Kernel_Name"]) for _, p in FAMILIES))]
        iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel]
        fam[name] = dict(launches=len(sel), sum_ms=sum(b - a for a, b in iv) / 1e6, busy_ms=union(iv) / 1e6)
    wall = (t1 - t0) / 1e6
    all_busy = union([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in inside]) / 1e6
    gemm_tf = steps * N * (R * (ENC_3X3_GF + REFINE_PROJ_GF) + (ENC_3X3_GF + SCORE_PROJ_GF)) / 1e3
    g = fam["gemm"]
    out = dict(region_wall_ms=wall, region_ms_per_step=wall / steps, bench_ms_per_step=bench["ms_per_step"], any_kernel_busy_ms=all_busy,
               idle_frac=1 - all_busy / wall, families=fam,
               roofline_concurrent=dict(kernel="fp_igemm_f16_fwd (k_conv_sw* + k_igemm*)", bound="mfma", algorithmic_TFLOP=gemm_tf,
                                        busy_ms=g["busy_ms"], achieved=gemm_tf / (g["busy_ms"] / 1e3), peak=MFMA_PEAK, unit="TFLOP/s",
                                        frac=gemm_tf / (g["busy_ms"] / 1e3) / MFMA_PEAK, busy_share_of_region=g["busy_ms"] / wall,
                                        mean_concurrency=g["sum_ms"] / max(g["busy_ms"], 1e-9)),
               clock=bench.get("clock"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
