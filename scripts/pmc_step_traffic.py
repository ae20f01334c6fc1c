"""HBM traffic of the dominant entry point over THE LAUNCHES bench.py's `roofline` AVERAGES: the two
rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes) are taken over

    python bench.py --serialize --no-graph --steps 1 --warmup 1 --trace-markers --no-kernel-table --no-cpu-baseline

i.e. one step of the timed workload with the sub-batch launches (126 hypotheses each) on one stream -- the launch sizes and
kernels of the per-kernel table -- and every dispatch between the two marker launches (k_depth_to_xyz on a 1 x 7 image) is
attributed to its entry point.  Output: profiles/traffic.json gets, per entry point, the SUM over the step's launches and the
mean per launch (`hbm_bytes_per_launch`, what bench.py prints next to `avg_launch_ms`).  Units / corrections: counters in KB
(x1024); FETCH_SIZE x2 on gfx950 (wide coalesced reads are tallied at half their bytes); WRITE_SIZE as is.

    python scripts/pmc_step_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]
"""
import collections
import csv
import json
import os
import sys

ENTRY = [("k_vertex", "fp_render_crops"), ("k_bin", "fp_render_crops"), ("k_raster", "fp_render_crops"), ("k_warp", "fp_warp_crops"),
         ("k_conv7x7s2", "fp_conv7x7s2_bn_relu_fwd"), ("k_stem", "fp_stem_fwd"), ("k_conv_sw", "fp_igemm_f16_fwd"), ("k_igemm_pp", "fp_igemm_f16_fwd"),
         ("k_igemm_f16", "fp_igemm_f16_fwd"), ("k_rows512<true, true, true>", "fp_encoder_tail_mean_fwd"), ("k_rows512<true, true", "fp_ffn_layernorm_mean_fwd"), ("k_rows512", "fp_linear_layernorm_fwd"), ("k_pack_w512", "fp_pack_linear512_f16"), ("k_linear512", "fp_linear512_f16_fwd"),
         ("k_ln_mean_finish", "fp_encoder_tail_mean_fwd"), ("k_layernorm_res512", "fp_layernorm_res_fwd"), ("k_colmean512", "fp_colmean_f16_fwd"),
         ("k_attention_f16", "fp_attention_f16_fwd"), ("k_rows_linear", "fp_rows_linear_fwd")]


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "k_depth_to_xyz" in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit(f"{path}: fewer than two marker launches (run bench.py with --trace-markers)")
    region = rows[marks[-2] + 1:marks[-1]]
    tot, n, kern = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(lambda: [0, 0.0])
    for r in region:
        for frag, entry in ENTRY:
            if frag in r["Kernel_Name"]:
                v = float(r["Counter_Value"])
                tot[entry] += v
                n[entry] += 1
                k = kern[(entry, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:])]
                k[0] += 1
                k[1] += v
                break
    return tot, n, kern


def main():
    (fetch, nf, kf), (write, nw, kw) = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    out = {}
    for e in sorted(set(fetch) | set(write)):
        fb, wb = 2.0 * 1024.0 * fetch.get(e, 0.0), 1024.0 * write.get(e, 0.0)
        n = nf.get(e, 0)
        if nw.get(e, n) != n:
            raise SystemExit(f"{e}: {n} dispatches in the FETCH pass, {nw.get(e)} in the WRITE pass -- not the same command")
        out[e] = {"launches": n, "fetch_bytes_sum": fb, "write_bytes_sum": wb, "hbm_bytes_sum": fb + wb,
                  "hbm_bytes_per_launch": (fb + wb) / max(n, 1),
                  "kernels": {name: {"launches": c, "fetch_bytes_sum": 2.0 * 1024.0 * v,
                                     "write_bytes_sum": 1024.0 * kw.get((ent, name), [0, 0.0])[1]}
                              for (ent, name), (c, v) in sorted(kf.items()) if ent == e},
                  "note": "all launches of ONE timed step of `bench.py --serialize` (sub-batch sizes, one stream) between the trace "
                          "markers; FETCH_SIZE x2 (gfx950), KB -> bytes; separate PMC passes"}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({e: {k: v for k, v in d.items() if k != "kernels"} for e, d in out.items()}, indent=1))


if __name__ == "__main__":
    main()
