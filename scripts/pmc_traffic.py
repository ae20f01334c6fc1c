"""Turns the two rocprofv3 PMC passes over scripts/run_kernels.py (--pmc FETCH_SIZE, --pmc WRITE_SIZE, collected in
separate runs as MI355X_MICROARCH.md prescribes) into profiles/traffic.json: HBM bytes per launch for every hand-written
kernel.  Units and corrections: the counters are in KB (x1024); on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read stream (x2); WRITE_SIZE is taken as is.

    python scripts/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]
"""
import collections, csv, json, os, sys

# kernel-name fragment -> key in traffic.json.  fp_igemm_f16_fwd is reported for its representative launch, the 256->256
# 3x3 convolution (k_conv_sw in the product library, k_igemm_pp / k_igemm_f16 in a profiling build); the in_proj GEMM of the
# same entry point (k_igemm_pp) is kept apart.
ENTRY = {"k_vertex": "fp_render_crops", "k_bin": "fp_render_crops", "k_raster": "fp_render_crops", "k_warp": "fp_warp_crops", "k_conv7x7s2": "fp_conv7x7s2_bn_relu_fwd",
         "k_conv_sw": "fp_igemm_f16_fwd", "k_igemm_pp": "fp_igemm_f16_fwd(in_proj)", "k_igemm_f16": "fp_igemm_f16_fwd(128x128)",
         "k_layernorm_res512": "fp_layernorm_res_fwd", "k_add_pe512": "fp_add_pe_f16_fwd",
         "k_colmean512": "fp_colmean_f16_fwd", "k_attention_f16": "fp_attention_f16_fwd"}


def load(path, counter):
    per_kernel = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for k in ENTRY:
            if k in r["Kernel_Name"]:
                per_kernel[k].append(float(r["Counter_Value"]))
    # an entry point that is several kernels (fp_render_crops = k_vertex + k_bin + k_raster): sum of the per-kernel
    # means over the warm launches
    agg = collections.defaultdict(float)
    n = {}
    for k, vals in per_kernel.items():
        warm = vals[1:] or vals
        agg[ENTRY[k]] += sum(warm) / len(warm)
        n[ENTRY[k]] = len(warm)
    return agg, n


def main():
    (fetch, nf), (write, _) = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for e in sorted(set(fetch) | set(write)):
        fb = 2.0 * 1024.0 * fetch.get(e, 0.0)
        wb = 1024.0 * write.get(e, 0.0)
        out[e] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                  "launches": nf.get(e, 0), "note": "FETCH_SIZE x2 (gfx950), KB -> bytes; mean over the warm launches of scripts/run_kernels.py"}
    path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
