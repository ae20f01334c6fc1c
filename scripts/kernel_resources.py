"""per-kernel resources of the gfx950 code objects of libfp_amd.so (VGPR / AGPR / SGPR counts, spills, static LDS):
    python scripts/kernel_resources.py [lib] [filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "foundationpose_amd", "csrc"))
import check_no_pk_f32 as ck

lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "foundationpose_amd", "csrc", "libfp_amd.so")
flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
for co in ck.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        txt = subprocess.run([f"{ck.LLVM}/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    cur = {}
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("done"):
            cur = {}
        if k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v
        if k == "name" and "vgpr_count" not in cur:
            cur["name"] = v
        if k == "symbol":
            name = subprocess.run(["c++filt", v.replace(".kd", "")], capture_output=True, text=True).stdout.strip()
            cur["sym"] = name
        if k == "wavefront_size":
            if flt in cur.get("sym", ""):
                print(f"{cur.get('sym','?')[:110]:110s} vgpr {cur.get('vgpr_count'):>3s} agpr {cur.get('agpr_count'):>3s} sgpr {cur.get('sgpr_count'):>3s} "
                      f"spill {cur.get('vgpr_spill_count')} lds {cur.get('group_segment_fixed_size')} scratch {cur.get('private_segment_fixed_size')} wg {cur.get('max_flat_workgroup_size')}")
            cur = {}
