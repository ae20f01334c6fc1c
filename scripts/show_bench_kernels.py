"""print the stage line and the per-entry-point table of a bench.py JSON line (gpurun logs): python scripts/show_bench_kernels.py FILE"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("ms_per_step", round(d["ms_per_step"], 3), "clock", {k: round(v, 1) for k, v in d.get("clock", {}).items() if isinstance(v, float)})
print("stage_raster_crop", d.get("stage_raster_crop"))
if "roofline" in d:
    print("roofline frac", d["roofline"]["frac"], "concurrent", d["roofline"].get("concurrent", {}).get("frac"))
for k, v in d.get("kernels", {}).items():
    print(f"  {k:28s} calls {v.get('calls'):5d}  avg {v.get('avg_ms') * 1e3:8.1f} us" + (f"  {v['GBps']:7.0f} GB/s" if "GBps" in v else "") +
          (f"  {v['TFLOPs']:6.0f} TF/s" if "TFLOPs" in v else ""))
