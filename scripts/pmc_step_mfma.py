"""MFMA-busy share per kernel over ONE timed step of `bench.py --serialize --no-graph --trace-markers` from a rocprofv3 PMC pass
with `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES` (its own run: no other trace domain).
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
(the matrix pipes' busy cycles summed over the chip's SIMDs over the cycles the chip was active for that dispatch; GRBM_GUI_ACTIVE
is summed over the 8 XCDs by rocprofv3 -- the convention of profiles/r02_b_pmc_conv*).

    python scripts/pmc_step_mfma.py <counter_collection.csv> [out.json]
"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by_disp = collections.defaultdict(dict)
for r in rows:
    d = by_disp[int(r["Dispatch_Id"])]
    d["name"] = r["Kernel_Name"]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    if "Start_Timestamp" in r and r.get("Start_Timestamp"):
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ids = sorted(by_disp)
marks = [i for i in ids if "k_depth_to_xyz" in by_disp[i]["name"]]
if len(marks) >= 2:
    ids = [i for i in ids if marks[-2] < i < marks[-1]]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for i in ids:
    d = by_disp[i]
    name = d["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
    a = agg[name]
    a["launches"] += 1
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "ns"):
        a[k] += d.get(k, 0.0)
out = {}
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    if a["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0:
        continue
    out[name] = dict(launches=int(a["launches"]), us_per_launch_under_profiler=a["ns"] / a["launches"] / 1e3,
                     mfma_busy=a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0, 1.0),
                     effective_clock_GHz=(a["GRBM_GUI_ACTIVE"] / 8.0) / max(a["ns"], 1.0))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, {a: round(b, 3) for a, b in v.items()})
