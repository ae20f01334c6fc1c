export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+)\b" | sort -u | tr '\n' ' ' | head -c 6000; echo
for t in 128x128 256x256; do
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT"; do
  d=gpurun_out/pmc9_${t}_$(echo $c | cut -c1-12 | tr ' ' '_')
  FP_IGEMM_TILE=$t timeout 120 rocprofv3 --pmc $c --output-format csv -d $d -o k -- python scripts/one_conv.py > /dev/null 2>&1
  python - "$d" "$t" <<'PY'
import csv,sys,collections,glob
d,t=sys.argv[1],sys.argv[2]
f=glob.glob(d+"/*counter_collection.csv")
if not f: print(t,"no output"); sys.exit()
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "igemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in agg.items():
    print(t, k, "last:", v[-1][0], "dur_us", v[-1][1]/1e3)
PY
done; done
