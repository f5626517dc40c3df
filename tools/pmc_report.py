"""Summarise rocprofv3 --pmc passes written by tools/pmc.sh: per kernel, mean counter values per dispatch."""
import collections, csv, glob, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{d}/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void pidm::", "")[:44]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if not any(s in k for s in sys.argv[2:] or ["conv", "la_", "gn_", "mid_", "layernorm"]):
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(f"== {k}  (n={len(next(iter(v.values())))})")
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print("   " + "  ".join(f"{c.replace('SQ_','')}={val:.3g}" for c, val in sorted(m.items())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        print(f"   waves={m.get('SQ_WAVES',0):.0f} MFMA insts/wave={m.get('SQ_INSTS_MFMA',0)/max(m.get('SQ_WAVES',1),1):.0f} "
              f"mfma_busy/gui_active={m['SQ_VALU_MFMA_BUSY_CYCLES']/max(m.get('GRBM_GUI_ACTIVE',1),1)/ (256*4) :.3f} "
              f"wait_any/wave_cyc={m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst/wave_cyc={m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} "
              f"active/wave_cyc={m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} lds_conf/lds_active={m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1):.2f}")
