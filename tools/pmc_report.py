"""Summarise rocprofv3 --pmc passes written by tools/pmc.sh: per kernel, mean counter values per dispatch."""
import collections, csv, glob, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{d}/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void pidm::", "")[:44]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    pats = [a_ for a_ in sys.argv[2:] if not a_.startswith("--") and not a_.replace(".", "").isdigit()]
    if not any(s in k for s in pats or ["conv", "la_", "gn_", "mid_", "layernorm"]):
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(f"== {k}  (n={len(next(iter(v.values())))})")
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print("   " + "  ".join(f"{c.replace('SQ_','')}={val:.3g}" for c, val in sorted(m.items())))
    if "SQ_INSTS_MFMA" in m and "GRBM_GUI_ACTIVE" in m:
        # Matrix-pipe occupancy from the INSTRUCTION count (SQ_VALU_MFMA_BUSY_CYCLES reads the same value for kernels of very
        # different duration on this stack and is not used): wave-level MFMAs x cycles each / (1024 SIMDs x kernel cycles).
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; cycles per MFMA: argv "--mfma-cycles N" (32 for the 8-pass bf16 32x32x16,
        # 64 for the 16-pass fp32 32x32x2; default 32).
        cyc = float(sys.argv[sys.argv.index("--mfma-cycles") + 1]) if "--mfma-cycles" in sys.argv else 32.0
        kcyc = m["GRBM_GUI_ACTIVE"] / 8.0
        print(f"   waves={m.get('SQ_WAVES',0):.0f} MFMA insts/wave={m.get('SQ_INSTS_MFMA',0)/max(m.get('SQ_WAVES',1),1):.0f} "
              f"mfma_pipe_occupancy={m['SQ_INSTS_MFMA'] * cyc / 1024.0 / max(kcyc, 1.0):.3f} (at {cyc:.0f} cycles per MFMA) "
              f"valu/mfma={m.get('SQ_INSTS_VALU',0)/max(m['SQ_INSTS_MFMA'],1):.1f} salu/mfma={m.get('SQ_INSTS_SALU',0)/max(m['SQ_INSTS_MFMA'],1):.1f} "
              f"wait_any/wave_cyc={m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst/wave_cyc={m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} "
              f"active/wave_cyc={m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} lds_conf/lds_active={m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1):.2f}")
