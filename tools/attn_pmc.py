"""Summarise a rocprofv3 --pmc counter_collection.csv of tools/attn_bench.py per attention kernel (matrix-pipe
utilisation, VALU instruction counts, wait split).  usage: python tools/attn_pmc.py <counter_collection.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"][:44]
    if "attn" not in k and "bwd" not in k:
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sum(dur[k]) / len(dur[k])
    cyc = m["SQ_BUSY_CYCLES"] / 32                       # summed over 32 shader engines
    out = ["%-44s %7.0f us" % (k, us), "SE-busy clock %.0f MHz" % (cyc / us)]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        out.append("mfma busy %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc))
    if "SQ_INSTS_VALU" in m:
        out.append("valu insts/SIMD %.0f" % (m["SQ_INSTS_VALU"] / 1024))
    if "SQ_ACTIVE_INST_VALU" in m:
        out.append("valu active %.3f" % (m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc))
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        out.append("waves/SIMD %.2f" % (w * 4 / 1024 / cyc))
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if n in m:
                out.append("%s %.3f" % (n[3:].lower(), m[n] / w))
    if "SQ_LDS_BANK_CONFLICT" in m:
        out.append("lds bank-conflict cycles / SE-busy %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / 256 / cyc))      # per CU
    print("  ".join(out))
