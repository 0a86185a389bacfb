#!/bin/bash
# What the score / dS^T stores of the attention kernels wait for: SQ's cycles per vector-memory write instruction, the
# texture addresser's stalls behind the L1 / L2, and the L2's stalls behind the memory fabric -- rocprofv3 --pmc passes
# (counters only) over tools/attn_ab.py for one build of the library.  usage: tools/attn_store_pmc.sh <out_dir> <library.so>
set -e -o pipefail
out=$1
lib=$2
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD" \
           "SQ_BUSY_CYCLES TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_BUSY_CYCLES TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum"; do
  i=$((i + 1))
  rocprofv3 --pmc $set -d "$out/pass$i" -o p --output-format csv -- python3 tools/attn_ab.py "$lib" 1 > "$out/pass$i.log" 2>&1
  f=$(find "$out/pass$i" -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set" >> "$out/summary.txt"
  python3 - "$f" >> "$out/summary.txt" <<'PY'
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:44]
    if "attn" not in k:
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sum(dur[k]) / len(dur[k])
    cyc = m["SQ_BUSY_CYCLES"] / 32          # per shader engine; 32 of them
    parts = ["%-44s %7.0f us  SE-busy cycles %.3g" % (k, us, cyc)]
    for n, v in sorted(m.items()):
        if n != "SQ_BUSY_CYCLES":
            parts.append("%s %.4g" % (n, v))
    if "SQ_INSTS_VMEM_WR" in m and m["SQ_INSTS_VMEM_WR"]:
        parts.append("cycles per write instruction %.0f" % (m["SQ_INST_CYCLES_VMEM_WR"] / m["SQ_INSTS_VMEM_WR"]))
    if "SQ_INSTS_VMEM_RD" in m and m["SQ_INSTS_VMEM_RD"]:
        parts.append("cycles per read instruction %.0f" % (m["SQ_INST_CYCLES_VMEM_RD"] / m["SQ_INSTS_VMEM_RD"]))
    print("  ".join(parts))
PY
  rm -rf "$out/pass$i"
done
cat "$out/summary.txt"
