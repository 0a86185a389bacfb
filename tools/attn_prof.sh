#!/bin/bash
# Per-kernel durations of tools/attn_bench.py (hand-written attention vs the library's) from a rocprofv3 kernel trace.
# usage: tools/attn_prof.sh <out_dir>
set -e -o pipefail
out=$1
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/raw" -o attn --output-format csv -- python3 tools/attn_bench.py > "$out/attn_bench.log" 2>&1
python3 - "$out" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/raw/**/attn_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "attn" in n or "bwd" in n:
        print("%-60s calls %4s  avg %9.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep what "$out/attn_bench.log"
