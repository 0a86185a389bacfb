#!/bin/bash
# Per-kernel device time of one benchmarked attack: rocprofv3 kernel trace + stats of `python3 bench.py <args>`, summarised
# by tools/kernel_stats_summary.py.  The raw per-launch trace (tens of MB) is deleted; the stats CSV, the summary and the
# bench line (stdout of the profiled run) are kept.   usage: tools/prof_attack.sh <out_dir> <tag> <bench.py args...>
set -e -o pipefail
out=$1; tag=$2; shift 2
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw_$tag" -o "$tag" -- python3 bench.py "$@" \
  > "$out/bench_${tag}_under_rocprof.json" 2> "$out/bench_${tag}.err"
stats=$(find "$out/raw_$tag" -name "${tag}_kernel_stats.csv" | head -1)
cp "$stats" "$out/bench_${tag}_kernel_stats.csv"
python3 tools/kernel_stats_summary.py "$out/bench_${tag}_kernel_stats.csv" 30 > "$out/bench_${tag}_summary.txt"
rm -rf "$out/raw_$tag"
tail -12 "$out/bench_${tag}_summary.txt"
