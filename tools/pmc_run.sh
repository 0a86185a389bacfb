#!/bin/bash
# Both PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) for every phase of tools/pmc_step.py, then the
# per-kernel summary (text) and the shape-keyed records bench.py reads (pmc_traffic.json).
# usage: tools/pmc_run.sh <out_dir> [phases...]
set -e -o pipefail
out=$1; shift
phases=${@:-step64 step256 cos13 cos25 ce ce_dead ce_dead_fresh attn sumsq}
mkdir -p "$out"
export TMPDIR=/tmp
for ph in $phases; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d "$out/raw" -o "${ph}_${ctr}" --output-format csv -- python3 tools/pmc_step.py $ph > "$out/raw_${ph}_${ctr}.log" 2>&1
  done
  f=$(find "$out/raw" -name "${ph}_FETCH_SIZE_counter_collection.csv" | head -1)
  w=$(find "$out/raw" -name "${ph}_WRITE_SIZE_counter_collection.csv" | head -1)
  shape=$(grep -h '^PMC_SHAPE ' "$out/raw_${ph}_FETCH_SIZE.log" | tail -1 | cut -d' ' -f2-)
  echo "== phase $ph  shape $shape" >> "$out/pmc_hot_kernels_summary.txt"
  python3 tools/pmc_summary.py "$f" "$w" --json "$out/pmc_traffic.json" --phase "$ph" --shape "$shape" >> "$out/pmc_hot_kernels_summary.txt"
done
cat "$out/pmc_hot_kernels_summary.txt"
