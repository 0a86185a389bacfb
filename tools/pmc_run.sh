#!/bin/bash
# Both PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) for every phase of tools/pmc_step.py, then the
# per-kernel summary.  usage: tools/pmc_run.sh <out_dir> [phases...]
set -e -o pipefail
out=$1; shift
phases=${@:-step64 step256 cos13 cos25 ce ce_dead attn sumsq}
mkdir -p "$out"
export TMPDIR=/tmp
for ph in $phases; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d "$out/raw" -o "${ph}_${ctr}" --output-format csv -- python3 tools/pmc_step.py $ph > "$out/raw_${ph}_${ctr}.log" 2>&1
  done
  f=$(find "$out/raw" -name "${ph}_FETCH_SIZE_counter_collection.csv" | head -1)
  w=$(find "$out/raw" -name "${ph}_WRITE_SIZE_counter_collection.csv" | head -1)
  echo "== phase $ph" >> "$out/pmc_hot_kernels_summary.txt"
  python3 tools/pmc_summary.py "$f" "$w" >> "$out/pmc_hot_kernels_summary.txt"
done
cat "$out/pmc_hot_kernels_summary.txt"
