#!/bin/bash
# Counter passes (rocprofv3 --pmc, one set per pass, no trace domains) over tools/attn_ab.py for ONE build of the library.
# usage: tools/attn_ab_pmc.sh <out_dir> <library.so>     -> <out_dir>/pass*/… and <out_dir>/summary.txt
set -e -o pipefail
out=$1
lib=$2
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i + 1))
  rocprofv3 --pmc $set -d "$out/pass$i" -o p --output-format csv -- python3 tools/attn_ab.py "$lib" 1 > "$out/pass$i.log" 2>&1
  f=$(find "$out/pass$i" -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set" >> "$out/summary.txt"
  python3 tools/attn_pmc.py "$f" >> "$out/summary.txt"
done
cat "$out/summary.txt"
