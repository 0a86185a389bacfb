#!/bin/bash
# Record the fastest hipBLASLt solution for every GEMM shape of a bench workload (PyTorch TunableOp, tuning ON), then A/B
# the recorded file against the library defaults on the same box.   usage: tools/tune_gemms.sh <out_dir> [bench.py args...]
# The CSV to track is <out_dir>/tunableop_mi355x_rocm72.csv (copy it to vqattack_amd/tuning/).
set -e -o pipefail
out=$1; shift
mkdir -p "$out"
csv="$out/tunableop_mi355x_rocm72.csv"
rm -f "$csv"
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME="$csv" PYTORCH_TUNABLEOP_VERBOSE=0 \
  PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 VQA_TUNED_GEMMS=off \
  python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-b256 "$@" > "$out/tuning_run.json" 2> "$out/tuning_run.err"
ls -la "$out"
wc -l "$csv"
for i in 1 2; do
  VQA_TUNED_GEMMS=off python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b256 "$@" > "$out/ab_default_$i.json" 2> "$out/ab_default_$i.err"
  VQA_TUNED_GEMMS="$csv" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b256 "$@" > "$out/ab_tuned_$i.json" 2> "$out/ab_tuned_$i.err"
done
for f in "$out"/ab_*.json; do echo "$f $(python3 -c "import json,sys; r=json.load(open('$f')); print(r['value'], r.get('tuned_gemms'))")"; done
