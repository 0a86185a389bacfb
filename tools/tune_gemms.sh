#!/bin/bash
# Record the fastest library solution (hipBLASLt / rocBLAS) for every GEMM shape of the bench workloads (PyTorch TunableOp,
# tuning ON), merge the per-workload results, then A/B the merged file against the library defaults on the same box.
#   usage: [TUNE_SET="default albef large batch1"] [AB=1] tools/tune_gemms.sh <out_dir>      (raw_*.csv of earlier calls are merged too)
# batch1 = the reference's own call shape (one sample: 591 / 586-row GEMMs at 384 px, 915-row at 480 px, ALBEF's 577 + text
# rows), through tools/bench_reference_style.py.
# The CSV to track is <out_dir>/tunableop_mi355x_rocm72.csv (copy it to vqattack_amd/tuning/).
set -e -o pipefail
out=$1
mkdir -p "$out"
tune() {   # tag, bench args...
  tag=$1; shift
  rm -f "$out/raw_${tag}"*.csv
  # RESUME=<tracked.csv>: start from already recorded shapes (TunableOp reads the file first and tunes only what is missing)
  [ -n "$RESUME" ] && cp "$RESUME" "$out/raw_${tag}0.csv"
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME="$out/raw_${tag}.csv" \
    PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 VQA_TUNED_GEMMS=off \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-b256 "$@" > "$out/tuning_${tag}.json" 2> "$out/tuning_${tag}.err"
  ls "$out"/raw_${tag}*.csv
}
tune_batch1() {
  rm -f "$out/raw_batch1"*.csv
  [ -n "$RESUME" ] && cp "$RESUME" "$out/raw_batch10.csv"
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME="$out/raw_batch1.csv" \
    PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 VQA_TUNED_GEMMS=off \
    python3 tools/bench_reference_style.py --models vlmo_base,albef_base --image480 --steps 2 --reps 1 --graph off \
    > "$out/tuning_batch1.json" 2> "$out/tuning_batch1.err"
  ls "$out"/raw_batch1*.csv
}
for w in ${TUNE_SET:-default albef large}; do
  case $w in
    batch1) tune_batch1 ;;
    default) tune default ;;
    albef) tune albef_b256 --model albef_base --batch 256 --pgd-steps 4 ;;
    large) tune vlmo_large_joint --model vlmo_large --batch 128 --joint 8 --pgd-steps 9 ;;
  esac
done
python3 tools/merge_tunable.py "$out/tunableop_mi355x_rocm72.csv" "$out"/raw_*.csv
wc -l "$out/tunableop_mi355x_rocm72.csv"
[ "${AB:-1}" = "1" ] || exit 0
for i in 1 2; do
  VQA_TUNED_GEMMS=off python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b256 > "$out/ab_default_$i.json" 2> "$out/ab_default_$i.err"
  VQA_TUNED_GEMMS="$out/tunableop_mi355x_rocm72.csv" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-b256 > "$out/ab_tuned_$i.json" 2> "$out/ab_tuned_$i.err"
done
for f in "$out"/ab_*.json; do echo "$f $(python3 -c "import json,sys; r=json.load(open('$f')); print(r['value'], r.get('tuned_gemms'))")"; done
