#!/bin/bash
# Further independent draws for tests/test_success_bits_base.py, start to finish on ONE gpurun box:
#   1. the ORACLE side on the box's host cores (one process per seed, the 16-core share divided between them, wall-clock
#      bounded: whatever prefix of a draw is attacked within the budget is scored and written --
#      tests/golden/make_asr_fixture.py --time-budget),
#   2. the PRODUCT side on the box's GPU for exactly these files (VQA_ASR_FULL=1),
#   3. the tie probe (tools/asr_tie_probe.py) for every sample whose success bit differs, while the oracle's adversarial
#      images still exist in the box's /tmp (they never travel: 1.8 MB each).
# usage: tools/asr_box_round.sh <out_dir under gpurun_out> "<vlmo seeds>" "<albef seeds>" [attack seconds]
# The fixtures land in <out_dir> and are copied into tests/golden/ of the BOX's snapshot so that the product run can read
# them (provisional: that tree does not travel back); copy them from <out_dir> to tests/golden/ afterwards -- only when this
# script exits 0: its exit status is the product-side pytest's.  Test infrastructure (runs oracle/).
set -o pipefail
out=$1; vs=($2); as=($3); budget=${4:-560}
mkdir -p "$out" /tmp/asr_cache_box
n=$(( ${#vs[@]} + ${#as[@]} ))
[ $n -gt 0 ] || { echo "usage: $0 <out_dir> \"<vlmo seeds>\" \"<albef seeds>\" [seconds]: no seed given"; exit 2; }
threads=$(( 16 / n )); [ $threads -lt 1 ] && threads=1
pids=(); keys=()
for s in "${vs[@]}"; do
  python tests/golden/make_asr_fixture.py --flavor vlmo --n 120 --seed $s --threads $threads --cache /tmp/asr_cache_box \
      --time-budget $budget --out "$out/asr_base_vlmo_s$s.json" > "$out/gen_vlmo_s$s.log" 2>&1 &
  pids+=($!); keys+=("vlmo_s$s")
done
for s in "${as[@]}"; do
  python tests/golden/make_asr_fixture.py --flavor albef --n 120 --seed $s --threads $threads --cache /tmp/asr_cache_box \
      --time-budget $budget --sizes 12,16 --out "$out/asr_base_albef_s$s.json" > "$out/gen_albef_s$s.log" 2>&1 &
  pids+=($!); keys+=("albef_s$s")
done
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
for k in "${keys[@]}"; do tail -n 1 "$out/gen_$k.log"; done
[ $fail -eq 0 ] || { echo "a fixture generator failed"; exit 1; }
sel=""
for k in "${keys[@]}"; do cp "$out/asr_base_$k.json" tests/golden/; sel="$sel${sel:+ or }$k"; done
tag=$(echo "${keys[@]}" | tr ' ' '_')
VQA_ASR_FULL=1 VQA_ASR_SETS_LOG="$out/sets_$tag.jsonl" python -m pytest tests/test_success_bits_base.py -m gpu -q -s \
    -k "$sel" > "$out/pytest_$tag.log" 2>&1
rc=$?
echo "pytest rc $rc"; grep -E "base: n =|passed|failed" "$out/pytest_$tag.log"
python - "$out" "$out/sets_$tag.jsonl" <<'PY'
import json, os, subprocess, sys
out, log = sys.argv[1:3]
for line in open(log) if os.path.exists(log) else []:
    rec = json.loads(line)
    if not rec["differ"]:
        continue
    seed = rec["fixture"].split("_s")[-1].split(".")[0]
    cache = "/tmp/asr_cache_box/{}_seed{}_b40".format(rec["flavor"], seed)
    cmd = [sys.executable, "tools/asr_tie_probe.py", "--fixture", os.path.join("tests", "golden", rec["fixture"]),
           "--samples", ",".join(str(s) for s in rec["differ"]), "--oracle-adv", cache]
    with open(os.path.join(out, "asr_tie_probes.jsonl"), "a") as f:
        subprocess.run(cmd, stdout=f, stderr=open(os.path.join(out, "tie_probe.err"), "a"))
PY
exit $rc
