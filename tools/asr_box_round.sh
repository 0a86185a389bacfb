#!/bin/bash
# One more independent draw per flavor for tests/test_success_bits_base.py, start to finish on ONE gpurun box:
#   1. the ORACLE side on the box's host cores (two processes, 8 threads each, wall-clock bounded: whatever prefix of the
#      draw is attacked within the budget is scored and written -- tests/golden/make_asr_fixture.py --time-budget),
#   2. the PRODUCT side on the box's GPU for exactly these two files (VQA_ASR_FULL=1),
#   3. the tie probe (tools/asr_tie_probe.py) for every sample whose success bit differs, while the oracle's adversarial
#      images still exist in the box's /tmp (they never travel: 1.8 MB each).
# usage: tools/asr_box_round.sh <out_dir under gpurun_out> <vlmo seed> <albef seed> [attack seconds]
# The fixtures land in <out_dir>; copy them to tests/golden/ afterwards.  Test infrastructure (runs oracle/).
set -o pipefail
out=$1; sv=$2; sa=$3; budget=${4:-560}
mkdir -p "$out" /tmp/asr_cache_box
fv=asr_base_vlmo_s$sv.json; fa=asr_base_albef_s$sa.json
python tests/golden/make_asr_fixture.py --flavor vlmo --n 120 --seed $sv --threads 8 --cache /tmp/asr_cache_box \
    --time-budget $budget --out "$out/$fv" > "$out/gen_vlmo_s$sv.log" 2>&1 &
pv=$!
python tests/golden/make_asr_fixture.py --flavor albef --n 120 --seed $sa --threads 8 --cache /tmp/asr_cache_box \
    --time-budget $budget --sizes 12,16 --out "$out/$fa" > "$out/gen_albef_s$sa.log" 2>&1 &
pa=$!
wait $pv; rv=$?
wait $pa; ra=$?
tail -n 2 "$out/gen_vlmo_s$sv.log" "$out/gen_albef_s$sa.log"
[ $rv -eq 0 ] && [ $ra -eq 0 ] || { echo "fixture generation failed ($rv, $ra)"; exit 1; }
cp "$out/$fv" "$out/$fa" tests/golden/
VQA_ASR_FULL=1 VQA_ASR_SETS_LOG="$out/sets_s${sv}_s${sa}.jsonl" python -m pytest tests/test_success_bits_base.py -m gpu -q -s \
    -k "vlmo_s$sv or albef_s$sa" > "$out/pytest_s${sv}_s${sa}.log" 2>&1
echo "pytest rc $?"; grep -E "base: n =|passed|failed" "$out/pytest_s${sv}_s${sa}.log"
python - "$out" "$sv" "$sa" <<'PY'
import json, os, subprocess, sys
out, sv, sa = sys.argv[1:4]
log = os.path.join(out, "sets_s{}_s{}.jsonl".format(sv, sa))
for line in open(log) if os.path.exists(log) else []:
    rec = json.loads(line)
    if not rec["differ"]:
        continue
    seed = sv if rec["flavor"] == "vlmo" else sa
    cache = "/tmp/asr_cache_box/{}_seed{}_b40".format(rec["flavor"], seed)
    cmd = [sys.executable, "tools/asr_tie_probe.py", "--fixture", os.path.join("tests", "golden", rec["fixture"]),
           "--samples", ",".join(str(s) for s in rec["differ"]), "--oracle-adv", cache]
    with open(os.path.join(out, "asr_tie_probes.jsonl"), "a") as f:
        subprocess.run(cmd, stdout=f, stderr=open(os.path.join(out, "tie_probe.err"), "a"))
PY
