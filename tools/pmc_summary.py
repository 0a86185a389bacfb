"""Summarise rocprofv3 --pmc counter_collection.csv files for the vqa:: kernels.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests at 64 B,
i.e. exactly half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section), so it is doubled here.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv>
                      [--json records.json --phase <name> --shape '<json>']
With --json every kernel's per-launch figures are appended to a JSON list of records
{phase, kernel, grid, shape, fetch_bytes, write_bytes, traffic_bytes, launches, dur_us, source_sha256}: bench.py matches
a run's launch against `kernel` and `shape` -- and the sha256 of the kernel's source file + common.hpp + the C-ABI header
(vqattack_amd.build.kernel_source_digest) against the tree it runs from -- before it copies `traffic_bytes` into its line.
"""
import csv
import json
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqattack_amd.build import kernel_source_digest  # noqa: E402


def load(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "vqa::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[(name, int(r["Grid_Size"]))].append((float(r["Counter_Value"]),
                                                  int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
opts = dict(zip(sys.argv[3::2], sys.argv[4::2]))
records = []
for key in sorted(fetch):
    if len(fetch[key]) > 1:          # the first launch of a kernel pays code-object / TLB warm-up: report the others
        fetch[key] = fetch[key][1:]
        if len(write.get(key, [])) > 1:
            write[key] = write[key][1:]
    f = sum(v for v, _ in fetch[key]) / len(fetch[key])
    w = sum(v for v, _ in write.get(key, [(0, 0)])) / max(len(write.get(key, [])), 1)
    dur = sum(d for _, d in fetch[key]) / len(fetch[key])
    fetch_b, write_b = 2 * f * 1024, w * 1024
    print("%-96s grid %9d  launches %d  FETCH_SIZE %.0f KiB (x2 -> %.1f MB)  WRITE_SIZE %.0f KiB (%.1f MB)  "
          "HBM traffic %.1f MB  dur(profiled) %.1f us" % (key[0][:96], key[1], len(fetch[key]), f, fetch_b / 1e6, w,
                                                         write_b / 1e6, (fetch_b + write_b) / 1e6, dur / 1e3))
    records.append(dict(phase=opts.get("--phase"), kernel=key[0], grid=key[1],
                        shape=json.loads(opts["--shape"]) if opts.get("--shape") else None,
                        fetch_bytes=int(round(fetch_b)), write_bytes=int(round(write_b)),
                        traffic_bytes=int(round(fetch_b + write_b)), launches=len(fetch[key]),
                        dur_us=round(dur / 1e3, 1), source_sha256=kernel_source_digest(key[0])))
if opts.get("--json"):
    path = opts["--json"]
    old = json.load(open(path)) if os.path.exists(path) else []
    old = [r for r in old if r.get("phase") != opts.get("--phase")]      # re-running a phase replaces its records
    json.dump(old + records, open(path, "w"), indent=1)
