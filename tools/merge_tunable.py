"""Merge TunableOp result files (one per tuned workload) into one: the validator header must be identical in all of them;
a shape recorded more than once keeps the entry with the shortest measured time.
usage: merge_tunable.py <out.csv> <in.csv>..."""
import sys

out, header, best = sys.argv[1], None, {}
for path in sys.argv[2:]:
    lines = [ln.strip() for ln in open(path) if ln.strip()]
    head = [ln for ln in lines if ln.startswith("Validator,")]
    if header is None:
        header = head
    elif head != header:
        raise SystemExit("validator header of {} differs: {} vs {}".format(path, head, header))
    for ln in lines:
        if ln.startswith("Validator,"):
            continue
        op, shape, solution, ms = ln.split(",")
        key = (op, shape)
        if key not in best or float(ms) < float(best[key][1]):
            best[key] = (solution, ms)
with open(out, "w") as f:
    f.write("\n".join(header) + "\n")
    for (op, shape), (solution, ms) in sorted(best.items()):
        f.write("{},{},{},{}\n".format(op, shape, solution, ms))
print("merged {} shapes from {} files into {}".format(len(best), len(sys.argv) - 2, out))
