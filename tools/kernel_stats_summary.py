"""Summary of a rocprofv3 `--kernel-trace --stats` kernel_stats.csv: the top kernels by device time, every hand-written
kernel (`vqa::`), and the split GEMM / attention / hand-written attack path / other.
usage: python tools/kernel_stats_summary.py <kernel_stats.csv> [top_n]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.1f ms over %d kernels" % (tot / 1e6, len(rows)))

    def line(r, tag=""):
        t = float(r["TotalDurationNs"])
        return "%s%7.3f%%  %9.1f ms  calls %6s  avg %9.2f us  %s" % (tag, t / tot * 100, t / 1e6, r["Calls"],
                                                                    float(r["AverageNs"]) / 1e3, r["Name"][:100])
    for r in rows[:top]:
        print(line(r))
    groups = {"library GEMM (Cijk_*)": 0.0, "vqa::attn_* (hand-written fp32 MFMA attention)": 0.0,
              "vqa::* attack path (step, loss, CE, text, image)": 0.0, "other (ATen elementwise, LayerNorm, GELU, copies)": 0.0}
    for r in rows:
        n, t = r["Name"], float(r["TotalDurationNs"])
        if n.startswith("Cijk"):
            groups["library GEMM (Cijk_*)"] += t
        elif "vqa::attn" in n:
            groups["vqa::attn_* (hand-written fp32 MFMA attention)"] += t
        elif "vqa::" in n:
            groups["vqa::* attack path (step, loss, CE, text, image)"] += t
        else:
            groups["other (ATen elementwise, LayerNorm, GELU, copies)"] += t
        if "vqa::" in n:
            print(line(r, "VQA "))
    for k, v in groups.items():
        print("SPLIT %6.2f%%  %9.1f ms  %s" % (v / tot * 100, v / 1e6, k))


if __name__ == "__main__":
    main()
