"""Per-kernel averages of every PMC counter in one or more rocpd sqlite files -> markdown table."""
import sqlite3
import sys


def main(out, paths):
    agg = {}
    for path in paths:
        db = sqlite3.connect(path)
        for name, ctr, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
            if "conv" not in name:
                continue
            a = agg.setdefault((name[:70], ctr), [0, 0.0])
            a[0] += 1
            a[1] += v
    kernels = sorted({k for k, _ in agg})
    ctrs = sorted({c for _, c in agg})
    lines = ["| counter | " + " | ".join(kernels) + " |", "|---|" + "---|" * len(kernels)]
    for c in ctrs:
        lines.append("| " + c + " | " + " | ".join(
            f"{agg[(k, c)][1] / agg[(k, c)][0]:.4g}" if (k, c) in agg else "-" for k in kernels) + " |")
    text = "\n".join(lines) + "\n"
    open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
