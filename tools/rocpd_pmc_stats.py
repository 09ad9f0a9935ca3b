"""Per-kernel average of one rocprofv3 PMC counter from a rocpd sqlite file (counter value per dispatch)."""
import json
import sqlite3
import sys


def main(path, counter, out=None):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = {}
    for name, v in rows:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += v
    res = {name[:120]: {"dispatches": a[0], "avg": a[1] / a[0], "total": a[1]} for name, a in agg.items()}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["total"])[:40])
    text = json.dumps({"counter": counter, "kernels": res}, indent=1)
    if out:
        open(out, "w").write(text + "\n")
    print(text[:1500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
