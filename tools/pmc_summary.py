#!/usr/bin/env python3
"""Per-kernel sum of a PMC counter from a rocprofv3 (rocpd SQLite) run: kernel, dispatches, total, per-dispatch."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    dcol = "dispatch_id" if "dispatch_id" in cols else None
    q = "select %s, %s, sum(value), count(distinct %s) from counters_collection group by 1, 2 order by 3 desc" % (kcol, ccol, dcol or "rowid")
    lines = ["Kernel,Counter,Total,Dispatches,PerDispatch"]
    for k, c, v, n in cur.execute(q):
        k = k.split("(")[0]
        if not (k.startswith("k_") or k.startswith("void k_")):
            continue
        lines.append('"%s",%s,%.1f,%d,%.1f' % (k, c, v, n, v / max(1, n)))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
