#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / min / max / % - the same table
`rocprofv3 --stats` prints, written as CSV so that it can be committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        name = name.split("(")[0]
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%d,%.1f,%.4f,%d,%d' % (name, a[0], a[1], a[1] / a[0], 100.0 * a[1] / total, a[2], a[3]))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
