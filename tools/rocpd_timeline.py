#!/usr/bin/env python3
"""Timeline of the libsvx kernels of the LAST bench step in a rocprofv3 (rocpd SQLite) kernel trace: name, start offset (ms),
duration (ms), one line per launch, so that overlap between the side streams can be read off."""
import os
import sqlite3
import sys


def main(db_path, marker="k_cigar_scan"):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    starts = [s for n, s, e in rows if marker in n and e - s > 100000]
    if not starts:
        return
    t0 = starts[-1]
    for n, s, e in rows:
        if s >= t0 and (os.environ.get("ALL") or n.startswith("k_") or "k_edit" in n or "k_cluster" in n or "k_linkage" in n):
            print("%-28s %9.3f %9.3f" % (n.split("(")[0].replace("void ", "")[:28], (s - t0) / 1e6, (e - s) / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:])
