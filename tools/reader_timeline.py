#!/usr/bin/env python3
"""Timeline of the LAST pass of tools/device_reader_rate.py in a rocprofv3 trace (rocpd SQLite, --kernel-trace --memory-copy-trace): every k_bgzf_inflate launch and
every memory copy above 1 MB with start offset and duration (ms) - do the H2D copies and the kernels of the three inflater slots overlap?"""
import sqlite3
import sys


def main(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    for n, s, e in cur.execute("select %s, start, end from kernels" % name_col):
        if any(k in n for k in ("k_bgzf_inflate", "k_crc32", "k_anchor", "k_walk", "k_measure", "k_fields", "k_cigar_copy", "k_name_")):
            ev.append((s, e, n.split("(")[0].replace("void ", "")))
    mc = [t for t in tables if "memory_cop" in t]
    for t in mc[:1]:
        c2 = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
        size_col = [c for c in c2 if "size" in c or "bytes" in c]
        name_c = [c for c in c2 if c in ("name", "kind", "direction")]
        q = "select start, end, %s, %s from %s" % (size_col[0] if size_col else "0", name_c[0] if name_c else "''", t)
        for s, e, b, k in cur.execute(q):
            if b and b > (1 << 20):
                ev.append((s, e, "copy %s %.0f MB" % (k, b / 1e6)))
    if not ev:
        print("no events; tables:", tables)
        return
    ev.sort()
    # the last pass: events after the last gap of more than 150 ms
    cut = 0
    for i in range(1, len(ev)):
        if ev[i][0] - max(x[1] for x in ev[max(0, i - 8):i]) > 150e6:
            cut = i
    t0 = ev[cut][0]
    for s, e, n in ev[cut:]:
        print("%-34s %9.3f %9.3f" % (n[:34], (s - t0) / 1e6, (e - s) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
