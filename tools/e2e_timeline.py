import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
def tables(stem):
    return [r[0] for r in con.execute("SELECT name FROM sqlite_master WHERE type='table' AND name LIKE ?", (stem + "%",))]
kd = tables("rocpd_kernel_dispatch")[0]; st = tables("rocpd_string")[0]
ks = tables("rocpd_info_kernel_symbol")
names = {i: n for i, n in con.execute("SELECT id, string FROM %s" % st)}
ksym = {i: n for i, n in con.execute("SELECT id, kernel_name FROM %s" % ks[0])} if ks else None
rows = []
for kid, s, e in con.execute("SELECT kernel_id, start, end FROM %s" % kd):
    n = (ksym[kid] if ksym else names.get(kid, str(kid))).split("(")[0]
    rows.append((s, e, "K " + n[:40]))
mc = tables("rocpd_memory_copy")
if mc:
    cols = [r[1] for r in con.execute("PRAGMA table_info(%s)" % mc[0])]
    for r in con.execute("SELECT start, end, size FROM %s" % mc[0]):
        rows.append((r[0], r[1], "C copy %.1f MB" % (r[2] / 1e6)))
rows.sort()
# the last pass: from the last but one 'k_bgzf_inflate' burst; print everything in the final 0.8 s, merging tiny kernels
t_end = rows[-1][1]
sel = [r for r in rows if r[0] > t_end - int(float(sys.argv[2]) * 1e9)]
t0 = sel[0][0]
for s, e, n in sel:
    if (e - s) > 300000 or "inflate" in n: print("%9.3f %9.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
