"""Kernel timeline of ONE small CLUSTER call (default 1000 reads) from a rocprofv3 kernel trace: run
   rocprofv3 --kernel-trace -d /tmp/sb -o p -- python tools/small_batch_trace.py run [n_reads]
then  python tools/small_batch_trace.py show <db>"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch, numpy as np
    from svim_amd import _abi, _lib, devsynth
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                              partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=False)
    p = _abi.Params.from_options(o)
    eng = _lib.Engine(0)
    b, genome, meta = devsynth.make_batch(n_reads=n, contig_len=max(2_000_000, 250 * n), seed=2, device="cuda:0")
    g_off = torch.tensor([0, genome.numel()], dtype=torch.int64, device="cuda:0")
    eng.set_genome(g_off, genome, on_device=True)
    rank = b.t["contig_rank"].cpu().numpy().astype(np.int32)
    bs = b.struct()
    for _ in range(5):
        eng.collect(bs, p, fetch=False); eng.cluster(p, rank, source=0, fetch=False)
    torch.cuda.synchronize()
    time.sleep(0.05)
    t0 = time.perf_counter()
    eng.cluster(p, rank, source=0, fetch=False)
    torch.cuda.synchronize()
    print("last cluster call: %.3f ms wall" % (1e3 * (time.perf_counter() - t0)))
else:
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    # the last call = everything after the largest idle gap near the end
    gaps = [(rows[i + 1][1] - rows[i][2], i + 1) for i in range(len(rows) - 1)]
    g, first = max(gaps[-400:])
    t0 = rows[first][1]
    busy = 0
    for nme, s, e in rows[first:]:
        print("%-40s %9.3f %9.3f" % (nme.split("(")[0].replace("void ", "")[:40], (s - t0) / 1e3, (e - s) / 1e3))
        busy += e - s
    print("%d launches, GPU busy %.1f us of %.1f us" % (len(rows) - first, busy / 1e3, (rows[-1][2] - t0) / 1e3))
