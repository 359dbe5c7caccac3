import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svim_amd import _abi, _lib, devsynth
o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                          partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=False)
p = _abi.Params.from_options(o)
b, genome, meta = devsynth.make_batch(n_reads=1000000, contig_len=250_000_000, seed=2, device="cuda:0")
bs = b.struct()
eng = _lib.Engine(0)
ts = []
for _ in range(5):
    try:
        eng.collect(bs, p, fetch=False)
    except Exception as e:
        print("collect:", str(e)[:80])
    ts.append(eng.stats()["t_cigar_scan_ms"])
print(os.environ.get("SVX_LIB", "tree"), "scan ms", ["%.3f" % t for t in ts])
